"""Shared helpers for the parity tests (env side)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Parity tolerances for ONE control step (5 substeps) started from the identical fp32 state, MEASURED (profiles/parity_r02.json,
# tools/parity_quantiles.py: 8192 envs x 12 teacher-forced steps, errors against the fp64 oracle):
#              median     p99      p99.9    max
#   HIP  qpos  1.8e-7   7.8e-7   1.4e-6   4.2e-2      oracle fp32  qpos  1.8e-7   8.6e-7   1.6e-6   4.2e-2
#   HIP  qvel  6.6e-7   3.8e-6   7.5e-6   5.4e-1      oracle fp32  qvel  7.2e-7   4.3e-6   8.3e-6   3.3e-1
#   HIP  qacc  2.6e-6   1.7e-5   3.9e-5   1.7e+0      oracle fp32  qacc  2.9e-6   2.0e-5   4.7e-5   4.6e-1
# i.e. the kernel is as close to fp64 as the oracle's own fp32 instantiation, at every quantile. The table below is 2x the oracle's
# own fp32 spread for (median, p99, p99.9); the tests that have the fp64 oracle at hand (check_against_oracle_spread) compare with
# the spread measured in the same run instead of with these constants. The extreme value is set by a handful of env-steps that sit
# on a discrete switch of the solver (contact on/off, friction row saturating, Newton iteration cap: 0.02 % of env-steps, the same
# share in the oracle's fp32-vs-fp64 comparison) and is bounded relative to the oracle's own extreme value.
TOL = dict(qpos=(4e-7, 2e-6, 4e-6, 0.1), qvel=(1.5e-6, 1e-5, 2e-5, 1.0), qacc=(6e-6, 4e-5, 1e-4, 4.0))   # (median, p99, p99.9, max)


def emu_lib(solver: str = "reg", sanitize: bool = False) -> C.CDLL:
    """Host emulation build of the kernel body (tests/emu) — test infrastructure. solver = "reg": the product kernel's register-resident
    Newton solver, its wave primitives (DPP broadcasts, butterflies, lane swaps) emulated lane by lane (kbj_wave.h); "lds": the LDS
    formulation that `make ldssolver` builds for the GPU A/B test. sanitize: -fsanitize=address,undefined (load it in a child process
    with libasan preloaded, tests/test_sanitize.py)."""
    name = "libkbj_emu" + ("" if solver == "reg" else "_lds") + ("_asan" if sanitize else "") + ".so"
    out = os.path.join(ROOT, "tests", "emu", "_build", name)
    src = os.path.join(ROOT, "tests", "emu", "kbj_env_emu.cpp")
    deps = [src] + [os.path.join(ROOT, "kbot-joystick_amd", "csrc", f) for f in ("kbj_env_core.h", "kbj_env_phys.h", "kbj_env_task.h", "kbj_wave.h")]
    deps.append(os.path.join(ROOT, "include", "kbj_model.h"))
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else ["-O2"]
        if solver == "lds":
            flags.append("-DKBJ_ARROW_LDS")
        subprocess.check_call(["g++", *flags, "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                               "-shared", "-o", out, src])
    return C.CDLL(out) if not sanitize else out


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def state_errors(es_ref: np.ndarray, es_got: np.ndarray):
    """Per-env error measures between two [N][ES] state arrays."""
    d = np.abs(es_ref.astype(np.float64) - es_got.astype(np.float64))
    return dict(qpos=d[:, 0:27].max(1),
                qvel=d[:, 28:54].max(1) / (1 + np.abs(es_ref[:, 28:54]).max(1)),
                qacc=d[:, 54:80].max(1) / (1 + np.abs(es_ref[:, 54:80]).max(1)))


def check_error_distribution(errs: dict, tol=TOL, label=""):
    for k, (med, p99, p999, mx) in tol.items():
        v = np.concatenate(errs[k])
        assert np.median(v) <= med, f"{label}{k}: median {np.median(v):.3e} > {med}"
        assert np.quantile(v, 0.99) <= p99, f"{label}{k}: p99 {np.quantile(v, 0.99):.3e} > {p99}"
        if v.size >= 20000:     # a 99.9th percentile needs samples
            assert np.quantile(v, 0.999) <= p999, f"{label}{k}: p99.9 {np.quantile(v, 0.999):.3e} > {p999}"
        assert v.max() <= mx, f"{label}{k}: max {v.max():.3e} > {mx}"


def check_against_oracle_spread(err_hip: dict, err_o32: dict, switch: np.ndarray, label=""):
    """HIP-vs-fp64 error distribution against the oracle's own fp32-vs-fp64 distribution measured on the SAME env-steps:
      * median, p99, p99.9 at most 2x the oracle's (floored at a few fp32 roundings of the quantity);
      * no heavier tail: the count of env-steps beyond 2x the oracle's p99.9 is at most 1.5x the oracle's own count (+5);
      * the extreme value inside the absolute bound of the tolerance table (a single env-step on a discrete switch sets it, in
        the oracle's fp32-vs-fp64 comparison just the same: the ratio of two such extremes is not a stable statistic);
      * beyond 2x the oracle's p99.9, at most 8 env-steps that the oracle does not itself flag as sitting on a discrete switch of
        the solver (`switch`: active contacts / force-carrying rows / iteration counts differ between two evaluations that differ
        only by rounding, or the iteration cap bites) - measured: 2-3 of 98k, the kernel's own rounding flips a switch there."""
    floor = dict(qpos=2e-7, qvel=1e-6, qacc=4e-6)
    sw = np.concatenate(switch)
    for k in ("qpos", "qvel", "qacc"):
        h, o = np.concatenate(err_hip[k]), np.concatenate(err_o32[k])
        for name, q in (("median", 0.5), ("p99", 0.99), ("p99.9", 0.999)):
            hq, oq = np.quantile(h, q), np.quantile(o, q)
            assert hq <= 2 * max(oq, floor[k]), f"{label}{k} {name}: HIP {hq:.3e} vs oracle fp32 {oq:.3e}"
        thr = 2 * np.quantile(o, 0.999)
        nh, no = int((h > thr).sum()), int((o > thr).sum())
        assert nh <= 1.5 * no + 5, f"{label}{k}: {nh} env-steps beyond {thr:.2e}, the oracle's fp32 run has {no}"
        assert h.max() <= TOL[k][3], f"{label}{k}: max {h.max():.3e} (oracle fp32 max {o.max():.3e})"
        unexplained = int(((h > thr) & ~sw).sum())
        assert unexplained <= 8, f"{label}{k}: {unexplained} outliers beyond {thr:.2e} on env-steps without a discrete solver switch"


def random_actions(model, rng, n, scale=0.3):
    return (np.tile(np.array(model.joint_bias, np.float32), (n, 1)) + rng.normal(size=(n, 20)).astype(np.float32) * scale)
