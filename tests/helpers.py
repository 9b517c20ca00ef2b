"""Shared helpers for the parity tests (env side)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Parity tolerances for ONE control step (5 substeps) started from the identical fp32 state.
# The dynamics have discrete events (contact on/off, Huber zones, Newton iteration cap), so a tiny share of
# env-steps amplifies rounding differences; the oracle's own fp32-vs-fp64 spread is the yardstick
# (measured: qpos median 1.8e-7, p99 2.4e-4, max 1.4e-2; see DESIGN.md "Parity").
TOL = dict(qpos=(2e-6, 2e-3, 0.1), qvel=(2e-5, 1e-2, 0.5), qacc=(1e-4, 3e-2, 2.0))   # (median, p99, max)


def emu_lib() -> C.CDLL:
    """Host emulation build of the kernel body (tests/emu) — test infrastructure."""
    out = os.path.join(ROOT, "tests", "emu", "_build", "libkbj_emu.so")
    src = os.path.join(ROOT, "tests", "emu", "kbj_env_emu.cpp")
    deps = [src] + [os.path.join(ROOT, "kbot-joystick_amd", "csrc", f) for f in ("kbj_env_core.h", "kbj_env_phys.h", "kbj_env_task.h")]
    deps.append(os.path.join(ROOT, "include", "kbj_model.h"))
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                               "-shared", "-o", out, src])
    return C.CDLL(out)


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def state_errors(es_ref: np.ndarray, es_got: np.ndarray):
    """Per-env error measures between two [N][ES] state arrays."""
    d = np.abs(es_ref.astype(np.float64) - es_got.astype(np.float64))
    return dict(qpos=d[:, 0:27].max(1),
                qvel=d[:, 28:54].max(1) / (1 + np.abs(es_ref[:, 28:54]).max(1)),
                qacc=d[:, 54:80].max(1) / (1 + np.abs(es_ref[:, 54:80]).max(1)))


def check_error_distribution(errs: dict, tol=TOL, label=""):
    for k, (med, p99, mx) in tol.items():
        v = np.concatenate(errs[k])
        assert np.median(v) <= med, f"{label}{k}: median {np.median(v):.3e} > {med}"
        assert np.quantile(v, 0.99) <= p99, f"{label}{k}: p99 {np.quantile(v, 0.99):.3e} > {p99}"
        assert v.max() <= mx, f"{label}{k}: max {v.max():.3e} > {mx}"


def random_actions(model, rng, n, scale=0.3):
    return (np.tile(np.array(model.joint_bias, np.float32), (n, 1)) + rng.normal(size=(n, 20)).astype(np.float32) * scale)
