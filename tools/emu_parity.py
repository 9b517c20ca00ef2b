"""CPU-only counterpart of tools/parity_quantiles.py for the HOST EMULATION of the env kernel body (tests/emu): teacher-forced control
steps; the emulated register solver ("reg"), the LDS formulation ("lds") and the fp32 oracle are all stepped from the fp32 oracle's
state and compared with the fp64 oracle on the same env-steps. Says - before a GPU is involved - whether a change to the solver's
arithmetic (reciprocals, reduction order, Huber clamp) moves the error quantiles the GPU parity tests bound.
usage: python3 tools/emu_parity.py [N=1024] [steps=12] [robot=kbot-headless] [terrain=0]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from kbot_joystick_amd.spec import compiler, layout as L
from oracle import oracle as O
from tests import helpers as H


def q(v):
    v = np.asarray(v, np.float64)
    return dict(median=float(np.median(v)), p99=float(np.quantile(v, 0.99)), p999=float(np.quantile(v, 0.999)), max=float(v.max()))


def main():
    kw = dict(a.split("=") for a in sys.argv[1:])
    N, steps, robot, terrain = int(kw.get("N", 1024)), int(kw.get("steps", 12)), kw.get("robot", "kbot-headless"), int(kw.get("terrain", 0))
    model = compiler.load_model(robot)
    cfg = L.default_config(num_envs=N, batch_size=min(512, N), **(dict(terrain_amp=0.05, terrain_wavelength=2.0) if terrain else {}))
    libs = {k: H.emu_lib(k) for k in ("reg", "lds")}
    if os.environ.get("KBJ_EMU_ALT_LIB"):      # an experimental build of the emulation beside the two standard ones (e.g. -DKBJ_SOLVER_RAW_RCP -DKBJ_EMU_RCP_1ULP)
        libs["alt"] = C.CDLL(os.environ["KBJ_EMU_ALT_LIB"])
    o32, o64 = (O.Oracle(model, cfg, seed=11, precision=p) for p in ("f32", "f64"))
    _, _, x0 = o32.reset_all()
    a1, c1, x1 = o32.new_obs()
    rng = np.random.default_rng(0)
    E = {k: {n: [] for n in ("qpos", "qvel", "qacc")} for k in list(libs) + ["o32"]}
    done_mismatch = {k: 0 for k in libs}
    for t in range(steps):
        act = H.random_actions(model, rng, N)
        ep, es = o32.ep.copy(), o32.es.copy()
        o64.ep[:], o64.es[:] = ep, es
        aux32, aux64 = x0.copy(), x0.copy()
        got = {}
        for k, lib in libs.items():
            e, s, auxe = ep.copy(), es.copy(), x0.copy()
            lib.kbj_emu_env_step(C.byref(model), C.byref(cfg), C.c_uint32(11), H.fptr(e), H.fptr(s), H.fptr(act), H.fptr(auxe), H.fptr(a1), H.fptr(c1), H.fptr(x1))
            got[k] = (s, auxe)
        _, _, x0 = o32.step(act, aux32)
        o64.step(act, aux64)
        same = (aux32[:, L.AUX["DONE"]] == aux64[:, L.AUX["DONE"]]) & (aux32[:, L.AUX["DONE"]] == 0)
        for k in libs:
            done_mismatch[k] += int((got[k][1][:, L.AUX["DONE"]] != aux32[:, L.AUX["DONE"]]).sum())
            for n, v in H.state_errors(o64.es, got[k][0]).items():
                E[k][n].append(v[same])
        for n, v in H.state_errors(o64.es, o32.es).items():
            E["o32"][n].append(v[same])
    out = dict(config=dict(N=N, steps=steps, robot=robot, terrain=bool(terrain)), done_mismatch_vs_oracle_f32=done_mismatch)
    for n in ("qpos", "qvel", "qacc"):
        out[n] = {k: q(np.concatenate(E[k][n])) for k in E}
        thr = 2 * out[n]["o32"]["p999"]
        out[n]["count_over_2x_oracle_p999"] = {k: int((np.concatenate(E[k][n]) > thr).sum()) for k in E}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
