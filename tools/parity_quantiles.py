"""Measured error distributions behind the parity tolerances (profiles/parity_rNN.json).

Teacher-forced control steps at the BASELINE env count: every step starts HIP, the fp32 oracle and the fp64 oracle from the SAME
fp32 state (the fp32 oracle's), with the same actions. Recorded per quantity (qpos, qvel, qacc - tests/helpers.state_errors):
    hip_vs_f64, oracle_f32_vs_f64   median / p99 / p99.9 / max
and, for the tail, how many env-steps exceed 2 x the oracle's own p99 / p99.9 and whether those steps are SENSITIVE ones: the fp64
oracle is re-run from a state perturbed by one fp32 rounding (relative 6e-8 on qpos / qvel) and its response is recorded - an env-step
whose fp64 result moves by more than a tenth of the HIP error under that perturbation amplifies rounding by construction (contact
switch, Huber-zone flip, Newton iteration cap), no arithmetic of any precision can agree on it.

usage: python3 tools/parity_quantiles.py [N=8192] [steps=12] [command=sampler|fixed] [robot=kbot-headless] [terrain=0] > profiles/parity_r02.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kbot_joystick_amd.host import binding as B
from kbot_joystick_amd.spec import compiler, layout as L
from oracle import oracle as O
from tests import helpers as H


def quantiles(v):
    v = np.asarray(v, np.float64)
    return dict(median=float(np.median(v)), p99=float(np.quantile(v, 0.99)), p999=float(np.quantile(v, 0.999)), max=float(v.max()))


def run(N, steps, command, robot, terrain, seed=11):
    model = compiler.load_model(robot)
    kw = {}
    if command == "fixed":
        kw.update(command_mode=1, fixed_command=[0.5] + [0.0] * 15)
    if terrain:
        kw.update(terrain_amp=0.05, terrain_wavelength=2.0)
    cfg = L.default_config(num_envs=N, batch_size=min(512, N), **kw)
    ctx = B.Context(model, cfg, 0, torch.cuda.current_stream().cuda_stream)
    dev = "cuda:0"
    z = lambda d: torch.zeros(N, d, device=dev)
    a, c, x = z(L.LD_ACTOR), z(L.LD_CRITIC), z(L.AUX["SIZE"])
    a2, c2, x2 = z(L.LD_ACTOR), z(L.LD_CRITIC), z(L.AUX["SIZE"])
    ctx.env_reset_all(seed, a, c, x)
    o32, o64, o64p = (O.Oracle(model, cfg, seed=seed, precision=p) for p in ("f32", "f64", "f64"))
    _, _, x0 = o32.reset_all()
    rng = np.random.default_rng(0)
    E = {k: {q: [] for q in ("qpos", "qvel", "qacc")} for k in ("hip", "o32", "pert")}
    SW = []
    obs = dict(actor=[], critic=[])
    A17 = {k: [] for k in ("comdist_hip", "comdist_o32", "touch_hip", "touch_o32")}     # next-observation com_distance (a17) and foot touch
    for t in range(steps):
        act = H.random_actions(model, rng, N)
        ep, es = o32.ep.copy(), o32.es.copy()
        ctx.env_set_state(ep, es)
        aux_t = torch.from_numpy(x0.copy()).cuda()
        aux32, aux64, aux64p = x0.copy(), x0.copy(), x0.copy()
        o64.ep[:], o64.es[:] = ep, es
        o64p.ep[:], o64p.es[:] = ep, es
        pert = 1 + 6e-8 * rng.choice([-1.0, 1.0], size=(N, 54))
        o64p.es[:, 0:27] *= pert[:, 0:27].astype(np.float32); o64p.es[:, 28:54] *= pert[:, 28:54].astype(np.float32)
        a32, c32, x0, d32 = o32.step_diag(act, aux32)
        a64, c64, x64, d64 = o64.step_diag(act, aux64)
        _, _, _, d64p = o64p.step_diag(act, aux64p)
        cap = cfg.solver_iterations
        # an env-step sits on a discrete switch when the solver's discrete state (active contacts, rows carrying force, saturated
        # friction rows, iteration counts) differs between two evaluations that differ only by rounding, or when the iteration cap bites
        switch = (d32 != d64).any(1) | (d64p != d64).any(1) | (d64[:, 0] >= cap) | (d32[:, 0] >= cap)
        ctx.env_step(torch.from_numpy(act).cuda(), aux_t, a2, c2, x2)
        ctx.synchronize()
        _, es_h = ctx.env_get_state()
        same = (aux32[:, L.AUX["DONE"]] == aux64[:, L.AUX["DONE"]]) & (aux32[:, L.AUX["DONE"]] == 0)   # resets re-draw the state: compare running envs
        for key, got in (("hip", es_h), ("o32", o32.es), ("pert", o64p.es)):
            for q, v in H.state_errors(o64.es, got).items():
                E[key][q].append(v[same])
        SW.append(switch[same])
        xh, cd, tc = x2.cpu().numpy(), L.AUX["COMDIST"], slice(L.AUX["TOUCH"], L.AUX["TOUCH"] + 2)
        A17["comdist_hip"].append(np.abs(xh[:, cd] - x64[:, cd])[same]); A17["comdist_o32"].append(np.abs(x0[:, cd] - x64[:, cd])[same])
        A17["touch_hip"].append(np.abs(xh[:, tc] - x64[:, tc]).max(1)[same]); A17["touch_o32"].append(np.abs(x0[:, tc] - x64[:, tc]).max(1)[same])
        obs["actor"].append(np.abs(a64 - a2.cpu().numpy()).max(1)[same])
        obs["critic"].append((np.abs(c64 - c2.cpu().numpy()) / (1 + np.abs(c64))).max(1)[same])
    out = dict(config=dict(N=N, steps=steps, command=command, robot=robot, terrain=bool(terrain), seed=seed), quantity={})
    for q in ("qpos", "qvel", "qacc"):
        hip, o32e, pe = (np.concatenate(E[k][q]) for k in ("hip", "o32", "pert"))
        qo = quantiles(o32e)
        sw = np.concatenate(SW)
        rec = dict(hip_vs_f64=quantiles(hip), oracle_f32_vs_f64=qo, f64_response_to_one_fp32_rounding=quantiles(pe), samples=int(hip.size),
                   discrete_switch_fraction=float(sw.mean()), hip_vs_f64_without_switch_steps=quantiles(hip[~sw]),
                   oracle_f32_vs_f64_without_switch_steps=quantiles(o32e[~sw]))
        for name in ("p99", "p999"):
            thr = 2 * qo[name]
            over = hip > thr
            sens = (pe > 0.1 * hip) | sw
            rec[f"over_2x_oracle_{name}"] = dict(threshold=thr, count=int(over.sum()), fraction=float(over.mean()),
                                                 oracle_f32_count=int((o32e > thr).sum()), explained_by_sensitivity=int((over & sens).sum()),
                                                 unexplained=int((over & ~sens).sum()))
        out["quantity"][q] = rec
    sw = np.concatenate(SW)
    for k, v in A17.items():
        v = np.concatenate(v)
        out[k + "_vs_f64"] = dict(quantiles(v), without_switch_steps=quantiles(v[~sw]), over_1e_3=int((v > 1e-3).sum()), over_1e_3_on_switch=int(((v > 1e-3) & sw).sum()))
    out["actor_obs_vs_f64"] = quantiles(np.concatenate(obs["actor"]))
    out["critic_obs_rel_vs_f64"] = quantiles(np.concatenate(obs["critic"]))
    ctx.close()
    return out


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    res = run(int(kw.get("N", 8192)), int(kw.get("steps", 12)), kw.get("command", "sampler"), kw.get("robot", "kbot-headless"), int(kw.get("terrain", 0)))
    json.dump(res, sys.stdout, indent=1)
    print()
