"""A/B of whole training iterations between builds / switches, alternating child processes on one box (box-to-box spread is ~1 %, run-to-run
on one box ~0.2 %: only alternating runs on the same box separate a 0.5 % effect).

usage: python tools/ab_bench.py [--rounds 3] [--iters 8] [--update-only] NAME=ENV1=V1,ENV2=V2 ... (a bare NAME= is the default build / switches)
  e.g. python tools/ab_bench.py base= critic_first=KBJ_CRITIC_FIRST=1 libB=KBJ_LIB_NAME=libkbj_b.so
Each leg: configs[1] (8192 envs, kbot-headless, fixed command), 2 warm-up iterations, then `iters` timed ones; prints ms per iteration per leg
and the per-configuration mean / min. --update-only times task.update() alone (the rollout is run once, untimed)."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, json
sys.path.insert(0, %r)
import torch
from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
task = HumanoidWalkingTask(launch_config(num_envs=8192, robot="kbot-headless", fixed_command=(0.5, 0.0, 0.0), seed=0))
iters, update_only = %d, %d
from kbot_joystick_amd.host.binding import KbjError
def it(fn):          # (timing experiments with deliberately wrong kernels trip the library's fail-stop checks: the time is still the time)
    try: fn()
    except KbjError as e:
        if not getattr(it, "warned", False): print("AB_WARN", str(e)[:120], flush=True); it.warned = True
for _ in range(2): it(task.train_iteration)
torch.cuda.synchronize()
if update_only:
    it(task.rollout); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): it(task.update)
else:
    t0 = time.perf_counter()
    for _ in range(iters): it(task.train_iteration)
torch.cuda.synchronize()
print("AB_MS", (time.perf_counter() - t0) / iters * 1e3, flush=True)
task.close()
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--update-only", action="store_true")
    ap.add_argument("legs", nargs="+")
    a = ap.parse_args()
    legs = []
    for spec in a.legs:
        name, _, envs = spec.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        legs.append((name, env))
    res = {name: [] for name, _ in legs}
    for r in range(a.rounds):
        for name, env in legs:
            e = {k: v for k, v in os.environ.items() if not k.startswith("KBJ_")}
            e.update(env)
            out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, a.iters, int(a.update_only))], capture_output=True, text=True, env=e, timeout=900)
            ms = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("AB_MS")]
            if out.returncode != 0 or not ms:
                print(f"[{name}] FAILED rc={out.returncode}: {out.stderr[-800:]}", flush=True)
                continue
            res[name].append(ms[0])
            print(f"round {r} {name:24s} {ms[0]:8.2f} ms", flush=True)
    print(json.dumps({k: {"runs": [round(x, 2) for x in v], "mean": round(sum(v) / len(v), 2) if v else None, "min": round(min(v), 2) if v else None} for k, v in res.items()}))


if __name__ == "__main__":
    main()
