#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the full K-Bot joystick training iteration (rollout + PPO update).

  python bench.py --gpus N --steps K --warmup W
For N > 1 launch with:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one training iteration over 8192 envs per GPU (BASELINE.json configs[1]; weak scaling): a 100-control-step
rollout (policy forward, 5 physics substeps, observations, terminations/resets, rewards) followed by GAE and
3 passes x 16 minibatches of BPTT + AdamW (launch hyper-parameters train.py:1761-1791, hidden 256).
Rank 0 prints ONE JSON line; `roofline` and `cpu_baseline` are measured in the same run (N = 1 for the CPU leg).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 8192
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
ENV_BYTES_PER_ENVSTEP = 13.3e3  # state r/w + per-env params + trajectory rows


def nn_flops_per_envstep(H: int, num_passes: int) -> float:
    """Algorithmic matmul FLOPs per env-step: one forward at rollout (actor + critic) and, per pass,
    forward + backward (2x forward) through the same weights."""
    lstm = 2 * (4 * H * 2 * H)
    actor = 65 * H + lstm + H * 40
    critic = 475 * H + lstm + H
    fwd = 2.0 * (actor + critic)
    return fwd * (1 + 3 * num_passes)


def cpu_baseline(seconds_budget: float = 25.0):
    """The CPU oracle (C++ env + torch actor-critic, a *port*: the JAX reference cannot run offline) on a bounded sample
    of the same workload: a scaled-down full iteration (64 envs x 100 steps, batch 64, 3 passes, hidden 256)."""
    import numpy as np
    import torch
    from kbot_joystick_amd.spec import compiler, layout as L
    from oracle import nn as ON
    from oracle.trainer import OracleTrainer
    from oracle import oracle as O
    n_envs = 64
    cfg = L.default_config(num_envs=n_envs, batch_size=n_envs, rollout_len=100, hidden_size=256, num_passes=3, command_mode=1)
    cfg.fixed_command[0] = 0.5
    model = compiler.load_model("kbot-headless")
    rng = np.random.default_rng(0)
    P = ON.param_count(256)
    params = (rng.uniform(-1, 1, P) / 16).astype(np.float32)
    tr = OracleTrainer(model, cfg, seed=0, params=params)
    tr._normal = lambda step: rng.standard_normal((n_envs, 20)).astype(np.float32)   # python threefry loop is not the thing timed
    t0 = time.time()
    tr.train_iteration()
    dt = time.time() - t0
    cores = max(int(O.lib().kbj_cpu_num_threads()), torch.get_num_threads())
    return dict(value=n_envs * 100 / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"oracle full iteration on {n_envs} envs x 100 steps, batch {n_envs}, 3 passes, hidden 256 ({dt:.1f} s wall)")


def pmc_traffic() -> dict:
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/rocprof_summary.py --pmc with the gfx950 FETCH_SIZE correction); {} when no PMC profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {k: v.get("hbm_bytes_per_launch") for k, v in json.load(f).items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=1, choices=[1, 3, 4],
                    help="BASELINE.json configs[i] workload: 1 = the metric's (kbot-headless, flat, fixed command); 3 = UnifiedCommand "
                         "sampler; 4 = full kbot on the sine terrain (extra measurements, not the headline line)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    wl = {1: dict(robot="kbot-headless", fixed_command=(0.5, 0.0, 0.0)),
          3: dict(robot="kbot-headless"),
          4: dict(robot="kbot", terrain="sine")}[args.config]
    wl_name = {1: "kbot-headless, {n} envs/GPU, flat ground, fixed joystick command (0.5,0,0)",
               3: "kbot-headless, {n} envs/GPU, flat ground, UnifiedCommand 6-mode sampler",
               4: "kbot (full), {n} envs/GPU, sine terrain (A 0.05 m, L 2 m), UnifiedCommand sampler, pushes + all randomizers"}[args.config]
    cfg = launch_config(num_envs=args.envs_per_gpu * world, hidden_size=args.hidden, seed=0, **wl)
    task = HumanoidWalkingTask(cfg, device=torch.device("cuda", local_rank), rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        task.train_iteration()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        task.train_iteration()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    env_steps = args.envs_per_gpu * world * task.T * args.steps
    value = env_steps / elapsed

    # ---- roofline leg: one extra iteration with HIP-event timing inside the library (rank 0) ----
    roofline = roofline2 = None
    if rank == 0:
        task.ctx.profile_begin()
    task.train_iteration()   # every rank takes part (gradient all-reduce); only rank 0 is instrumented
    torch.cuda.synchronize()
    if rank == 0:
        prof = task.ctx.profile_end()
        nn_s = prof["nn_ms"] * 1e-3
        env_s = prof["env_step_ms"] * 1e-3
        steps_gpu = args.envs_per_gpu * task.T
        flops = nn_flops_per_envstep(args.hidden, cfg.num_passes) * steps_gpu
        iter_s = elapsed / args.steps
        ach_tf = flops / iter_s / 1e12       # the env kernel overlaps the policy GEMMs, so the NN path is priced against the whole iteration
        nn_roof = dict(bound="mfma", kernel="whole iteration: all fp32-MFMA work (policy steps + PPO update) / iteration wall time", achieved=round(ach_tf, 3),
                       peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach_tf / PEAK_FP32_MFMA_TFLOPS, 4), traffic=None,
                       iteration_ms=round(iter_s * 1e3, 2), ppo_grad_ms_per_iteration=round(prof["nn_ms"], 2), ppo_grad_calls=prof["nn_launches"])
        per_launch = env_s / max(prof["env_step_launches"], 1)
        envs_per_launch = args.envs_per_gpu * task.T / max(prof["env_step_launches"], 1)
        ach_gbs = envs_per_launch * ENV_BYTES_PER_ENVSTEP / per_launch / 1e9
        traffic = pmc_traffic()
        env_roof = dict(bound="hbm", kernel="env_step_kernel", achieved=round(ach_gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(ach_gbs / PEAK_HBM_GBS, 5), traffic=traffic.get("env_step_kernel"), avg_launch_us=round(per_launch * 1e6, 1),
                        launches=prof["env_step_launches"], envs_per_launch=int(envs_per_launch), total_ms=round(prof["env_step_ms"], 2),
                        note="latency/issue-bound per-env solver; HBM is not the limiter (DESIGN.md)")
        kernels = [env_roof]
        for k in prof["kernels"]:      # every launch bracketed by HIP events on its own stream inside libkbj.so
            if k["name"] == "env_step_kernel":
                continue
            tf = k["flops"] / (k["total_ms"] * 1e-3) / 1e12
            kernels.append(dict(bound="mfma", kernel=k["name"], achieved=round(tf, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                                frac=round(tf / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic.get(k["name"]),
                                avg_launch_us=round(k["total_ms"] * 1e3 / k["launches"], 1), launches=k["launches"], total_ms=round(k["total_ms"], 2)))
        kernels.sort(key=lambda r: -r["total_ms"])
        roofline = kernels[0]          # the dominant kernel by summed launch time
        roofline2 = dict(path=nn_roof, kernels=kernels[1:])

    if world > 1:
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    out = {
        "metric": "env-steps/sec (whole node), K-Bot joystick 8192 envs @1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl_name.format(n=args.envs_per_gpu) + ", full iteration: "
                               f"100-step rollout + PPO update (batch 512/GPU, 3 passes, LSTM hidden {args.hidden}, depth 2)",
                   "envs_per_gpu": args.envs_per_gpu, "rollout_steps": task.T, "batch_size_per_gpu": cfg.batch_size, "num_passes": cfg.num_passes,
                   "hidden_size": args.hidden, "baseline_config": args.config, "parallelism": f"env-sharded dp{world}, grad all-reduce per optimizer step"},
        "roofline": roofline, "roofline_secondary": roofline2, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
