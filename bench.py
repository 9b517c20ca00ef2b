#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the full K-Bot joystick training iteration (rollout + PPO update).

  python bench.py --gpus N --steps K --warmup W
For N > 1 either launch it under torchrun (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
... bench.py --gpus N ...) or call it directly: it then starts its N ranks itself (a torch.distributed.run child process,
before this process touches the GPU) and exits with the child's status. Ranks talk over RCCL (backend "nccl").

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
PATH_BYTES_PER_ENVSTEP = 169e3  # rollout 13.3 KB + update 156 KB (trajectory re-reads + BPTT stash, 3 passes) per env-step


def nn_flops_per_envstep(H: int, num_passes: int) -> float:
    """Algorithmic matmul FLOPs per env-step: one forward at rollout (actor + critic) and, per pass,
    forward + backward (2x forward) through the same weights."""
    lstm = 2 * (4 * H * 2 * H)
    actor = 65 * H + lstm + H * 40
    critic = 475 * H + lstm + H
    fwd = 2.0 * (actor + critic)
    return fwd * (1 + 3 * num_passes)


def cpu_baseline(repeats: int = 3):
    """The CPU oracle (C++ OpenMP env + torch actor-critic, a *port*: the JAX reference cannot run offline) on bounded samples of the
    same workload, full iterations (rollout + GAE + 3 passes, hidden 256) of configs[1]:
      * LARGE: 2048 envs x 100 steps, batch 512 - a problem big enough for the host's cores (the env part is one OpenMP task per env, the
        minibatch matrices are 512 rows). Thread sweep (16 ... usable cores) on a 2048-env x 20-step probe, then ONE timed full iteration
        at the best setting. This is the line's `value`: the best the host does on this path.
      * SMALL: 256 envs x 100 steps, batch 256 (round 3's sample: small-problem overhead, best at 16 threads), median of `repeats`.
    plus the configs[0] plumbing line (4 envs x 64 steps, batch 4). `cores` = the threads used for `value`."""
    import numpy as np
    import torch
    from kbot_joystick_amd.spec import compiler, layout as L
    from oracle import nn as ON
    from oracle.trainer import OracleTrainer
    from oracle import oracle as O
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # The CPU time this process may actually use: a container's cgroup quota is NOT visible in the affinity mask. Round 5's sweep on the GPU
    # box ran 16 / 32 threads at 4.46 / 8.13 s per probe: the box gives a one-GPU job the CPU time of about 16 cores (cpu.max), so 32 busy
    # threads are 2x oversubscribed - OpenMP's and torch's workers spin at their barriers while their partners wait for a time slice. The
    # sweep therefore stops at the quota (when one is set), and the line reports it.
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            quota = parse(open(path).read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    budget_cores = usable if quota is None else max(1, min(usable, int(quota + 0.5)))
    model = compiler.load_model("kbot-headless")

    def set_threads(n):
        torch.set_num_threads(n)
        O.lib().kbj_cpu_set_num_threads(n)

    def make(n_envs, T, batch=None):
        cfg = L.default_config(num_envs=n_envs, batch_size=batch or n_envs, rollout_len=T, hidden_size=256, num_passes=3, command_mode=1)
        cfg.fixed_command[0] = 0.5
        rng = np.random.default_rng(0)
        params = (rng.uniform(-1, 1, ON.param_count(256)) / 16).astype(np.float32)
        tr = OracleTrainer(model, cfg, seed=0, params=params)
        tr._normal = lambda step: rng.standard_normal((n_envs, 20)).astype(np.float32)   # python threefry loop is not the thing timed
        return tr

    def timed(tr):
        t0 = time.perf_counter()
        tr.train_iteration()
        return time.perf_counter() - t0

    def sweep_threads(probe, candidates, label, budget_s):
        """smallest first; stops at the first setting clearly slower than the best so far (oversubscribed pools get slower) or when the
        time budget is spent; every point is printed as it is measured (a silent sweep looks hung)"""
        sweep, spent = {}, 0.0
        for n in candidates:
            if n > budget_cores and sweep:
                break
            set_threads(n)
            if not sweep:
                probe.train_iteration()   # warm-up (allocator, thread pools)
            sweep[n] = round(timed(probe), 3)
            spent += sweep[n]
            print(f"bench.py: cpu_baseline thread sweep ({label}): {n} threads -> {sweep[n]:.3f} s per probe iteration", file=sys.stderr, flush=True)
            if sweep[n] > 1.3 * min(sweep.values()) or spent > budget_s:
                break
        return sweep

    # ---- small sample (round 3's): 256 envs ----
    sweep_s = sweep_threads(make(128, 50), (8, 16, 32, 64, 128, 256), "128 envs x 50 steps", 30.0)
    best_s = min(sweep_s, key=sweep_s.get)
    set_threads(best_s)
    tr = make(256, 100)
    tr.train_iteration()              # warm-up at the chosen setting
    ts1 = [timed(tr) for _ in range(repeats)]
    t1 = float(np.median(ts1))
    del tr
    # ---- large sample: 2048 envs, batch 512 ----
    sweep_l = sweep_threads(make(2048, 20, 512), (8, 16, 32, 64, 128, 256), "2048 envs x 20 steps, batch 512", 60.0)
    best_l = min(sweep_l, key=sweep_l.get)
    set_threads(best_l)
    T_large = 50                      # 2048 envs x 50 steps per timed iteration: ~10 s each at the round-5 rate, three of them = the bounded sample
    trl = make(2048, T_large, 512)
    trl.train_iteration()             # warm-up at the chosen setting (its own buffers)
    tsl = [timed(trl) for _ in range(repeats)]
    t_large = float(np.median(tsl))
    print(f"bench.py: cpu_baseline large sample: 2048 envs x {T_large} steps at {best_l} threads -> median of {repeats} = {t_large:.2f} s "
          f"(runs {', '.join('%.2f' % t for t in tsl)})", file=sys.stderr, flush=True)
    del trl
    set_threads(best_s)
    tr0 = make(4, 64)
    tr0.train_iteration()
    t0 = float(np.median([timed(tr0) for _ in range(3)]))
    return dict(value=2048 * T_large / t_large, unit="env-steps/s", cores=best_l, kind="port", host_cores=os.cpu_count(), usable_cores=usable,
                cgroup_cpu_quota_cores=quota, samples=repeats,
                sample=f"oracle full iteration (rollout + GAE + 3 passes) on 2048 envs x {T_large} steps, batch 512, hidden 256: median of {repeats} iterations = "
                       f"{t_large:.2f} s (runs {', '.join('%.2f' % t for t in tsl)}) at the best of the thread sweep ({best_l} threads; the affinity mask shows "
                       f"{usable} of the host's {os.cpu_count()} cores, the cgroup CPU quota is {'none' if quota is None else '%.1f cores' % quota}: the sweep stops "
                       f"at the quota, more busy threads than that only oversubscribe it - round 5's 32-thread point was 1.8x slower than 16 for that reason)",
                thread_sweep_probe="2048 envs x 20 steps, batch 512: seconds per iteration by thread count (stops at the first setting > 1.3x the best or after 60 s)",
                thread_sweep_seconds_per_iteration={str(k): v for k, v in sweep_l.items()},
                small_sample=dict(value=256 * 100 / t1, unit="env-steps/s", cores=best_s,
                                  sample=f"256 envs x 100 steps, batch 256: median of {repeats} = {t1:.2f} s (runs {', '.join('%.2f' % t for t in ts1)}) at {best_s} threads",
                                  thread_sweep_seconds_per_iteration={str(k): v for k, v in sweep_s.items()}),
                config0=dict(value=4 * 64 / t0, unit="env-steps/s", sample=f"configs[0]: 4 envs x 64 steps, batch 4, 3 passes, hidden 256: median of 3 = {t0:.2f} s"))


def source_fingerprint() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): ties a committed PMC profile to the code it measured."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "kbot-joystick_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files += sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic() -> tuple[dict, dict]:
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_traffic.py with the gfx950 FETCH_SIZE correction) and the profile's provenance. The bytes are only reported
    when the profile was taken on the kernel sources this run was built from (source fingerprint); otherwise traffic is null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return {}, dict(status="no committed PMC profile")
    with open(path) as f:
        doc = json.load(f)
    meta = doc.pop("_meta", {})
    fp = source_fingerprint()
    if meta.get("source_fingerprint") != fp:
        print(f"bench.py: profiles/pmc_traffic.json was taken on other kernel sources ({meta.get('source_fingerprint')} != {fp}): "
              "roofline.traffic = null (re-run tools/pmc_traffic.py)", file=sys.stderr)
        return {}, dict(status="stale", profile_fingerprint=meta.get("source_fingerprint"), source_fingerprint=fp, git=meta.get("git"))
    iters = max(1, int(meta.get("iterations", 3)))   # training iterations the profiled command ran (warm-up + steps + the roofline leg)
    total = sum(v.get("hbm_bytes_per_launch", 0) * v.get("launches", 0) for v in doc.values())
    return {k: v.get("hbm_bytes_per_launch") for k, v in doc.items()}, dict(status="current", source_fingerprint=fp, git=meta.get("git"),
                                                                             hbm_bytes_per_iteration=total / iters, profiled_iterations=iters)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a torch.distributed.run child BEFORE this process
    makes any GPU call (never exec from a GPU-initialised process) and hand back the child's exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (RCCL across processes)
    return subprocess.call(cmd, env=env)


def run_leg_in_child(leg: str, args, timeout_s: float) -> dict:
    """A non-headline leg (`--leg variant` / `--leg train_loop`) in a FRESH child process of this one: a hard fault, abort or hang in it can
    cost its own object of the line, never the headline that is already measured (a try / except only catches Python exceptions). The child is
    started with subprocess (fork + exec of a new interpreter - never an exec of this GPU-initialised process) and prints one JSON object."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", leg, "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--envs-per-gpu", str(args.envs_per_gpu), "--hidden", str(args.hidden), "--config", str(args.config), "--allreduce", args.allreduce,
           "--loop-iterations", str(args.loop_iterations)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return dict(value=None, error=f"leg {leg}: no result after {timeout_s:.0f} s (child killed)")
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except json.JSONDecodeError:
                break
    return dict(value=None, error=f"leg {leg}: child exited with {out.returncode}: {out.stderr.strip()[-300:]}")


def leg_variant(args, cfg_kw, wl) -> dict:
    """Child process: the headline workload with kbj_config.gemm_bf16x3, timed like the headline."""
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    task = HumanoidWalkingTask(launch_config(gemm_bf16x3=True, **cfg_kw, **wl), device=torch.device("cuda", 0))
    for _ in range(max(1, min(args.warmup, 2))):
        task.train_iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        task.train_iteration()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = dict(name="gemm_bf16x3", NOT_THE_HEADLINE=True, value=round(args.envs_per_gpu * task.T * args.steps / el, 1), unit="env-steps/s",
               ms_per_step=round(el / args.steps * 1e3, 2), process="child of bench.py (own context, nothing else on the GPU)",
               what="kbj_config.gemm_bf16x3 = 1: input-gradient and weight-gradient GEMMs of the PPO update as 6 bf16 MFMA products per fp32 product "
                    "(operands split exactly into three bf16 pieces, fp32 accumulation); everything else identical. Error study: DESIGN.md section 10b, "
                    "parity: tests/test_gpu_switches.py")
    task.close()
    return out


def leg_train_loop(args, cfg_kw, wl) -> dict:
    """Child process: HumanoidWalkingTask.launch() as a user of the reference runs it (train.py:1759-1792) - scalar logging every iteration (CSV +
    TensorBoard event file), deterministic validation rollouts, ckpt.bin rewritten on a timer - on the headline workload. The reference's
    block validates every 100 iterations and saves every 60 s; over the bounded length of this leg that would be no validation at all, so
    the leg validates every 25 iterations and saves every 10 s (both MORE often than the reference: a conservative figure)."""
    import tempfile
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    n = max(1, args.loop_iterations)
    with tempfile.TemporaryDirectory(prefix="kbj_train_loop_") as run_dir:
        cfg = launch_config(valid_every_n_steps=25, save_every_n_seconds=10, **cfg_kw, **wl)
        task = HumanoidWalkingTask.launch(cfg, num_iterations=n, run_dir=run_dir, quiet=True)
        st = dict(task.loop_stats)
        ck = os.path.getsize(os.path.join(run_dir, "checkpoints", "ckpt.bin"))
        task.close()
    steady = st.get("steady_env_steps_per_s", st["env_steps_per_s"])
    return dict(name="train_loop", value=round(steady, 1), unit="env-steps/s", iterations=st["iterations"],
                value_including_first_iteration=round(st["env_steps_per_s"], 1), first_iteration_seconds=round(st.get("first_iteration_seconds", 0.0), 3),
                ms_per_iteration=round((st["loop_seconds"] - st.get("first_iteration_seconds", 0.0)) / max(st["iterations"] - 1, 1) * 1e3, 2), validations=st["validations"],
                validation_seconds=round(st["validation_seconds"], 3), checkpoints_in_loop=st["checkpoints"],
                checkpoint_seconds_in_line=round(st["checkpoint_seconds"], 3), final_checkpoint_seconds=round(st.get("final_checkpoint_seconds", 0.0), 3),
                checkpoint_bytes=ck, process="child of bench.py (own context, nothing else on the GPU)",
                what="HumanoidWalkingTask.launch(): every iteration followed by task.scalars() + CSV / TensorBoard logging; validate() (64 envs x "
                     "render_length_seconds, argmax actions) every 25 iterations; ckpt.bin every 10 s (device arrays snapshotted in line, container "
                     "written by a background thread). `value` = iterations 2..n (wall time from the end of the first iteration - which loads every "
                     "kernel's code object and allocates the staging buffers, as the headline's warm-up steps do outside ITS timed region - to the end of "
                     "the last, validations / checkpoints / logging inside); `value_including_first_iteration` = all n")


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
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the gradient all-reduce (nccl = RCCL over xGMI)")
    ap.add_argument("--share-gpu", action="store_true", help="diagnostic: every rank on GPU 0 (rehearses the N-rank flow on a 1-GPU box; use with --backend gloo)")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1 only: what the RCCL leg costs THIS schedule on one GPU. The same task is timed three times in one process: without a "
                         "process group, then with backend nccl at world size 1 and the gradient all-reduce forced (RCCL's stream and kernels join the "
                         "context's lanes), then with the overlapped actor-slice exchange; the line's value is the forced per-step run, the object "
                         "`forced_collective` holds all three")
    ap.add_argument("--gemm-bf16x3", action="store_true",
                    help="NOT the headline: kbj_config.gemm_bf16x3 - the update's large backward GEMMs on the bf16 matrix cores through the exact three-way "
                         "split of their fp32 operands (DESIGN.md section 10b). The line says so in `variant`, `dtype` and `config.workload`")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra, non-headline legs (the gemm_bf16x3 variant and the launch()-based training loop, each run in a child process after the headline is measured)")
    ap.add_argument("--leg", default=None, choices=["variant", "train_loop"], help="internal: run ONE non-headline leg in this (child) process and print its JSON object")
    ap.add_argument("--loop-iterations", type=int, default=64, help="length of the train_loop leg (launch() with validation / checkpoints / logging)")
    ap.add_argument("--allreduce", default=os.environ.get("KBJ_ALLREDUCE", "per_step"), choices=["per_step", "per_pass"],
                    help="per_step (default): all-reduce before every optimizer step; per_pass: accumulate a pass, one all-reduce + one step per pass")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))          # nothing has touched the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}")
    import torch
    if not torch.cuda.is_available():
        sys.exit(f"bench.py needs a HIP device (rank {rank} of {world})")
    if args.share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    wl = {1: dict(robot="kbot-headless", fixed_command=(0.5, 0.0, 0.0)),
          3: dict(robot="kbot-headless"),
          4: dict(robot="kbot", terrain="sine")}[args.config]
    wl_name = {1: "kbot-headless, {n} envs/GPU, flat ground, fixed joystick command (0.5,0,0)",
               3: "kbot-headless, {n} envs/GPU, flat ground, UnifiedCommand 6-mode sampler",
               4: "kbot (full), {n} envs/GPU, sine terrain (A 0.05 m, L 2 m), UnifiedCommand sampler, pushes + all randomizers"}[args.config]
    cfg_kw = dict(num_envs=args.envs_per_gpu * world, hidden_size=args.hidden, seed=0, allreduce=args.allreduce)
    if args.leg:       # child process of a one-GPU bench.py: one non-headline leg, one JSON object
        if world != 1:
            sys.exit("bench.py: --leg is the one-GPU child of bench.py")
        print(json.dumps(leg_variant(args, cfg_kw, wl) if args.leg == "variant" else leg_train_loop(args, cfg_kw, wl)), flush=True)
        return
    cfg = launch_config(gemm_bf16x3=args.gemm_bf16x3, **cfg_kw, **wl)
    task = HumanoidWalkingTask(cfg, device=torch.device("cuda", local_rank), rank=rank, world_size=world)
    if args.share_gpu and world > 1:
        # the persistent LSTM recurrences need their whole grid resident; kbj_create checks that per context, but ranks that SHARE a GPU share
        # its CUs. The numbers come from the library (kbj_recurrence_residency), not from a formula restated here. Half the slots: the ranks'
        # GEMM and env workgroups want room too (measured: 4 ranks x hidden 64 on 256 CUs could not place their grids, gpurun_out/r05r; since
        # round 6 that case fails within the 2 s wait bound instead of crawling, but refusing is friendlier).
        grid, conc, slots = task.ctx.recurrence_residency()
        if world * conc * grid > slots // 2:
            sys.exit(f"bench.py --share-gpu: {world} ranks x {conc} concurrent recurrence launches x {grid} workgroups (batch {cfg.batch_size}, hidden {args.hidden}) "
                     f"do not fit half of the {slots} resident workgroup slots of the one GPU they share: use a smaller --hidden or at most {max(1, slots // (2 * conc * grid))} ranks")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(warmup, steps):
        for _ in range(warmup):
            task.train_iteration()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            task.train_iteration()
        barrier()
        return time.perf_counter() - t0

    from kbot_joystick_amd.host import dist as dist_util
    forced = None
    if args.force_collective:
        if world != 1:
            sys.exit("bench.py: --force-collective is the one-GPU measurement (--gpus 1)")
        import torch.distributed as dist
        plain = timed_steps(args.warmup, args.steps)                       # the schedule without a process group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        dist_util.FORCE_COLLECTIVE = True                                  # world size 1 would skip the collective
        forced = dict(ms_per_step_no_process_group=round(plain / args.steps * 1e3, 2))
    elapsed = timed_steps(args.warmup, args.steps)
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        ts = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ts, torch.tensor([elapsed], device="cuda", dtype=torch.float64))
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in ts]
        elapsed = max(float(t.item()) for t in ts)                          # MAX over ranks
    # the rank count the line reports comes from the process group itself: its size and an all-reduce of ones through the backend
    pg_ranks, pg_counted, pg_backend = 1, 1, None
    if world > 1 or args.force_collective:
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        pg_ranks, pg_counted, pg_backend = dist.get_world_size(), int(round(float(ones.item()))), dist.get_backend()
    env_steps = args.envs_per_gpu * world * task.T * args.steps
    value = env_steps / elapsed

    # ---- roofline leg: one extra iteration with HIP-event timing inside the library (rank 0) ----
    roofline = roofline2 = hbm = None
    if rank == 0:
        task.ctx.profile_begin()
        dist_util.TIMING = []          # events around every gradient all-reduce of the instrumented iteration
    task.train_iteration()   # every rank takes part (gradient all-reduce); only rank 0 is instrumented
    torch.cuda.synchronize()
    allreduce_ms, allreduce_calls = 0.0, 0
    if rank == 0:
        allreduce_calls = len(dist_util.TIMING)
        allreduce_ms = float(sum(a.elapsed_time(b) for a, b in dist_util.TIMING))
        dist_util.TIMING = None
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
        traffic, traffic_src = pmc_traffic()
        env_roof = dict(bound="hbm", kernel="env_step_kernel", achieved=round(ach_gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(ach_gbs / PEAK_HBM_GBS, 5), traffic=traffic.get("env_step_kernel"), avg_launch_us=round(per_launch * 1e6, 1),
                        launches=prof["env_step_launches"], envs_per_launch=int(envs_per_launch), total_ms=round(prof["env_step_ms"], 2),
                        note="vector-issue bound (wave64 VALU instruction = 4 SIMD cycles; valu_issue_utilisation in profiles/*_pmc_env_step.json); HBM is not the limiter (DESIGN.md section 5)")
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
        roofline["traffic_profile"] = traffic_src
        roofline2 = dict(path=nn_roof, kernels=kernels[1:])
        # whole-path HBM: counter bytes of every kernel of an iteration (committed PMC profile of the same kernel sources) against the
        # iteration time, beside the algorithmic figure of SURVEY.md section 8d (169 KB per env-step)
        alg = PATH_BYTES_PER_ENVSTEP * steps_gpu / iter_s / 1e9
        hbm = dict(peak=PEAK_HBM_GBS, unit="GB/s", algorithmic=round(alg, 1), algorithmic_frac=round(alg / PEAK_HBM_GBS, 4),
                   algorithmic_bytes_per_env_step=PATH_BYTES_PER_ENVSTEP, measured=None, measured_frac=None, measured_bytes_per_iteration=None,
                   note="per GPU; measured = sum over all kernels of (PMC bytes per launch x launches) of one iteration / iteration time, null when the "
                        "committed profile is of other kernel sources")
        if traffic_src.get("hbm_bytes_per_iteration"):
            meas = traffic_src["hbm_bytes_per_iteration"] / iter_s / 1e9
            hbm.update(measured=round(meas, 1), measured_frac=round(meas / PEAK_HBM_GBS, 4), measured_bytes_per_iteration=round(traffic_src["hbm_bytes_per_iteration"]))

    # ---- non-headline legs, each in a child process started AFTER everything that feeds the headline is measured: (1) the same workload with
    # kbj_config.gemm_bf16x3 (the update's large GEMMs on the bf16 matrix cores through the exact three-way split of their fp32 operands,
    # DESIGN.md section 10b); (2) the training loop a user runs - HumanoidWalkingTask.launch() with logging, validation rollouts and periodic
    # checkpoints. A hard fault or hang in either costs its own object, not the line.
    variant = train_loop = None
    rollout_steps = task.T
    if not args.gemm_bf16x3 and not args.no_variants and not args.force_collective and world == 1:
        # (one GPU only: a multi-rank job reports its headline and nothing else)
        task.ctx.synchronize(); task.close(); task = None      # the headline context goes first: the child has the GPU to itself
        torch.cuda.empty_cache()
        variant = run_leg_in_child("variant", args, 120 + 3.0 * (args.steps + args.warmup))
        variant.setdefault("name", "gemm_bf16x3"); variant.setdefault("NOT_THE_HEADLINE", True)
        train_loop = run_leg_in_child("train_loop", args, 180 + 3.0 * args.loop_iterations)
        train_loop.setdefault("name", "train_loop")
        if train_loop.get("value"):
            train_loop["vs_headline"] = round(train_loop["value"] / value, 4)
    if forced is not None:      # third leg: the actor's gradient slice all-reduced on a second stream under the critic's tail
        forced.update(ms_per_step_forced_allreduce=round(elapsed / args.steps * 1e3, 2), allreduce_ms_per_iteration=round(allreduce_ms, 3),
                      allreduce_calls_per_iteration=allreduce_calls)
        task.config.overlap_allreduce = True
        forced["ms_per_step_forced_allreduce_overlapped"] = round(timed_steps(1, args.steps) / args.steps * 1e3, 2)
        task.config.overlap_allreduce = False
        forced["note"] = ("one GPU, backend nccl, world size 1: the all-reduce moves no data between GPUs; what is measured is RCCL's kernel launch + its "
                          "stream joining the schedule's lanes (DESIGN.md section 8)")
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
        "dtype": "f32" if not args.gemm_bf16x3 else "f32 (backward GEMMs: exact 3-way bf16 split of the fp32 operands, fp32 accumulation)", "data": "synthetic",
        "variant": "headline: plain fp32-MFMA kernels" if not args.gemm_bf16x3 else "NOT THE HEADLINE: --gemm-bf16x3 (kbj_config.gemm_bf16x3)",
        "config": {"workload": ("[variant gemm_bf16x3] " if args.gemm_bf16x3 else "") + wl_name.format(n=args.envs_per_gpu) + ", full iteration: "
                               f"100-step rollout + PPO update (batch 512/GPU, 3 passes, LSTM hidden {args.hidden}, depth 2)",
                   "envs_per_gpu": args.envs_per_gpu, "rollout_steps": rollout_steps, "batch_size_per_gpu": cfg.batch_size, "num_passes": cfg.num_passes,
                   "hidden_size": args.hidden, "baseline_config": args.config, "parallelism": f"env-sharded dp{world}, grad all-reduce {'per optimizer step' if args.allreduce == 'per_step' else 'once per pass (accumulated)'}"
                                  + (f" [{args.backend}{', ranks share GPU 0' if args.share_gpu else ''}]" if world > 1 else "")},
        "roofline": roofline, "roofline_secondary": roofline2, "cpu_baseline": cpu,
        # data-parallel exchange: ranks in the process group (1 = no collective runs), gradient all-reduce time of one iteration (HIP events
        # around every dist.all_reduce of the instrumented iteration, rank 0) and the whole-path HBM fraction
        "rccl_ranks": pg_ranks, "rccl_ranks_counted_by_allreduce": pg_counted, "collective_backend": pg_backend,
        "rank_ms_per_step": {"min": round(min(rank_ms), 2), "max": round(max(rank_ms), 2)}, "forced_collective": forced,
        "allreduce_ms_per_iteration": round(allreduce_ms, 3), "allreduce_calls_per_iteration": allreduce_calls,
        "hbm_whole_path": hbm,
        "variant_gemm_bf16x3": variant,
        # the number launch() delivers: the same workload through the reference-shaped training loop (logging + validation + checkpoints)
        "train_loop": train_loop,
    }
    print(json.dumps(out), flush=True)
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
