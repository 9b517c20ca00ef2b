"""Micro-benchmark of kbj_policy_step alone (the rollout's per-step network work, no env kernel beside it): per-kernel HIP-event times of
the layer-step kernels. usage: bench_policy.py [N] [H]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kbot_joystick_amd.spec import compiler, layout as L
from kbot_joystick_amd.host import binding as B, buffers
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = compiler.load_model("kbot-headless"); cfg = L.default_config(num_envs=N, batch_size=min(512, N), hidden_size=H)
ctx = B.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
a, c = torch.zeros(N, 68, device=dev), torch.zeros(N, 476, device=dev)
a[:, :65] = torch.randn(N, 65, device=dev, generator=g); c[:, :475] = torch.randn(N, 475, device=dev, generator=g)
params = torch.zeros(ctx.param_count(), device=dev); ctx.init_params(1, params)
carry = buffers.CarryBuffers(N, H, cfg.depth, dev)
act, lp, v = torch.zeros(N, 20, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
for t in range(20): ctx.policy_step(params, a, c, carry.c, 1, t, False, act, lp, v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 100
e0.record()
for t in range(K): ctx.policy_step(params, a, c, carry.c, 1, 100 + t, False, act, lp, v)
e1.record(); torch.cuda.synchronize()
print(f"N={N} H={H} policy_step {e0.elapsed_time(e1) / K * 1e3:.1f} us/step (actor + critic, one stream) checksum {float(act.sum()):.6f} {float(v.sum()):.6f}")
ctx.profile_begin()
for t in range(K): ctx.policy_step(params, a, c, carry.c, 1, 300 + t, False, act, lp, v)
torch.cuda.synchronize()
for k in ctx.profile_end()["kernels"]:
    if k["launches"]:
        print(f"  {k['name'][:60]:60s} {k['launches']:5d} launches  {k['total_ms'] / k['launches'] * 1e3:8.1f} us  {k['flops'] / k['total_ms'] / 1e9:7.1f} TFLOP/s")
ctx.close()
