"""Micro-benchmark of kbj_env_step alone. usage: bench_env.py N [key=value ...] (kbj_config overrides)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kbot_joystick_amd.spec import compiler, layout as L
from kbot_joystick_amd.host import binding as B
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
m = compiler.load_model("kbot-headless"); cfg = L.default_config(num_envs=N, batch_size=min(512, N), **kw)
ctx = B.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
dev = "cuda:0"
a, c, x = torch.zeros(N, 68, device=dev), torch.zeros(N, 476, device=dev), torch.zeros(N, 72, device=dev)
a2, c2, x2 = torch.zeros_like(a), torch.zeros_like(c), torch.zeros_like(x)
ctx.env_reset_all(1, a, c, x)
act = torch.from_numpy(np.tile(np.array(m.joint_bias, np.float32), (N, 1))).cuda()
act = act + float(os.environ.get("KBJ_ACT_NOISE", "0.3")) * torch.randn(N, 20, device=dev, generator=torch.Generator(device=dev).manual_seed(0))   # a random-init policy's spread
for _ in range(30): ctx.env_step(act, x, a2, c2, x2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 40
e0.record()
for _ in range(K): ctx.env_step(act, x, a2, c2, x2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"N={N} {kw} env_step {ms:.3f} ms/step -> {N/ms*1e3:.3e} env-steps/s; 100-step rollout {ms*100:.1f} ms")
