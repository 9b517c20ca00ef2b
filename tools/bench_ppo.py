"""Micro-benchmark of kbj_ppo_grad (one minibatch) on synthetic data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kbot_joystick_amd.spec import compiler, layout as L
from kbot_joystick_amd.host import binding as B, buffers
Bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, T, H = max(1024, Bs), 100, 256
m = compiler.load_model("kbot-headless"); cfg = L.default_config(num_envs=N, batch_size=Bs, rollout_len=T, hidden_size=H)
ctx = B.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
P = ctx.param_count(); params = torch.zeros(P, device="cuda"); ctx.init_params(1, params)
tr = buffers.TrajBuffers(T, N, H, 2, "cuda")
tr.actor_obs.normal_(); tr.critic_obs.normal_(); tr.action.normal_(); tr.logp.normal_(); tr.value.normal_(); tr.reward.uniform_()
ctx.gae(tr.c, tr.adv, tr.target)
grad, met = torch.zeros(P, device="cuda"), torch.zeros(10, device="cuda")
idx = torch.randperm(N)[:Bs].int().cuda()
for _ in range(2): ctx.ppo_grad(params, tr.c, idx, Bs, tr.adv, tr.target, grad, met)
ctx.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 5; e0.record()
for _ in range(K): ctx.ppo_grad(params, tr.c, idx, Bs, tr.adv, tr.target, grad, met)
e1.record(); ctx.synchronize()
print(f"ppo_grad {e0.elapsed_time(e1)/K:.2f} ms per minibatch (B={Bs}, T={T}, H={H}) -> x48 = {e0.elapsed_time(e1)/K*48:.0f} ms/iteration")
ctx.profile_begin()
ctx.ppo_grad(params, tr.c, idx, Bs, tr.adv, tr.target, grad, met)
ctx.synchronize()
for k in ctx.profile_end()["kernels"]:
    print(f"  {k['name']:44s} n={k['launches']:3d} avg={k['total_ms'] * 1e3 / k['launches']:8.1f} us  {k['flops'] / (k['total_ms'] * 1e-3) / 1e12:6.1f} TF")
