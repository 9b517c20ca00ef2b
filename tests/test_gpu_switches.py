"""Every formulation switch of the schedule (csrc/kbj_nn.hip `Sched`, read from the environment once per context) has a parity test here:
the non-default path must give the same minibatch gradient as autograd through the oracle (the a9 tolerances of tests/test_gpu_nn.py)
and the same rollout as the default path. README.md's switch list is this file's parameter list.

Also here: the deterministic update (kbj_config.deterministic), the forward-only on-policy pass (kbj_ppo_forward = get_ppo_variables)
and the RCCL leg of the data-parallel exchange on the one GPU a test box has (backend "nccl", world_size 1)."""
import os

import numpy as np
import pytest

from kbot_joystick_amd.spec import compiler, layout as L

pytestmark = pytest.mark.gpu

# name -> environment of the NON-default formulation
SWITCHES = {
    "KBJ_FOLD_ACTOR=0": {"KBJ_FOLD_ACTOR": "0"},
    "KBJ_FOLD_CRITIC=0": {"KBJ_FOLD_CRITIC": "0"},
    "KBJ_SEQ_FUSE=0": {"KBJ_SEQ_FUSE": "0"},
    "KBJ_SEQ_FUSE_OBS=0": {"KBJ_SEQ_FUSE_OBS": "0"},
    "KBJ_FUSED_CRITIC_HEAD=0": {"KBJ_FUSED_CRITIC_HEAD": "0"},
    "KBJ_ONE_STREAM=1": {"KBJ_ONE_STREAM": "1"},
    "KBJ_ROLLOUT_STEP=0": {"KBJ_ROLLOUT_STEP": "0"},
    "KBJ_DETERMINISTIC=1": {"KBJ_DETERMINISTIC": "1"},
    "KBJ_DEBUG=1": {"KBJ_DEBUG": "1"},
    "KBJ_CRITIC_LANE=2nd": {"KBJ_CRITIC_LANE": "2nd"},   # rounds 1-5 lane assignment: critic chain on the context's second stream, actor on the caller's
    "KBJ_BWD16=0": {"KBJ_BWD16": "0"},           # backward recurrences on the 32 x 32-tile form everywhere (round 4's kernel)
    "KBJ_BWD16=0+DET": {"KBJ_BWD16": "0", "KBJ_DETERMINISTIC": "1"},
    "KBJ_GEMM_X3=1": {"KBJ_GEMM_X3": "1"},       # = kbj_config.gemm_bf16x3: the backward pass's large GEMMs through the exact three-way bf16 split
}


def _ctx(monkeypatch, env, N, B, T, H, **kw):
    import torch
    from kbot_joystick_amd.host import binding as Bd
    for k in list(os.environ):
        if k.startswith("KBJ_") and k not in ("KBJ_LIB_NAME",):
            monkeypatch.delenv(k)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = compiler.load_model("kbot-headless")
    cfg = L.default_config(num_envs=N, batch_size=B, rollout_len=T, hidden_size=H, **kw)
    ctx = Bd.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)     # the switches are read here, once per context
    for k in env:
        monkeypatch.delenv(k)
    return m, cfg, ctx


def _problem(torch, m, cfg, ctx, N, B, T, H, seed=0):
    """A synthetic minibatch problem (as tests/test_gpu_nn.py) + the autograd gradient of the oracle."""
    from kbot_joystick_amd.host import buffers
    from oracle import nn as ON
    from tests.test_gpu_nn import _synthetic_traj
    P = ctx.param_count()
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(11, params)
    tr = _synthetic_traj(torch, buffers, N, T, H, seed=seed)
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    p64 = params.detach().cpu().double()
    g = torch.Generator(device="cpu").manual_seed(5)
    idx = torch.randperm(N, generator=g)[:B].int()
    ao, co = tr.actor_obs[:T].cpu().double(), tr.critic_obs[:T].cpu().double()
    act, done = tr.action.cpu().double(), tr.done.cpu().double()
    with torch.no_grad():
        ca = [[tr.carry0_actor_hc[l, k].cpu().double() for k in range(2)] for l in range(2)]
        cc = [[tr.carry0_critic_hc[l, k].cpu().double() for k in range(2)] for l in range(2)]
        lp, v, en, *_ = ON.ppo_variables(ON.unflatten(p64, H), cfg, jb, ao, co, act, done, ca, cc, tr.carry0_lpf.cpu().double())
    tr.logp.copy_((lp + 0.3 * torch.randn(T, N, generator=g).double()).float())
    tr.value.copy_((v + 0.3 * torch.randn(T, N, generator=g).double()).float())
    ctx.gae(tr.c, tr.adv, tr.target)
    ctx.synchronize()
    return params, tr, idx, (lp, v, en), (jb, p64, ao, co, act, done)


def _autograd(torch, cfg, tr, idx, H, aux):
    from oracle import nn as ON
    jb, p64, ao, co, act, done = aux
    adv_o, tgt_o = ON.gae(tr.value.cpu().double(), tr.reward.cpu().double(), done, cfg.gamma, cfg.lam)
    pf = p64.clone().requires_grad_(True)
    ii = idx.long()
    ca = [[tr.carry0_actor_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]
    cc = [[tr.carry0_critic_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]
    lp, v, en, *_ = ON.ppo_variables(ON.unflatten(pf, H), cfg, jb, ao[:, ii], co[:, ii], act[:, ii], done[:, ii], ca, cc, tr.carry0_lpf.cpu().double()[ii])
    loss, _ = ON.ppo_loss(cfg, lp, v, en, tr.logp.cpu().double()[:, ii], tr.value.cpu().double()[:, ii], adv_o[:, ii], tgt_o[:, ii])
    loss.backward()
    return pf.grad, float(loss.detach())


@pytest.mark.parametrize("name", [k for k in SWITCHES if k != "KBJ_ROLLOUT_STEP=0"])
@pytest.mark.parametrize("H,N,B,T", [(64, 12, 8, 7), (256, 96, 64, 9)])      # T = 9 in 4 chunks: 3 + 3 + 3 steps (one chunk empty); in 3: 3 + 3 + 3
def test_switch_gradient_matches_autograd(monkeypatch, name, H, N, B, T):
    import torch
    from oracle import nn as ON
    m, cfg, ctx = _ctx(monkeypatch, SWITCHES[name], N, B, T, H)
    params, tr, idx, _, aux = _problem(torch, m, cfg, ctx, N, B, T, H)
    P = ctx.param_count()
    grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
    ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
    ctx.synchronize()
    go, loss = _autograd(torch, cfg, tr, idx, H, aux)
    gg = grad.cpu().double()
    assert abs(float(metrics[0]) - loss) < 2e-4 * (1 + abs(loss))
    off = 0
    for leaf, shp in ON.param_shapes(H):
        n = int(np.prod(shp))
        a, b = gg[off:off + n], go[off:off + n]
        assert (a - b).abs().max() / (b.abs().max() + 1e-12) < 2e-3, (name, leaf)
        off += n
    assert (gg - go).norm() / go.norm() < 1e-4
    ctx.close()


@pytest.mark.parametrize("name", ["KBJ_ROLLOUT_STEP=0", "KBJ_FOLD_ACTOR=0", "KBJ_GEMM_X3=1"])
def test_switch_policy_step_matches_default(monkeypatch, name):
    """Rollout-side switches: one policy step (all carries, action mode, value) against the default formulation."""
    import torch
    from kbot_joystick_amd.host import buffers
    N, H = 100, 128
    outs = []
    for env in ({}, SWITCHES[name]):
        m, cfg, ctx = _ctx(monkeypatch, env, N, 50, 4, H)
        params = torch.zeros(ctx.param_count(), device="cuda:0")
        ctx.init_params(3, params)
        g = torch.Generator(device="cpu").manual_seed(0)
        aobs = torch.zeros(N, L.LD_ACTOR); aobs[:, :65] = torch.randn(N, 65, generator=g)
        cobs = torch.zeros(N, L.LD_CRITIC); cobs[:, :475] = torch.randn(N, 475, generator=g)
        carry = buffers.CarryBuffers(N, H, 2, "cuda:0")
        carry.actor_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5); carry.critic_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5)
        carry.lpf.copy_(torch.randn(N, 20, generator=g) * 0.3)
        a, lp, v = torch.zeros(N, 20, device="cuda:0"), torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
        for step in range(3):
            ctx.policy_step(params, aobs.cuda(), cobs.cuda(), carry.c, 7, step, True, a, lp, v)
        ctx.synchronize()
        outs.append([x.cpu().clone() for x in (a, v, carry.actor_hc, carry.critic_hc, carry.lpf)])
        ctx.close()
    for x, y in zip(*outs):
        assert (x - y).abs().max() < 2e-5


def test_deterministic_update_is_bit_reproducible(monkeypatch):
    """kbj_config.deterministic: two updates from the same state give bit-identical parameters (fixed-order reductions instead of fp32 /
    fp64 atomics); and the per-pass exchange variant then equals its manual restatement to 1e-7."""
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask
    from tests.test_gpu_host import _small
    for k in list(os.environ):
        if k.startswith("KBJ_") and k != "KBJ_LIB_NAME":
            monkeypatch.delenv(k)
    cfg = _small(num_envs=128, batch_size=32, num_passes=2, deterministic=True)
    task = HumanoidWalkingTask(cfg)
    assert task.kcfg.deterministic == 1
    task.rollout()
    snap = [t.clone() for t in (task.params, task.opt_m, task.opt_v)]
    results = []
    for _ in range(3):
        for t, s in zip((task.params, task.opt_m, task.opt_v), snap):
            t.copy_(s)
        task.opt_step = 0
        task.update()
        torch.cuda.synchronize()
        results.append((task.params.clone(), task.opt_m.clone(), task.opt_v.clone()))
    for r in results[1:]:
        for a, b in zip(results[0], r):
            assert torch.equal(a, b)
    assert not torch.equal(results[0][0], snap[0])
    # per-pass exchange variant against its manual restatement (tests/test_gpu_host.py loosens this to a few learning rates under atomics)
    acc_task = HumanoidWalkingTask(_small(num_envs=128, batch_size=32, num_passes=2, allreduce="per_pass", deterministic=True))
    ref = HumanoidWalkingTask(_small(num_envs=128, batch_size=32, num_passes=2, deterministic=True))
    acc_task.rollout()
    for name in ("actor_obs", "critic_obs", "aux", "action", "logp", "value", "reward", "carry0_actor_hc", "carry0_critic_hc", "carry0_lpf"):
        getattr(ref.traj, name).copy_(getattr(acc_task.traj, name))
    acc_task.update()
    ref.ctx.gae(ref.traj.c, ref.traj.adv, ref.traj.target)
    for p in range(2):
        g = torch.Generator(device="cpu"); g.manual_seed((cfg.seed * 1000003 + 0 * 97 + p) & 0x7FFFFFFF)
        perm = torch.randperm(128, generator=g).int().cuda()
        acc = torch.zeros_like(ref.params)
        for mb in range(4):
            ref.ctx.ppo_grad(ref.params, ref.traj.c, perm[32 * mb:32 * mb + 32].contiguous(), 32, ref.traj.adv, ref.traj.target, ref.grad, ref.metrics)
            acc += ref.grad
        ref.ctx.adamw_step(ref.params, ref.opt_m, ref.opt_v, acc, p + 1, 0.25)
    torch.cuda.synchronize()
    assert float((acc_task.params - ref.params).abs().max()) <= 1e-7
    for t in (task, acc_task, ref):
        t.ctx.close()


def test_next_minibatch_prefetch_changes_nothing(monkeypatch):
    """kbj_ppo_prefetch is a scheduling hint: with the deterministic reductions the gradient of minibatch k + 1 is bit-identical whether its
    head gathers were queued behind minibatch k (under the optimizer step) or run inside its own call; a prefetch for OTHER indices is
    ignored; and an update with the hint (HumanoidWalkingTask.update issues it) equals a manual loop without it."""
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask
    from tests.test_gpu_host import _small
    for k in list(os.environ):
        if k.startswith("KBJ_") and k != "KBJ_LIB_NAME":
            monkeypatch.delenv(k)
    cfg = _small(num_envs=128, batch_size=32, num_passes=1, deterministic=True)
    task = HumanoidWalkingTask(cfg)
    task.rollout()
    task.ctx.gae(task.traj.c, task.traj.adv, task.traj.target)
    perm = torch.randperm(128, generator=torch.Generator().manual_seed(1)).int().cuda()
    i0, i1, i2 = (perm[32 * k:32 * k + 32].contiguous() for k in range(3))
    g = [torch.zeros_like(task.params) for _ in range(4)]
    call = lambda idx, out: task.ctx.ppo_grad(task.params, task.traj.c, idx, 32, task.traj.adv, task.traj.target, out, task.metrics)
    call(i0, g[0]); call(i1, g[1])                                   # reference: no hint
    call(i0, g[2]); task.ctx.ppo_prefetch(task.traj.c, i1); call(i1, g[3])
    torch.cuda.synchronize()
    assert torch.equal(g[0], g[2]) and torch.equal(g[1], g[3])
    task.ctx.ppo_prefetch(task.traj.c, i2); call(i1, g[3])           # a hint for other indices is not used
    torch.cuda.synchronize()
    assert torch.equal(g[1], g[3])
    # whole update (hinted) against the manual loop (not hinted)
    ref = HumanoidWalkingTask(cfg)
    for name in ("actor_obs", "critic_obs", "aux", "action", "logp", "value", "reward", "carry0_actor_hc", "carry0_critic_hc", "carry0_lpf"):
        getattr(ref.traj, name).copy_(getattr(task.traj, name))
    task.update()
    ref.ctx.gae(ref.traj.c, ref.traj.adv, ref.traj.target)
    gen = torch.Generator(device="cpu"); gen.manual_seed((cfg.seed * 1000003 + 0 * 97 + 0) & 0x7FFFFFFF)
    pm = torch.randperm(128, generator=gen).int().cuda()
    for mb in range(4):
        ref.ctx.ppo_grad(ref.params, ref.traj.c, pm[32 * mb:32 * mb + 32].contiguous(), 32, ref.traj.adv, ref.traj.target, ref.grad, ref.metrics)
        ref.ctx.adamw_step(ref.params, ref.opt_m, ref.opt_v, ref.grad, mb + 1, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(task.params, ref.params)
    task.ctx.close(); ref.ctx.close()


def test_deterministic_gradient_at_the_baseline_minibatch(monkeypatch):
    """512 envs x 100 steps, H = 256: the deterministic gradient is bit-identical call to call and within 1e-5 (relative L2) of the atomic one."""
    import torch
    N = B = 512; T, H = 100, 256
    grads = {}
    for det in (1, 0):
        m, cfg, ctx = _ctx(monkeypatch, {}, N, B, T, H, deterministic=det)
        params, tr, idx, _, _ = _problem(torch, m, cfg, ctx, N, B, T, H)
        P = ctx.param_count()
        out = []
        for _ in range(2):
            grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
            ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
            ctx.synchronize()
            out.append(grad.clone())
        grads[det] = out
        ctx.close()
    assert torch.equal(grads[1][0], grads[1][1])
    assert float((grads[1][0] - grads[0][0]).norm() / grads[0][0].norm()) < 1e-5


def test_call_sequences_do_not_change_the_gradient(monkeypatch):
    """Round 6: kbj_ppo_grad leaves the recurrences' hand-off counters cleared for the next call (each lane clears its nets' blocks behind its
    last recurrence) and kbj_adamw_step skips its accumulator clear when the preceding kbj_ppo_grad already made it. Every other call sequence
    must fall back to its own clears: in deterministic mode (bit-reproducible) the gradient of the same minibatch is bit-identical whatever ran
    before it - another kbj_ppo_grad, a kbj_ppo_forward (which uses the forward counters and clears nothing at its tail), two optimizer steps in a
    row (the second clears its accumulator itself) - and the optimizer step from the same state gives the same parameters."""
    import torch
    N, B, T, H = 96, 32, 9, 128
    m, cfg, ctx = _ctx(monkeypatch, {}, N, B, T, H, deterministic=1)
    params, tr, idx, _, _ = _problem(torch, m, cfg, ctx, N, B, T, H)
    idx = idx.cuda()
    P = ctx.param_count()
    new = lambda: torch.zeros(P, device="cuda:0")
    met = torch.zeros(10, device="cuda:0")
    g = [new() for _ in range(4)]
    grad = lambda out: ctx.ppo_grad(params, tr.c, idx, B, tr.adv, tr.target, out, met)
    grad(g[0])                                                            # first call of the context: clears everything itself
    grad(g[1])                                                            # behind a kbj_ppo_grad: counters cleared by its tail
    lp, v = torch.zeros(T, B, device="cuda:0"), torch.zeros(T, B, device="cuda:0")
    ctx.ppo_forward(params, tr.c, idx, B, lp, v)                          # forward-only pass: uses the counters, leaves them dirty
    grad(g[2])
    p1, m1, v1 = params.clone(), new(), new()
    p2, m2, v2 = params.clone(), new(), new()
    ctx.adamw_step(p1, m1, v1, g[2], 1, 1.0)                              # behind kbj_ppo_grad: no accumulator clear in front of it
    ctx.adamw_step(p2, m2, v2, g[2], 1, 1.0)                              # a second step in a row: clears its accumulator itself
    grad(g[3])                                                            # behind optimizer steps
    ctx.synchronize()
    assert torch.isfinite(g[0]).all() and float(g[0].abs().max()) > 0
    assert torch.equal(g[0], g[1]) and torch.equal(g[0], g[2]) and torch.equal(g[0], g[3])
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and not torch.equal(p1, params)
    ctx.close()


def test_recurrence_residency_query(monkeypatch):
    """kbj_recurrence_residency: what a host needs to co-locate contexts on one GPU. At the BASELINE minibatch a recurrence launch is
    (512 / 32) x (256 / 32) = 128 workgroups (forward 32 x 32 tiles; backward 16 x 64 tiles: the same count), two launches are in flight (one on the
    one-stream schedule), and the device holds at least one workgroup of the worst-fitting recurrence kernel per CU."""
    import torch
    m, cfg, ctx = _ctx(monkeypatch, {}, 512, 512, 10, 256)
    grid, conc, slots = ctx.recurrence_residency()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert (grid, conc) == (128, 2) and slots >= cus and slots % cus == 0 and conc * grid <= slots
    ctx.close()
    m, cfg, ctx = _ctx(monkeypatch, {"KBJ_ONE_STREAM": "1"}, 64, 64, 10, 64)
    assert ctx.recurrence_residency()[:2] == (4, 1)
    ctx.close()


def test_gemm_bf16x3_gradient_at_the_baseline_minibatch(monkeypatch):
    """kbj_config.gemm_bf16x3 at the BASELINE minibatch (512 envs x 100 steps, H = 256): every large backward GEMM (input gradient
    51200 x 256 x 1024, paired weight gradients 1024 x 512 x 51200 and the folded layer-0 pairs with their ragged second problem) runs
    on the split kernel. The gradient is held against fp64 autograd through the oracle with the a9 bounds of the exact path (rel-L2 1e-4,
    per leaf 2e-3 of the leaf's largest entry) - and must not be further from it than the exact path is (x 1.25) - and against the exact
    path itself (rel-L2 1e-5, per leaf 1e-4): the two differ in rounding order only, as two runs of the exact path with its atomics do.
    The kernel names the library reports for the split launches are the instantiations as rocprofv3 prints them."""
    import torch
    from oracle import nn as ON
    N, B, T, H = 512, 512, 100, 256
    grads, go = {}, None
    for x3 in (0, 1):
        m, cfg, ctx = _ctx(monkeypatch, {}, N, B, T, H, gemm_bf16x3=x3)
        params, tr, idx, _, aux = _problem(torch, m, cfg, ctx, N, B, T, H)
        P = ctx.param_count()
        grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
        ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
        ctx.synchronize()
        grads[x3] = grad.clone()
        if go is None:
            go, _ = _autograd(torch, cfg, tr, idx, H, aux)       # same seeds in both legs: one oracle gradient serves both
        if x3:
            ctx.profile_begin()
            ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
            prof = ctx.profile_end()
            names = {k["name"]: k["launches"] for k in prof["kernels"]}
            # gemm_x3_kernel<TM, A_KC, B_KC, GEN>: weight-gradient pairs (both operands row-contiguous), input gradients (A k-contiguous), the
            # critic's input projection (general form: bias + row gather)
            assert names.get("kbj::gemm_x3_kernel<2, false, false, false>", 0) >= 4 and names.get("kbj::gemm_x3_kernel<2, true, false, false>", 0) >= 2, names
            assert names.get("kbj::gemm_x3_kernel<2, true, true, true>", 0) >= 1, names
            assert not any(n.startswith("kbj::gemm_x3_kernel<f") or n.startswith("kbj::gemm_x3_kernel<t") for n in names), names
        ctx.close()
    assert torch.isfinite(grads[1]).all()
    assert float((grads[1] - grads[0]).norm() / grads[0].norm()) < 1e-5
    g0, g1 = grads[0].cpu().double(), grads[1].cpu().double()
    e0, e1 = float((g0 - go).norm() / go.norm()), float((g1 - go).norm() / go.norm())
    assert e1 < 1e-4 and e1 <= 1.25 * e0 + 1e-7, (e0, e1)
    off = 0
    for name, shp in ON.param_shapes(H):
        n = int(np.prod(shp))
        a, b = grads[1][off:off + n], grads[0][off:off + n]
        assert float((a - b).abs().max() / (b.abs().max() + 1e-12)) < 1e-4, name
        assert float((g1[off:off + n] - go[off:off + n]).abs().max() / (go[off:off + n].abs().max() + 1e-12)) < 2e-3, name
        off += n


@pytest.mark.parametrize("H,N,B,T", [(64, 24, 8, 7), (256, 128, 64, 9), (100, 40, 20, 5)])      # 100: a zero-padded hidden size
def test_ppo_forward_matches_oracle_and_gradient_pass(monkeypatch, H, N, B, T):
    """kbj_ppo_forward (get_ppo_variables, train.py:1510-1524) on a trajectory the rollout did not produce: log_probs / values / entropy /
    action_std against the oracle's ppo_variables (the a9 tolerances), for every minibatch of the env set."""
    import torch
    from oracle import nn as ON
    m, cfg, ctx = _ctx(monkeypatch, {}, N, B, T, H)
    params, tr, _, (lp, v, en), aux = _problem(torch, m, cfg, ctx, N, B, T, H, seed=3)
    dev = "cuda:0"
    for mb in range(N // B):
        idx = torch.arange(mb * B, (mb + 1) * B, dtype=torch.int32)
        o_lp, o_v, o_en = (torch.zeros(T, B, device=dev) for _ in range(3))
        o_sd, o_mu = torch.zeros(T, B, 20, device=dev), torch.zeros(T, B, 20, device=dev)
        ctx.ppo_forward(params, tr.c, idx.cuda(), B, o_lp, o_v, o_en, o_sd, o_mu)
        ctx.synchronize()
        sl = slice(mb * B, (mb + 1) * B)
        assert (o_lp.cpu().double() - lp[:, sl]).abs().max() < 2e-4
        assert (o_v.cpu().double() - v[:, sl]).abs().max() < 2e-5
        assert (o_en.cpu().double() - en[:, sl]).abs().max() < 2e-4
        assert float(o_sd.min()) > 0 and float(o_sd.max()) <= cfg.max_std + 1e-6
        # entropy of a diagonal Gaussian from the returned standard deviations (train.py:1486-1487)
        ent = (0.5 + 0.5 * np.log(2 * np.pi) + o_sd.cpu().double().log()).sum(-1)
        assert (ent - o_en.cpu().double()).abs().max() < 1e-4
        lp2 = ON.gaussian_logp(tr.action.cpu().double()[:, sl], o_mu.cpu().double(), o_sd.cpu().double())
        assert (lp2 - o_lp.cpu().double()).abs().max() < 1e-4
    ctx.close()


def test_get_ppo_variables_reproduces_the_rollout(monkeypatch):
    """The separate on-policy pass of the reference (get_ppo_variables with the pre-update model) against what the rollout's own policy
    steps stored: same parameters, same observations, same carries -> log-probs and values agree to 1e-5 (mean) / 1e-4 (max)."""
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask
    from tests.test_gpu_host import _small
    for k in list(os.environ):
        if k.startswith("KBJ_") and k != "KBJ_LIB_NAME":
            monkeypatch.delenv(k)
    task = HumanoidWalkingTask(_small(num_envs=128, batch_size=64, hidden_size=128, rollout_length_seconds=0.4))
    task.rollout(); task.rollout()       # the second rollout starts from non-trivial carries and contains resets
    torch.cuda.synchronize()
    pv = task.get_ppo_variables(task.traj)
    torch.cuda.synchronize()
    dl, dv = (pv["log_probs"] - task.traj.logp).abs(), (pv["values"] - task.traj.value).abs()
    assert float(dl.mean()) < 1e-5 and float(dl.max()) < 1e-4, (float(dl.mean()), float(dl.max()))
    assert float(dv.mean()) < 1e-5 and float(dv.max()) < 1e-4, (float(dv.mean()), float(dv.max()))
    assert pv["action_std"].shape == (task.T, task.N, 20) and torch.isfinite(pv["entropy"]).all()
    base = task.get_ppo_variables()
    assert base["log_probs"] is task.traj.logp
    task.ctx.close()


def test_rccl_allreduce_world_size_one(tmp_path):
    """The RCCL leg of the data-parallel exchange on one GPU: a child process initialises torch.distributed with backend "nccl"
    (= RCCL), world_size 1, and runs the product path - a full iteration incl. the per-step gradient all-reduce on the compute stream
    between the persistent recurrences - then checks it against the same iteration without a process group (bit-identical: the sum
    over one rank is the identity and grad_scale is 1)."""
    import subprocess
    import sys
    script = tmp_path / "rccl1.py"
    script.write_text('''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
from kbot_joystick_amd.host import dist as D
cfg = dict(num_envs=128, batch_size=64, hidden_size=64, rollout_length_seconds=0.2, robot="kbot-headless", seed=5, deterministic=True)
ref = HumanoidWalkingTask(launch_config(**cfg))
ref.train_iteration(); torch.cuda.synchronize()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
task = HumanoidWalkingTask(launch_config(**cfg), rank=0, world_size=1)
calls = []
orig = dist.all_reduce
def counted(t, *a, **k):
    calls.append(t.numel()); return orig(t, *a, **k)
dist.all_reduce = counted
D.FORCE_COLLECTIVE = True          # world_size 1 would skip the collective: run it anyway (the exchange step itself is what is tested)
task.train_iteration(); torch.cuda.synchronize()
assert len(calls) == task.kcfg.num_passes * (task.N // task.B) and calls[0] == task.P, calls
assert torch.equal(task.params, ref.params), float((task.params - ref.params).abs().max())
t = torch.ones(4, device="cuda"); orig(t); assert float(t.sum()) == 4.0
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", len(calls))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
