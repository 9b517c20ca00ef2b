"""GPU parity tests of the actor-critic / PPO kernels through the C ABI against the torch CPU oracle (oracle/nn.py).
fp32 MFMA products are exact fp32 fma chains, so the tolerance is fp32 round-off of differently ordered sums."""
import numpy as np
import pytest

from kbot_joystick_amd.spec import compiler, layout as L

pytestmark = pytest.mark.gpu


def _setup(N, B, T, H, **kw):
    import torch
    from kbot_joystick_amd.host import binding as Bd, buffers
    m = compiler.load_model("kbot-headless")
    cfg = L.default_config(num_envs=N, batch_size=B, rollout_len=T, hidden_size=H, **kw)
    ctx = Bd.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
    return m, cfg, ctx, torch, buffers


@pytest.mark.parametrize("H,N", [(64, 96), (128, 96), (256, 96), (256, 100)])     # 128 = the reference dataclass default (train.py:78-81), 256 = the launch value;
def test_policy_step_matches_oracle(H, N):                                         # N = 100: a ragged last row group of the layer-step kernel
    m, cfg, ctx, torch, buffers = _setup(N, 32, 4, H)
    P = ctx.param_count()
    from oracle import nn as ON
    assert P == ON.param_count(H)
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(3, params)
    ctx.synchronize()
    pn = params.cpu().numpy()
    assert np.isfinite(pn).all() and abs(pn.mean()) < 1e-2 and np.abs(pn).max() <= 1 / np.sqrt(min(65, H)) + 1e-6
    g = torch.Generator(device="cpu").manual_seed(0)
    aobs = torch.zeros(N, L.LD_ACTOR); aobs[:, :65] = torch.randn(N, 65, generator=g)
    cobs = torch.zeros(N, L.LD_CRITIC); cobs[:, :475] = torch.randn(N, 475, generator=g)
    carry = buffers.CarryBuffers(N, H, 2, "cuda:0")
    carry.actor_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5)
    carry.critic_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5)
    carry.lpf.copy_(torch.randn(N, 20, generator=g) * 0.3)
    hc_a0, hc_c0, lpf0 = carry.actor_hc.cpu().double(), carry.critic_hc.cpu().double(), carry.lpf.cpu().double()
    action, logp, value = torch.zeros(N, 20, device="cuda:0"), torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
    ctx.policy_step(params, aobs.cuda(), cobs.cuda(), carry.c, 7, 5, True, action, logp, value)
    ctx.synchronize()
    p = ON.unflatten(params.detach().cpu().double(), H)
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    out_a, ca = ON.net_forward(p, "actor", aobs[:, :65].double(), [[hc_a0[l, 0], hc_a0[l, 1]] for l in range(2)])
    mean, std, lpf1 = ON.actor_head(out_a, aobs.double(), lpf0, jb, cfg)
    out_c, cc = ON.net_forward(p, "critic", cobs[:, :475].double(), [[hc_c0[l, 0], hc_c0[l, 1]] for l in range(2)])
    assert (action.cpu().double() - mean).abs().max() < 2e-5            # argmax -> mode
    assert (value.cpu().double() - out_c[:, 0]).abs().max() < 2e-5
    assert (logp.cpu().double() - ON.gaussian_logp(mean, mean, std)).abs().max() < 1e-4
    for l in range(2):
        assert (carry.actor_hc[l, 0].cpu().double() - ca[l][0]).abs().max() < 1e-5
        assert (carry.actor_hc[l, 1].cpu().double() - ca[l][1]).abs().max() < 1e-5
        assert (carry.critic_hc[l, 1].cpu().double() - cc[l][1]).abs().max() < 1e-5
    assert (carry.lpf.cpu().double() - lpf1).abs().max() < 1e-5
    # sampling: same seed/step -> same draw; different step -> different draw; logp consistent with the sample
    a1, a2, lp1 = torch.zeros_like(action), torch.zeros_like(action), torch.zeros_like(logp)
    c2, c3, c4 = (buffers.CarryBuffers(N, H, 2, "cuda:0") for _ in range(3))
    ctx.policy_step(params, aobs.cuda(), cobs.cuda(), c2.c, 7, 5, False, a1, lp1, value)
    ctx.policy_step(params, aobs.cuda(), cobs.cuda(), c3.c, 7, 5, False, a2, logp, value)
    ctx.synchronize()
    assert torch.equal(a1, a2)
    ctx.policy_step(params, aobs.cuda(), cobs.cuda(), c4.c, 7, 6, False, a2, logp, value)
    ctx.synchronize()
    assert not torch.equal(a1, a2)
    out0, _ = ON.net_forward(p, "actor", aobs[:, :65].double(), ON.zero_carry(N, H, 2, torch.float64))
    mean0, std0, _ = ON.actor_head(out0, aobs.double(), torch.zeros(N, 20, dtype=torch.float64), jb, cfg)
    assert (lp1.cpu().double() - ON.gaussian_logp(a1.cpu().double(), mean0, std0)).abs().max() < 1e-3
    z = ((a1.cpu().double() - mean0) / std0).flatten()
    assert abs(z.mean()) < 0.1 and abs(z.std() - 1) < 0.1               # unit Gaussian draws
    ctx.close()


def _synthetic_traj(torch, buffers, N, T, H, seed=0, mirror=False, depth=2):
    g = torch.Generator(device="cpu").manual_seed(seed)
    tr = buffers.TrajBuffers(T, N, H, depth, "cuda:0", mirror=mirror)
    tr.actor_obs[:, :, :65] = (torch.randn(T + 1, N, 65, generator=g) * 0.5).cuda()
    tr.critic_obs[:, :, :475] = (torch.randn(T + 1, N, 475, generator=g) * 0.5).cuda()
    tr.action.copy_(torch.randn(T, N, 20, generator=g) * 0.3)
    done = (torch.rand(T, N, generator=g) < 0.15).float() * torch.where(torch.rand(T, N, generator=g) < 0.5, -1.0, 1.0)
    tr.aux[:T, :, L.AUX["DONE"]] = done.cuda()
    tr.reward.copy_(torch.rand(T, N, generator=g))
    tr.carry0_actor_hc.copy_(torch.randn(depth, 2, N, H, generator=g) * 0.3)
    tr.carry0_critic_hc.copy_(torch.randn(depth, 2, N, H, generator=g) * 0.3)
    tr.carry0_lpf.copy_(torch.randn(N, 20, generator=g) * 0.2)
    if mirror:
        tr.carry0_actor_mirror_hc.copy_(torch.randn(depth, 2, N, H, generator=g) * 0.3)
        tr.carry0_critic_mirror_hc.copy_(torch.randn(depth, 2, N, H, generator=g) * 0.3)
        tr.carry0_lpf_mirror.copy_(torch.randn(N, 20, generator=g) * 0.2)
    return tr


@pytest.mark.parametrize("H,N,B,T", [(64, 12, 8, 7), (128, 70, 35, 6), (256, 40, 32, 9), (384, 300, 256, 5), (512, 600, 512, 6),   # wide layers: two lanes / one stream (kbj_nn.hip SEQ_FUSED_MAX_H)
                                     (256, 512, 512, 100)])   # the last one = the BASELINE minibatch
def test_ppo_grad_matches_autograd(H, N, B, T):
    m, cfg, ctx, torch, buffers = _setup(N, B, T, H)
    from oracle import nn as ON
    P = ctx.param_count()
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(11, params)
    tr = _synthetic_traj(torch, buffers, N, T, H)
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    p64 = params.detach().cpu().double()
    pd = ON.unflatten(p64, H)
    g = torch.Generator(device="cpu").manual_seed(5)
    idx = torch.randperm(N, generator=g)[:B].int()
    ao, co = tr.actor_obs[:T].cpu().double(), tr.critic_obs[:T].cpu().double()
    act, done = tr.action.cpu().double(), tr.done.cpu().double()
    with torch.no_grad():
        ca = [[tr.carry0_actor_hc[l, k].cpu().double() for k in range(2)] for l in range(2)]
        cc = [[tr.carry0_critic_hc[l, k].cpu().double() for k in range(2)] for l in range(2)]
        lp, v, en, *_ = ON.ppo_variables(pd, cfg, jb, ao, co, act, done, ca, cc, tr.carry0_lpf.cpu().double())
    tr.logp.copy_((lp + 0.3 * torch.randn(T, N, generator=g).double()).float())       # some ratios leave the clip range
    tr.value.copy_((v + 0.3 * torch.randn(T, N, generator=g).double()).float())
    ctx.gae(tr.c, tr.adv, tr.target)
    ctx.synchronize()
    adv_o, tgt_o = ON.gae(tr.value.cpu().double(), tr.reward.cpu().double(), done, cfg.gamma, cfg.lam)
    assert (tr.adv.cpu().double() - adv_o).abs().max() < 1e-5 and (tr.target.cpu().double() - tgt_o).abs().max() < 1e-5
    grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
    ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
    ctx.synchronize()
    # oracle: autograd through the minibatch
    pf = p64.clone().requires_grad_(True)
    pdg = ON.unflatten(pf, H)
    ii = idx.long()
    ca = [[tr.carry0_actor_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]
    cc = [[tr.carry0_critic_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]
    lp, v, en, *_ = ON.ppo_variables(pdg, cfg, jb, ao[:, ii], co[:, ii], act[:, ii], done[:, ii], ca, cc, tr.carry0_lpf.cpu().double()[ii])
    loss, mt = ON.ppo_loss(cfg, lp, v, en, tr.logp.cpu().double()[:, ii], tr.value.cpu().double()[:, ii], adv_o[:, ii], tgt_o[:, ii])
    loss.backward()
    go = pf.grad
    mg = metrics.cpu().double()
    for k, name in enumerate(["loss", "policy", "value", "entropy", "clipfrac", "kl", "adv_mean", "adv_std"]):
        assert abs(mg[k] - float(mt[name].detach())) < 2e-4 * (1 + abs(float(mt[name].detach()))), (name, float(mg[k]), float(mt[name].detach()))
    assert 0.02 < float(mt["clipfrac"]) < 0.98                         # both clip branches exercised
    gg = grad.cpu().double()
    off = 0
    for name, shp in ON.param_shapes(H):
        n = int(np.prod(shp))
        a, b = gg[off:off + n], go[off:off + n]
        err = (a - b).abs().max() / (b.abs().max() + 1e-12)
        assert err < 2e-3, (name, float(err), float(b.abs().max()))
        off += n
    assert (gg - go).norm() / go.norm() < 1e-4
    # caller-supplied advantage statistics (kbj_set_advantage_sums, the data-parallel variant of SURVEY 8e): the sums of ANOTHER set - here
    # the whole rollout's advantages, as if seven more ranks had contributed - must give the oracle's gradient under that normalisation,
    # and NULL must restore the minibatch's own
    if T < 100:
        from kbot_joystick_amd.host import dist as D
        sums = D.global_advantage_sums(tr.adv, 1)
        ctx.set_advantage_sums(sums)
        g2, m2 = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
        ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, g2, m2)
        ctx.synchronize()
        pf2 = p64.clone().requires_grad_(True)
        lp2, v2, en2, *_ = ON.ppo_variables(ON.unflatten(pf2, H), cfg, jb, ao[:, ii], co[:, ii], act[:, ii], done[:, ii], ca, cc, tr.carry0_lpf.cpu().double()[ii])
        loss2, mt2 = ON.ppo_loss(cfg, lp2, v2, en2, tr.logp.cpu().double()[:, ii], tr.value.cpu().double()[:, ii], adv_o[:, ii], tgt_o[:, ii],
                                 adv_sums=torch.stack([adv_o.sum(), (adv_o ** 2).sum(), torch.tensor(float(adv_o.numel()), dtype=torch.float64)]))
        loss2.backward()
        assert (g2.cpu().double() - pf2.grad).norm() / pf2.grad.norm() < 1e-4
        assert abs(float(m2[6]) - float(adv_o.mean())) < 1e-5 and abs(float(m2[7]) - float(adv_o.std(unbiased=False))) < 1e-5
        assert (g2.cpu().double() - gg).norm() / gg.norm() > 1e-3           # it IS a different normalisation
        ctx.set_advantage_sums(None)
        g3 = torch.zeros(P, device="cuda:0")
        ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, g3, m2)
        ctx.synchronize()
        assert (g3.cpu().double() - gg).norm() / gg.norm() < 1e-5
    # AdamW + global-norm clip
    mom, var = torch.zeros(P, device="cuda:0"), torch.zeros(P, device="cuda:0")
    p_o, m_o, v_o = params.cpu().double().clone(), torch.zeros(P, dtype=torch.float64), torch.zeros(P, dtype=torch.float64)
    for step in (1, 2):
        ctx.adamw_step(params, mom, var, grad, step, 0.5)
        ON.adamw_step(cfg, p_o, m_o, v_o, gg.clone(), step, 0.5)
    ctx.synchronize()
    assert (params.cpu().double() - p_o).abs().max() < 1e-6
    ctx.close()


@pytest.mark.parametrize("H,D", [(192, 2), (128, 1), (64, 3), (256, 4), (96, 2), (200, 2), (40, 1), (7, 3), (320, 2), (512, 1), (400, 3)])
def test_other_depths_and_hidden_sizes_match_oracle(H, D):
    """`hidden_size` and `depth` are user fields of the reference config (train.py:78-85). The library serves any hidden size up to 512
    (multiples of 64 natively, above 256 on the wide schedule of kbj_nn.hip SEQ_FUSED_MAX_H; the others zero padded to the next one at the ABI boundary: parameters, carries and gradients keep the
    caller's hidden_size layout) and depth 1..4: one policy step (mode, value, every carry plane) and one minibatch gradient against the
    torch oracle / autograd."""
    N, B, T = 70, 35, 6
    m, cfg, ctx, torch, buffers = _setup(N, B, T, H, depth=D)
    from oracle import nn as ON
    P = ctx.param_count()
    assert P == ON.param_count(H, D)
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(4, params)
    p64 = params.detach().cpu().double()
    pd = ON.unflatten(p64, H, D)
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    g = torch.Generator(device="cpu").manual_seed(1)
    # ---- policy step ----
    aobs = torch.zeros(N, L.LD_ACTOR); aobs[:, :65] = torch.randn(N, 65, generator=g)
    cobs = torch.zeros(N, L.LD_CRITIC); cobs[:, :475] = torch.randn(N, 475, generator=g)
    carry = buffers.CarryBuffers(N, H, D, "cuda:0")
    carry.actor_hc.copy_(torch.randn(D, 2, N, H, generator=g) * 0.5)
    carry.critic_hc.copy_(torch.randn(D, 2, N, H, generator=g) * 0.5)
    carry.lpf.copy_(torch.randn(N, 20, generator=g) * 0.3)
    hc_a0, hc_c0, lpf0 = carry.actor_hc.cpu().double(), carry.critic_hc.cpu().double(), carry.lpf.cpu().double()
    action, logp, value = torch.zeros(N, 20, device="cuda:0"), torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
    ctx.policy_step(params, aobs.cuda(), cobs.cuda(), carry.c, 7, 5, True, action, logp, value)
    ctx.synchronize()
    out_a, ca = ON.net_forward(pd, "actor", aobs[:, :65].double(), [[hc_a0[l, 0], hc_a0[l, 1]] for l in range(D)], D)
    mean, std, lpf1 = ON.actor_head(out_a, aobs.double(), lpf0, jb, cfg)
    out_c, cc = ON.net_forward(pd, "critic", cobs[:, :475].double(), [[hc_c0[l, 0], hc_c0[l, 1]] for l in range(D)], D)
    assert (action.cpu().double() - mean).abs().max() < 2e-5
    assert (value.cpu().double() - out_c[:, 0]).abs().max() < 2e-5
    for l in range(D):
        for k in range(2):
            assert (carry.actor_hc[l, k].cpu().double() - ca[l][k]).abs().max() < 1e-5
            assert (carry.critic_hc[l, k].cpu().double() - cc[l][k]).abs().max() < 1e-5
    # ---- one minibatch gradient ----
    tr = _synthetic_traj(torch, buffers, N, T, H, depth=D)
    idx = torch.randperm(N, generator=g)[:B].int()
    ao, co = tr.actor_obs[:T].cpu().double(), tr.critic_obs[:T].cpu().double()
    act, done = tr.action.cpu().double(), tr.done.cpu().double()
    with torch.no_grad():
        c_a = [[tr.carry0_actor_hc[l, k].cpu().double() for k in range(2)] for l in range(D)]
        c_c = [[tr.carry0_critic_hc[l, k].cpu().double() for k in range(2)] for l in range(D)]
        lp, v, en, *_ = ON.ppo_variables(pd, cfg, jb, ao, co, act, done, c_a, c_c, tr.carry0_lpf.cpu().double(), D)
    tr.logp.copy_((lp + 0.3 * torch.randn(T, N, generator=g).double()).float())
    tr.value.copy_((v + 0.3 * torch.randn(T, N, generator=g).double()).float())
    ctx.gae(tr.c, tr.adv, tr.target)
    grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
    ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
    ctx.synchronize()
    adv_o, tgt_o = ON.gae(tr.value.cpu().double(), tr.reward.cpu().double(), done, cfg.gamma, cfg.lam)
    pf = p64.clone().requires_grad_(True)
    pdg = ON.unflatten(pf, H, D)
    ii = idx.long()
    c_a = [[tr.carry0_actor_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(D)]
    c_c = [[tr.carry0_critic_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(D)]
    lp, v, en, *_ = ON.ppo_variables(pdg, cfg, jb, ao[:, ii], co[:, ii], act[:, ii], done[:, ii], c_a, c_c, tr.carry0_lpf.cpu().double()[ii], D)
    loss, mt = ON.ppo_loss(cfg, lp, v, en, tr.logp.cpu().double()[:, ii], tr.value.cpu().double()[:, ii], adv_o[:, ii], tgt_o[:, ii])
    loss.backward()
    go, gg = pf.grad, grad.cpu().double()
    assert abs(float(metrics[0]) - float(loss.detach())) < 2e-4 * (1 + abs(float(loss.detach())))
    off = 0
    for name, shp in ON.param_shapes(H, D):
        n = int(np.prod(shp))
        a, b = gg[off:off + n], go[off:off + n]
        assert (a - b).abs().max() / (b.abs().max() + 1e-12) < 2e-3, name
        off += n
    assert off == P and (gg - go).norm() / go.norm() < 1e-4
    ctx.close()


def _make_obs_consistent(torch, tr):
    """The mirror of roll/pitch/zero-command is re-derived from the projected gravity / command by the reference
    (train.py:1596-1623); give the synthetic rows the same internal consistency the env's rows have."""
    from oracle import nn as ON
    for obs in (tr.actor_obs, tr.critic_obs):
        g = obs[..., 42:45].cpu().double()
        g = g / g.norm(dim=-1, keepdim=True)
        obs[..., 40:45] = ON._encode_pg(g).float().cuda()
        obs[..., 48] = (obs[..., 49:52].norm(dim=-1) < 1e-3).float()


def test_mirror_obs_and_carries_match_oracle():
    """kbj_policy_step with the mirror losses on: the mirror-branch carries advance on mirrored observations."""
    N, H = 64, 64
    m, cfg, ctx, torch, buffers = _setup(N, 32, 4, H, actor_mirror_loss_scale=1.0, critic_mirror_loss_scale=0.01)
    from oracle import nn as ON
    params = torch.zeros(ctx.param_count(), device="cuda:0")
    ctx.init_params(3, params)
    tr = _synthetic_traj(torch, buffers, N, 1, H, seed=4, mirror=True)
    _make_obs_consistent(torch, tr)
    carry = buffers.CarryBuffers(N, H, 2, "cuda:0", mirror=True)
    g = torch.Generator(device="cpu").manual_seed(1)
    carry.actor_mirror_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5)
    carry.critic_mirror_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5)
    carry.lpf_mirror.copy_(torch.randn(N, 20, generator=g) * 0.3)
    am0, cm0, lm0 = carry.actor_mirror_hc.cpu().double(), carry.critic_mirror_hc.cpu().double(), carry.lpf_mirror.cpu().double()
    action, logp, value = torch.zeros(N, 20, device="cuda:0"), torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
    ctx.policy_step(params, tr.actor_obs[0], tr.critic_obs[0], carry.c, 7, 0, True, action, logp, value)
    ctx.synchronize()
    p = ON.unflatten(params.cpu().double(), H)
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    ao_m = ON.mirror_actor_obs(tr.actor_obs[0].cpu().double(), m)
    co_m = ON.mirror_critic_obs(tr.critic_obs[0].cpu().double(), m)
    out_am, cam = ON.net_forward(p, "actor", ao_m[:, :65], [[am0[l, 0], am0[l, 1]] for l in range(2)])
    _, _, lpf_m1 = ON.actor_head(out_am, ao_m, lm0, jb, cfg)
    _, ccm = ON.net_forward(p, "critic", co_m[:, :475], [[cm0[l, 0], cm0[l, 1]] for l in range(2)])
    for l in range(2):
        for k in range(2):
            assert (carry.actor_mirror_hc[l, k].cpu().double() - cam[l][k]).abs().max() < 1e-5
            assert (carry.critic_mirror_hc[l, k].cpu().double() - ccm[l][k]).abs().max() < 1e-5
    assert (carry.lpf_mirror.cpu().double() - lpf_m1).abs().max() < 1e-5
    # carry reset clears the mirror branches too
    done = torch.zeros(N, device="cuda:0"); done[::2] = 1.0
    ctx.carry_reset(carry.c, done, 1)
    ctx.synchronize()
    assert float(carry.actor_mirror_hc[:, :, ::2].abs().max()) == 0 and float(carry.lpf_mirror[::2].abs().max()) == 0
    assert float(carry.critic_mirror_hc[:, :, 1::2].abs().max()) > 0
    ctx.close()


@pytest.mark.parametrize("H,N,B,T", [(64, 12, 8, 7), (256, 40, 32, 5)])
def test_ppo_grad_with_mirror_losses_matches_autograd(H, N, B, T):
    """Row a10: action/value mirror aux losses (train.py:1463-1481) in loss, metrics and gradient."""
    m, cfg, ctx, torch, buffers = _setup(N, B, T, H, actor_mirror_loss_scale=1.0, critic_mirror_loss_scale=0.25)
    from oracle import nn as ON
    P = ctx.param_count()
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(11, params)
    tr = _synthetic_traj(torch, buffers, N, T, H, mirror=True)
    _make_obs_consistent(torch, tr)
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    p64 = params.detach().cpu().double()
    g = torch.Generator(device="cpu").manual_seed(5)
    idx = torch.randperm(N, generator=g)[:B].int()
    ii = idx.long()
    ao, co = tr.actor_obs[:T].cpu().double(), tr.critic_obs[:T].cpu().double()
    act, done = tr.action.cpu().double(), tr.done.cpu().double()
    carry = lambda t: [[t[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]

    def variables(pd):
        return ON.ppo_variables_mirror(pd, cfg, m, jb, ao[:, ii], co[:, ii], act[:, ii], done[:, ii], carry(tr.carry0_actor_hc), carry(tr.carry0_critic_hc),
                                       tr.carry0_lpf.cpu().double()[ii], carry(tr.carry0_actor_mirror_hc), carry(tr.carry0_critic_mirror_hc),
                                       tr.carry0_lpf_mirror.cpu().double()[ii])
    with torch.no_grad():
        lp, v, *_ = variables(ON.unflatten(p64, H))
    lp_old = torch.zeros(T, N, dtype=torch.float64); v_old = torch.zeros(T, N, dtype=torch.float64)
    lp_old[:, ii] = lp + 0.3 * torch.randn(T, B, generator=g).double()
    v_old[:, ii] = v + 0.3 * torch.randn(T, B, generator=g).double()
    tr.logp.copy_(lp_old.float()); tr.value.copy_(v_old.float())
    ctx.gae(tr.c, tr.adv, tr.target)
    grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
    ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
    ctx.synchronize()
    adv_o, tgt_o = ON.gae(tr.value.cpu().double(), tr.reward.cpu().double(), done, cfg.gamma, cfg.lam)
    pf = p64.clone().requires_grad_(True)
    lp, v, en, la, lc, *_ = variables(ON.unflatten(pf, H))
    loss, mt = ON.ppo_loss(cfg, lp, v, en, tr.logp.cpu().double()[:, ii], tr.value.cpu().double()[:, ii], adv_o[:, ii], tgt_o[:, ii])
    total = loss + la.mean() + lc.mean()
    total.backward()
    go, gg, mg = pf.grad, grad.cpu().double(), metrics.cpu().double()
    assert float(la.mean()) > 1e-4 and float(lc.mean()) > 1e-6                 # the aux terms are live
    assert abs(mg[8] - float(la.mean())) < 2e-4 * (1 + float(la.mean()))
    assert abs(mg[9] - float(lc.mean())) < 2e-4 * (1 + float(lc.mean()))
    assert abs(mg[0] - float(total)) < 2e-4 * (1 + abs(float(total)))
    off = 0
    for name, shp in ON.param_shapes(H):
        n = int(np.prod(shp))
        a, b = gg[off:off + n], go[off:off + n]
        err = (a - b).abs().max() / (b.abs().max() + 1e-12)
        assert err < 2e-3, (name, float(err), float(b.abs().max()))
        off += n
    assert (gg - go).norm() / go.norm() < 1e-4
    ctx.close()


def test_full_size_minibatch_gradient_is_permutation_invariant():
    """BASELINE minibatch (512 envs x 100 steps, H = 256): reordering the envs inside the minibatch changes row-group membership,
    tile assignment and atomic accumulation order but not the loss or the gradient (beyond fp32 summation order)."""
    N, B, T, H = 1024, 512, 100, 256
    m, cfg, ctx, torch, buffers = _setup(N, B, T, H)
    P = ctx.param_count()
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(2, params)
    tr = _synthetic_traj(torch, buffers, N, T, H, seed=3)
    g = torch.Generator(device="cpu").manual_seed(0)
    tr.logp.copy_(torch.randn(T, N, generator=g) * 0.3 - 20.0)
    tr.value.copy_(torch.randn(T, N, generator=g) * 0.3)
    ctx.gae(tr.c, tr.adv, tr.target)
    idx = torch.randperm(N, generator=g)[:B].int()
    perm = idx[torch.randperm(B, generator=g)]
    out = []
    for ii in (idx, perm):
        grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
        ctx.ppo_grad(params, tr.c, ii.cuda(), B, tr.adv, tr.target, grad, metrics)
        ctx.synchronize()
        out.append((grad.cpu().double(), metrics.cpu().double()))
    (g0, m0), (g1, m1) = out
    assert torch.isfinite(g0).all() and float(g0.norm()) > 0
    assert (g0 - g1).norm() / g0.norm() < 1e-4
    assert (m0 - m1).abs().max() < 1e-4 * (1 + m0.abs().max())
    ctx.close()


@pytest.mark.parametrize("ea,ec,H", [(3, 3, 256), (7, 5, 128), (35, 64, 64)])
def test_user_observation_columns_match_oracle(ea, ec, H):
    """f3 widening: `ea` / `ec` user floats appended behind the reference's 65 / 475 columns of the actor / critic rows
    (kbj_config.extra_obs_*), as a user of the reference appends a term to the lists run_actor / run_critic concatenate (train.py:1351-1433).
    The input projections are [H][65 + ea] / [H][475 + ec]; the policy step and the minibatch gradient follow oracle/nn.py at the widened
    sizes. (3, 3): the actor row keeps its 68-float stride and stays on the fused observation kernels; the others take the general path."""
    N, B, T = 40, 32, 6
    m, cfg, ctx, torch, buffers = _setup(N, B, T, H, extra_obs_actor=ea, extra_obs_critic=ec)
    from oracle import nn as ON
    na, nc, lda, ldc = L.obs_widths(cfg)
    assert (na, nc) == (65 + ea, 475 + ec) and lda % 4 == 0 and ldc % 4 == 0
    P = ctx.param_count()
    assert P == ON.param_count(H, 2, (ea, ec)) and ctx.actor_param_count() == L.param_count(H, 2, (ea, ec))[0]
    params = torch.zeros(P, device="cuda:0")
    ctx.init_params(3, params)
    p64 = params.detach().cpu().double()
    p = ON.unflatten(p64, H, 2, (ea, ec))
    assert p["actor.input_proj.weight"].shape == (H, na) and p["critic.input_proj.weight"].shape == (H, nc)
    assert float(p["actor.input_proj.weight"][:, 65:].abs().max()) > 0            # the user columns have live weights
    jb = torch.tensor(list(m.joint_bias), dtype=torch.float64)
    g = torch.Generator(device="cpu").manual_seed(1)
    # ---- policy step ----
    aobs = torch.zeros(N, lda); aobs[:, :na] = torch.randn(N, na, generator=g)
    cobs = torch.zeros(N, ldc); cobs[:, :nc] = torch.randn(N, nc, generator=g)
    carry = buffers.CarryBuffers(N, H, 2, "cuda:0")
    carry.actor_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5); carry.critic_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.5)
    hc_a0, hc_c0 = carry.actor_hc.cpu().double(), carry.critic_hc.cpu().double()
    action, logp, value = torch.zeros(N, 20, device="cuda:0"), torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
    ctx.policy_step(params, aobs.cuda(), cobs.cuda(), carry.c, 7, 5, True, action, logp, value)
    ctx.synchronize()
    out_a, _ = ON.net_forward(p, "actor", aobs.double(), [[hc_a0[l, 0], hc_a0[l, 1]] for l in range(2)])
    mean, std, _ = ON.actor_head(out_a, aobs.double(), torch.zeros(N, 20, dtype=torch.float64), jb, cfg)
    out_c, _ = ON.net_forward(p, "critic", cobs.double(), [[hc_c0[l, 0], hc_c0[l, 1]] for l in range(2)])
    assert (action.cpu().double() - mean).abs().max() < 2e-5 and (value.cpu().double() - out_c[:, 0]).abs().max() < 2e-5
    # the user columns matter: zeroing them changes the outputs
    a0 = aobs.clone(); a0[:, 65:] = 0
    act0 = torch.zeros_like(action)
    ctx.policy_step(params, a0.cuda(), cobs.cuda(), buffers.CarryBuffers(N, H, 2, "cuda:0").c, 7, 5, True, act0, logp, value)
    ctx.synchronize()
    # ---- minibatch gradient ----
    tr = buffers.TrajBuffers(T, N, H, 2, "cuda:0", ld_actor=lda, ld_critic=ldc)
    tr.actor_obs[:, :, :na] = (torch.randn(T + 1, N, na, generator=g) * 0.5).cuda()
    tr.critic_obs[:, :, :nc] = (torch.randn(T + 1, N, nc, generator=g) * 0.5).cuda()
    tr.action.copy_(torch.randn(T, N, 20, generator=g) * 0.3)
    done = (torch.rand(T, N, generator=g) < 0.15).float() * torch.where(torch.rand(T, N, generator=g) < 0.5, -1.0, 1.0)
    tr.aux[:T, :, L.AUX["DONE"]] = done.cuda()
    tr.reward.copy_(torch.rand(T, N, generator=g))
    tr.carry0_actor_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.3); tr.carry0_critic_hc.copy_(torch.randn(2, 2, N, H, generator=g) * 0.3)
    tr.carry0_lpf.copy_(torch.randn(N, 20, generator=g) * 0.2)
    idx = torch.randperm(N, generator=g)[:B].int()
    ii = idx.long()
    ao, co = tr.actor_obs[:T].cpu().double(), tr.critic_obs[:T].cpu().double()
    act, dn = tr.action.cpu().double(), tr.done.cpu().double()
    with torch.no_grad():
        ca = [[tr.carry0_actor_hc[l, k].cpu().double() for k in range(2)] for l in range(2)]
        cc = [[tr.carry0_critic_hc[l, k].cpu().double() for k in range(2)] for l in range(2)]
        lp, v, en, *_ = ON.ppo_variables(p, cfg, jb, ao, co, act, dn, ca, cc, tr.carry0_lpf.cpu().double())
    tr.logp.copy_((lp + 0.3 * torch.randn(T, N, generator=g).double()).float())
    tr.value.copy_((v + 0.3 * torch.randn(T, N, generator=g).double()).float())
    ctx.gae(tr.c, tr.adv, tr.target)
    grad, metrics = torch.zeros(P, device="cuda:0"), torch.zeros(10, device="cuda:0")
    ctx.ppo_grad(params, tr.c, idx.cuda(), B, tr.adv, tr.target, grad, metrics)
    ctx.synchronize()
    pf = p64.clone().requires_grad_(True)
    pg = ON.unflatten(pf, H, 2, (ea, ec))
    ca = [[tr.carry0_actor_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]
    cc = [[tr.carry0_critic_hc[l, k].cpu().double()[ii] for k in range(2)] for l in range(2)]
    lp, v, en, *_ = ON.ppo_variables(pg, cfg, jb, ao[:, ii], co[:, ii], act[:, ii], dn[:, ii], ca, cc, tr.carry0_lpf.cpu().double()[ii])
    loss, mt = ON.ppo_loss(cfg, lp, v, en, tr.logp.cpu().double()[:, ii], tr.value.cpu().double()[:, ii], tr.adv.cpu().double()[:, ii], tr.target.cpu().double()[:, ii])
    loss.backward()
    gg, go = grad.cpu().double(), pf.grad
    assert abs(float(metrics[0]) - float(loss.detach())) < 2e-4 * (1 + abs(float(loss.detach())))
    off = 0
    for name, shp in ON.param_shapes(H, 2, (ea, ec)):
        n = int(np.prod(shp))
        err = (gg[off:off + n] - go[off:off + n]).abs().max() / (go[off:off + n].abs().max() + 1e-12)
        assert err < 2e-3, (name, float(err))
        if name.endswith("input_proj.weight"):      # the user columns' own gradient block
            k0 = 65 if name.startswith("actor") else 475
            blk_g, blk_o = gg[off:off + n].view(shp)[:, k0:], go[off:off + n].view(shp)[:, k0:]
            assert float(blk_o.abs().max()) > 0 and (blk_g - blk_o).abs().max() / blk_o.abs().max() < 2e-3
        off += n
    assert (gg - go).norm() / go.norm() < 1e-4
    assert not torch.equal(act0, action)
    ctx.close()
