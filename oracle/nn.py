"""CPU oracle of the actor-critic, PPO loss, GAE and AdamW — TEST INFRASTRUCTURE ONLY.

Restates with plain torch ops (CPU, fp32 or fp64, autograd for gradients):
  * Actor / Critic forward               train.py:847-1004 (equinox Linear / LSTMCell: gate order i,f,g,o, single bias)
  * actor head: softplus/clip std, joint-bias + arm-command bias, one-pole low-pass, diagonal Gaussian
                                          train.py:924-941 (ksim.lowpass_one_pole, distrax.MultivariateNormalDiag)
  * per-step PPO variables with carry reset on done     train.py:1435-1508
  * sample_action                         train.py:1545-1572
  * PPO clipped loss, GAE, global-norm clip, AdamW      ksim / optax defaults restated in DESIGN.md (not in the
    reference tree; hyper-parameters train.py:1763-1770, optimizer train.py:1059-1077)
equinox / distrax / optax are not installable offline, so parity with the JAX reference is UNPINNED; the pieces
are pinned against torch.nn.LSTMCell, torch.distributions.Normal and torch.optim.AdamW in tests/test_oracle_nn.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch

NU = 20
NOBS_ACTOR, NOBS_CRITIC = 65, 475
LOG_2PI = math.log(2.0 * math.pi)


def param_shapes(H: int, depth: int = 2, extra_obs=(0, 0)):
    """Flat parameter layout = equinox leaf order of Model(actor, critic) (kbj.h). extra_obs = (actor, critic) user observation columns
    appended behind the reference's 65 / 475 (kbj_config.extra_obs_*): they widen the input projections."""
    shapes = []
    for net, nin, nout in (("actor", NOBS_ACTOR + extra_obs[0], 2 * NU), ("critic", NOBS_CRITIC + extra_obs[1], 1)):
        shapes.append((f"{net}.input_proj.weight", (H, nin)))
        shapes.append((f"{net}.input_proj.bias", (H,)))
        for l in range(depth):
            shapes.append((f"{net}.rnns.{l}.weight_ih", (4 * H, H)))
            shapes.append((f"{net}.rnns.{l}.weight_hh", (4 * H, H)))
            shapes.append((f"{net}.rnns.{l}.bias", (4 * H,)))
        shapes.append((f"{net}.output_proj.weight", (nout, H)))
        shapes.append((f"{net}.output_proj.bias", (nout,)))
    return shapes


def unflatten(flat: torch.Tensor, H: int, depth: int = 2, extra_obs=(0, 0)) -> dict:
    out, off = {}, 0
    for name, shp in param_shapes(H, depth, extra_obs):
        n = int(np.prod(shp))
        out[name] = flat[off:off + n].view(shp)
        off += n
    assert off == flat.numel(), (off, flat.numel())
    return out


def param_count(H: int, depth: int = 2, extra_obs=(0, 0)) -> int:
    return sum(int(np.prod(s)) for _, s in param_shapes(H, depth, extra_obs))


def lstm_cell(x, h, c, w_ih, w_hh, b):
    g = x @ w_ih.T + h @ w_hh.T + b
    H = h.shape[-1]
    i, f, gg, o = g[..., :H], g[..., H:2 * H], g[..., 2 * H:3 * H], g[..., 3 * H:]
    i, f, o, gg = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(gg)
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def net_forward(p: dict, net: str, obs, hc, depth: int = 2):
    """obs [B, >= nin] (columns beyond the input projection's width - the row's pad - are ignored); hc [depth][2][B,H] -> (out [B,nout], new hc)"""
    obs = obs[..., :p[f"{net}.input_proj.weight"].shape[1]]
    x = obs @ p[f"{net}.input_proj.weight"].T + p[f"{net}.input_proj.bias"]
    new = []
    for l in range(depth):
        h, c = lstm_cell(x, hc[l][0], hc[l][1], p[f"{net}.rnns.{l}.weight_ih"], p[f"{net}.rnns.{l}.weight_hh"], p[f"{net}.rnns.{l}.bias"])
        new.append((h, c))
        x = h
    return x @ p[f"{net}.output_proj.weight"].T + p[f"{net}.output_proj.bias"], new


def actor_head(out, obs, lpf, joint_bias, cfg):
    """train.py:924-941 -> (mean after the low-pass, std, new lpf state)"""
    mean = out[..., :NU] + joint_bias
    mean = torch.cat([mean[..., :10], mean[..., 10:] + obs[..., NOBS_ACTOR - 10:NOBS_ACTOR]], -1)
    std = torch.clamp((torch.nn.functional.softplus(out[..., NU:]) + cfg.min_std) * cfg.var_scale, max=cfg.max_std)
    y = lpf + cfg.lpf_alpha * (mean - lpf)
    return y, std, y


def gaussian_logp(a, mean, std):
    return (-0.5 * ((a - mean) / std) ** 2 - torch.log(std) - 0.5 * LOG_2PI).sum(-1)


def gaussian_entropy(std):
    return (0.5 + 0.5 * LOG_2PI + torch.log(std)).sum(-1)


def zero_carry(B, H, depth, dtype):
    return [[torch.zeros(B, H, dtype=dtype), torch.zeros(B, H, dtype=dtype)] for _ in range(depth)]


def ppo_variables(p, cfg, joint_bias, actor_obs, critic_obs, actions, done, carry_a, carry_c, lpf, depth=2):
    """_ppo_scan_fn over time (train.py:1435-1508) for a batch: inputs [T,B,...]; returns logp [T,B], value, entropy, std and final carries."""
    T = actor_obs.shape[0]
    logps, values, ents = [], [], []
    for t in range(T):
        out_a, carry_a = net_forward(p, "actor", actor_obs[t], carry_a, depth)
        mean, std, lpf = actor_head(out_a, actor_obs[t], lpf, joint_bias, cfg)
        logps.append(gaussian_logp(actions[t], mean, std))
        ents.append(gaussian_entropy(std))
        out_c, carry_c = net_forward(p, "critic", critic_obs[t], carry_c, depth)
        values.append(out_c[..., 0])
        keep = (done[t] == 0).to(mean.dtype)[:, None]                 # carry <- initial carry where done (train.py:1502-1506)
        carry_a = [[h * keep, c * keep] for h, c in carry_a]
        carry_c = [[h * keep, c * keep] for h, c in carry_c]
        lpf = lpf * keep
    return torch.stack(logps), torch.stack(values), torch.stack(ents), carry_a, carry_c, lpf


def gae(values, rewards, done, gamma, lam):
    """values/rewards/done [T,N] -> (advantages, targets); bootstrap V_T := V_{T-1}, no bootstrap through done."""
    T = values.shape[0]
    adv = torch.zeros_like(values)
    last = torch.zeros_like(values[0])
    for t in reversed(range(T)):
        keep = (done[t] == 0).to(values.dtype)
        v_next = values[t + 1] if t + 1 < T else values[t]
        delta = rewards[t] + gamma * v_next * keep - values[t]
        last = delta + gamma * lam * keep * last
        adv[t] = last
    return adv, adv + values


def ppo_loss(cfg, logp, value, entropy, logp_old, value_old, adv, target, adv_sums=None):
    """Clipped PPO objective over one minibatch (all tensors [T,B]); returns (loss, metrics dict). adv_sums = (sum adv, sum adv^2, count)
    of the batch to normalise with (the global minibatch of a data-parallel job, kbj_set_advantage_sums); None = this minibatch's own."""
    if adv_sums is None:
        mean, std = adv.mean(), adv.std(unbiased=False)
    else:
        mean = adv_sums[0] / adv_sums[2]
        std = torch.sqrt(torch.clamp(adv_sums[1] / adv_sums[2] - mean * mean, min=0.0))
    a = (adv - mean) / (std + cfg.adv_eps)
    lr = torch.clamp(logp - logp_old, -cfg.log_ratio_clip, cfg.log_ratio_clip)
    ratio = torch.exp(lr)
    surr = torch.minimum(ratio * a, torch.clamp(ratio, 1 - cfg.clip_param, 1 + cfg.clip_param) * a)
    pol = -surr.mean()
    v_clip = value_old + torch.clamp(value - value_old, -cfg.value_clip, cfg.value_clip)
    vl = 0.5 * torch.maximum((value - target) ** 2, (v_clip - target) ** 2).mean()
    ent = entropy.mean()
    loss = pol + cfg.value_loss_coef * vl - cfg.entropy_coef * ent
    clipfrac = ((ratio - 1).abs() > cfg.clip_param).to(value.dtype).mean()
    kl = (logp_old - logp).mean()
    return loss, dict(loss=loss, policy=pol, value=vl, entropy=ent, clipfrac=clipfrac, kl=kl, adv_mean=mean, adv_std=std)


def adamw_step(cfg, p, m, v, g, step, grad_scale=1.0):
    """optax.adamw(lr, b1, b2, eps, weight_decay) after global-norm clipping (in place on numpy/torch arrays)."""
    g = g * grad_scale
    norm = torch.sqrt((g.double() ** 2).sum()).to(g.dtype)
    g = g * torch.clamp(cfg.max_grad_norm / (norm + 1e-6), max=1.0)
    m.mul_(cfg.adam_b1).add_(g, alpha=1 - cfg.adam_b1)
    v.mul_(cfg.adam_b2).addcmul_(g, g, value=1 - cfg.adam_b2)
    mh = m / (1 - cfg.adam_b1 ** step)
    vh = v / (1 - cfg.adam_b2 ** step)
    p.sub_(cfg.learning_rate * (mh / (torch.sqrt(vh) + cfg.adam_eps) + cfg.weight_decay * p))
    return norm


# ---------------------------------------------------------------------------------------------------------------------
# mirror aux losses (train.py:1463-1481, 1574-1756): the packed observation vectors are unpacked to the raw pieces the
# reference mirrors, mirrored exactly as train.py does, and packed again (train.py:1329-1433)
# ---------------------------------------------------------------------------------------------------------------------
def mirror_joints(j):
    """train.py:1574-1582 (legs swapped, arm blocks kept in place, everything negated)."""
    return -torch.cat([j[..., 5:10], j[..., 0:5], j[..., 10:15], j[..., 15:20]], -1)


def _joint_norm(model, dtype):
    bias = torch.tensor(list(model.joint_bias), dtype=dtype)
    lo = torch.tensor(list(model.joint_lo), dtype=dtype)
    hi = torch.tensor(list(model.joint_hi), dtype=dtype)
    return bias, torch.maximum(bias - lo, hi - bias)


def _encode_pg(g):
    roll = torch.atan2(g[..., 1], -g[..., 2])
    pitch = torch.atan2(-g[..., 0], torch.sqrt(g[..., 1] ** 2 + g[..., 2] ** 2))
    return torch.cat([roll[..., None], pitch[..., None], g / g.norm(dim=-1, keepdim=True)], -1)


def _mirror_shared_block(x, model):
    """The first 65 entries of both observation vectors (train.py:1367-1374 / 1410-1416)."""
    bias, rng = _joint_norm(model, x.dtype)
    q = x[..., 0:20] * rng + bias
    v = x[..., 20:40] * 10.0
    g = x[..., 42:45]
    gyro = x[..., 45:48]
    cmd = x[..., 49:65]
    q_m, v_m = mirror_joints(q), mirror_joints(v)
    g_m = torch.stack([g[..., 0], -g[..., 1], g[..., 2]], -1)                       # train.py:1596-1603, 1616-1623
    gyro_m = torch.stack([-gyro[..., 0], gyro[..., 1], -gyro[..., 2]], -1)          # train.py:1588-1595, 1608-1615
    arms = mirror_joints(torch.cat([torch.zeros_like(cmd[..., :10]), cmd[..., 6:16]], -1))[..., 10:20]
    cmd_m = torch.cat([cmd[..., 0:1], -cmd[..., 1:2], -cmd[..., 2:3], cmd[..., 3:4], -cmd[..., 4:5], cmd[..., 5:6], arms], -1)  # :1737-1755
    zc = (cmd_m[..., :3].norm(dim=-1, keepdim=True) < 1e-3).to(x.dtype)
    return torch.cat([(q_m - bias) / rng, v_m / 10.0, _encode_pg(g_m), gyro_m, zc, cmd_m], -1)


def mirror_actor_obs(x, model):
    out = torch.zeros_like(x)
    out[..., :NOBS_ACTOR] = _mirror_shared_block(x, model)
    return out


def mirror_critic_obs(x, model):
    out = torch.zeros_like(x)
    out[..., :65] = _mirror_shared_block(x, model)
    out[..., 65], out[..., 66] = x[..., 66], x[..., 65]                              # touch L <-> R (train.py:1625-1626)
    fp = x[..., 67:73]
    out[..., 67:73] = torch.stack([fp[..., 3], -fp[..., 4], fp[..., 5], fp[..., 0], -fp[..., 1], fp[..., 2]], -1)  # :1627-1637
    out[..., 73:76] = x[..., 73:76]                                                  # base position (train.py:1638)
    bq = x[..., 76:80]
    out[..., 76:80] = torch.stack([bq[..., 0], -bq[..., 1], -bq[..., 2], bq[..., 3]], -1)   # train.py:1639-1647
    ci = x[..., 80:310].reshape(*x.shape[:-1], 23, 10)
    sgn_ci = torch.tensor([1, 1, -1, 1, 1, 1, 1, -1, 1, -1], dtype=x.dtype)          # index pattern of train.py:1652-1665
    out[..., 80:310] = (ci * sgn_ci).reshape(*x.shape[:-1], 230)
    cv = x[..., 310:448].reshape(*x.shape[:-1], 23, 6)
    sgn_cv = torch.tensor([1, -1, 1, -1, 1, -1], dtype=x.dtype)                      # train.py:1670-1689
    out[..., 310:448] = (cv * sgn_cv).reshape(*x.shape[:-1], 138)
    lv, av = x[..., 448:451], x[..., 451:454]
    out[..., 448:451] = torch.stack([lv[..., 0], -lv[..., 1], lv[..., 2]], -1)       # train.py:1691-1698
    out[..., 451:454] = torch.stack([-av[..., 0], av[..., 1], -av[..., 2]], -1)      # train.py:1699-1706
    out[..., 454:474] = mirror_joints(x[..., 454:474])                               # actuator force / 4 (train.py:1708)
    out[..., 474] = x[..., 474]                                                      # base height (train.py:1709)
    return out


def ppo_variables_mirror(p, cfg, model, joint_bias, actor_obs, critic_obs, actions, done, carry_a, carry_c, lpf, carry_am, carry_cm, lpf_m, depth=2):
    """_ppo_scan_fn with the mirror branches (train.py:1435-1508): returns logp, value, entropy, the two aux-loss series and the carries."""
    T = actor_obs.shape[0]
    logps, values, ents, la, lc = [], [], [], [], []
    for t in range(T):
        out_a, carry_a = net_forward(p, "actor", actor_obs[t], carry_a, depth)
        mean, std, lpf = actor_head(out_a, actor_obs[t], lpf, joint_bias, cfg)
        logps.append(gaussian_logp(actions[t], mean, std)); ents.append(gaussian_entropy(std))
        out_c, carry_c = net_forward(p, "critic", critic_obs[t], carry_c, depth)
        values.append(out_c[..., 0])
        ao_m, co_m = mirror_actor_obs(actor_obs[t], model), mirror_critic_obs(critic_obs[t], model)
        out_am, carry_am = net_forward(p, "actor", ao_m, carry_am, depth)
        mean_m, _, lpf_m = actor_head(out_am, ao_m, lpf_m, joint_bias, cfg)
        la.append(((mean - mirror_joints(mean_m)) ** 2).mean(-1) * cfg.actor_mirror_loss_scale)       # train.py:1470-1473
        out_cm, carry_cm = net_forward(p, "critic", co_m, carry_cm, depth)
        lc.append((out_c[..., 0] - out_cm[..., 0]) ** 2 * cfg.critic_mirror_loss_scale)               # train.py:1481
        keep = (done[t] == 0).to(mean.dtype)[:, None]
        carry_a = [[h * keep, c * keep] for h, c in carry_a]; carry_c = [[h * keep, c * keep] for h, c in carry_c]
        carry_am = [[h * keep, c * keep] for h, c in carry_am]; carry_cm = [[h * keep, c * keep] for h, c in carry_cm]
        lpf, lpf_m = lpf * keep, lpf_m * keep
    return (torch.stack(logps), torch.stack(values), torch.stack(ents), torch.stack(la), torch.stack(lc),
            carry_a, carry_c, lpf, carry_am, carry_cm, lpf_m)
