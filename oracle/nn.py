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


def param_shapes(H: int, depth: int = 2):
    """Flat parameter layout = equinox leaf order of Model(actor, critic) (kbj.h)."""
    shapes = []
    for net, nin, nout in (("actor", NOBS_ACTOR, 2 * NU), ("critic", NOBS_CRITIC, 1)):
        shapes.append((f"{net}.input_proj.weight", (H, nin)))
        shapes.append((f"{net}.input_proj.bias", (H,)))
        for l in range(depth):
            shapes.append((f"{net}.rnns.{l}.weight_ih", (4 * H, H)))
            shapes.append((f"{net}.rnns.{l}.weight_hh", (4 * H, H)))
            shapes.append((f"{net}.rnns.{l}.bias", (4 * H,)))
        shapes.append((f"{net}.output_proj.weight", (nout, H)))
        shapes.append((f"{net}.output_proj.bias", (nout,)))
    return shapes


def unflatten(flat: torch.Tensor, H: int, depth: int = 2) -> dict:
    out, off = {}, 0
    for name, shp in param_shapes(H, depth):
        n = int(np.prod(shp))
        out[name] = flat[off:off + n].view(shp)
        off += n
    assert off == flat.numel(), (off, flat.numel())
    return out


def param_count(H: int, depth: int = 2) -> int:
    return sum(int(np.prod(s)) for _, s in param_shapes(H, depth))


def lstm_cell(x, h, c, w_ih, w_hh, b):
    g = x @ w_ih.T + h @ w_hh.T + b
    H = h.shape[-1]
    i, f, gg, o = g[..., :H], g[..., H:2 * H], g[..., 2 * H:3 * H], g[..., 3 * H:]
    i, f, o, gg = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(gg)
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def net_forward(p: dict, net: str, obs, hc, depth: int = 2):
    """obs [B, nin]; hc [depth][2][B,H] -> (out [B,nout], new hc)"""
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
        out_a, carry_a = net_forward(p, "actor", actor_obs[t][..., :NOBS_ACTOR], carry_a, depth)
        mean, std, lpf = actor_head(out_a, actor_obs[t], lpf, joint_bias, cfg)
        logps.append(gaussian_logp(actions[t], mean, std))
        ents.append(gaussian_entropy(std))
        out_c, carry_c = net_forward(p, "critic", critic_obs[t][..., :NOBS_CRITIC], carry_c, depth)
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


def ppo_loss(cfg, logp, value, entropy, logp_old, value_old, adv, target):
    """Clipped PPO objective over one minibatch (all tensors [T,B]); returns (loss, metrics dict)."""
    a = (adv - adv.mean()) / (adv.std(unbiased=False) + cfg.adv_eps)
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
    return loss, dict(loss=loss, policy=pol, value=vl, entropy=ent, clipfrac=clipfrac, kl=kl, adv_mean=adv.mean(), adv_std=adv.std(unbiased=False))


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
