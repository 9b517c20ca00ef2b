"""CPU oracle of one full training iteration (rollout + PPO update) — TEST INFRASTRUCTURE ONLY.

Glues the C++ env oracle (oracle.py) and the torch actor-critic oracle (nn.py) in the same order as the product's
kbj_rollout / kbj_gae / kbj_ppo_grad / kbj_adamw_step. Used by tests (end-to-end parity on tiny sizes), by
__graft_entry__.smoke() and as bench.py's `cpu_baseline` ("port": the reference itself cannot run offline).
Parity with the JAX reference is UNPINNED.
"""
from __future__ import annotations

import numpy as np
import torch

from . import nn as ON
from . import oracle as O


class OracleTrainer:
    def __init__(self, model, cfg, seed: int = 0, params: np.ndarray | None = None, precision: str = "f32"):
        from kbot_joystick_amd.spec import layout as L
        self.L, self.model, self.cfg, self.seed = L, model, cfg, seed
        self.N, self.T, self.H, self.B = cfg.num_envs, cfg.rollout_len, cfg.hidden_size, cfg.batch_size
        self.dt = torch.float32 if precision == "f32" else torch.float64
        self.env = O.Oracle(model, cfg, seed, precision)
        self.D = int(cfg.depth)
        P = ON.param_count(self.H, self.D)
        self.params = torch.zeros(P, dtype=self.dt) if params is None else torch.tensor(params, dtype=self.dt)
        self.m, self.v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.jb = torch.tensor(list(model.joint_bias), dtype=self.dt)
        self.carry_a = ON.zero_carry(self.N, self.H, self.D, self.dt)
        self.carry_c = ON.zero_carry(self.N, self.H, self.D, self.dt)
        self.lpf = torch.zeros(self.N, 20, dtype=self.dt)
        self.obs = self.env.reset_all()
        self.opt_step = 0
        self.iteration = 0

    def _normal(self, step: int) -> np.ndarray:
        """Same draw as actor_head_sample_kernel: threefry(key=(seed ^ ACTION stream, env), ctr=(step, joint)), Box-Muller."""
        z = np.zeros((self.N, 20), np.float32)
        key0 = (self.seed ^ (5 * 0x9E3779B9)) & 0xFFFFFFFF
        for n in range(self.N):
            for j in range(20):
                b0, b1 = O.threefry(key0, self.cfg.env_id_offset + n, step, j)
                u1 = np.float32(((b0 >> 8) + 1) * (1.0 / 16777216.0)); u2 = np.float32((b1 >> 8) * (1.0 / 16777216.0))
                z[n, j] = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
        return z

    def rollout(self, actions: np.ndarray | None = None):
        """Returns dict of trajectory arrays; `actions` [T,N,20] overrides sampling (teacher forcing)."""
        L, T, N = self.L, self.T, self.N
        a_obs = np.zeros((T + 1, N, L.LD_ACTOR), np.float32); c_obs = np.zeros((T + 1, N, L.LD_CRITIC), np.float32)
        aux = np.zeros((T + 1, N, L.AUX["SIZE"]), np.float32)
        act = np.zeros((T, N, 20), np.float32); logp = np.zeros((T, N), np.float32); value = np.zeros((T, N), np.float32)
        a_obs[0], c_obs[0], aux[0] = self.obs
        self.carry0 = ([[x.clone() for x in l] for l in self.carry_a], [[x.clone() for x in l] for l in self.carry_c], self.lpf.clone())
        p = ON.unflatten(self.params, self.H, self.D)
        with torch.no_grad():
            for t in range(T):
                ao, co = torch.tensor(a_obs[t], dtype=self.dt), torch.tensor(c_obs[t], dtype=self.dt)
                out_a, self.carry_a = ON.net_forward(p, "actor", ao[:, :65], self.carry_a, self.D)
                mean, std, self.lpf = ON.actor_head(out_a, ao, self.lpf, self.jb, self.cfg)
                out_c, self.carry_c = ON.net_forward(p, "critic", co[:, :475], self.carry_c, self.D)
                if actions is None:
                    a = mean + std * torch.tensor(self._normal(self.iteration * T + t), dtype=self.dt)
                else:
                    a = torch.tensor(actions[t], dtype=self.dt)
                act[t] = a.float().numpy()
                logp[t] = ON.gaussian_logp(a, mean, std).float().numpy()
                value[t] = out_c[:, 0].float().numpy()
                a_obs[t + 1], c_obs[t + 1], aux[t + 1] = self.env.step(act[t], aux[t])
                keep = torch.tensor((aux[t, :, L.AUX["DONE"]] == 0), dtype=self.dt)[:, None]
                self.carry_a = [[h * keep, c * keep] for h, c in self.carry_a]
                self.carry_c = [[h * keep, c * keep] for h, c in self.carry_c]
                self.lpf = self.lpf * keep
        self.obs = (a_obs[T], c_obs[T], aux[T])
        reward, comps = self.env.rewards(aux[:T])
        self.traj = dict(actor_obs=a_obs, critic_obs=c_obs, aux=aux, action=act, logp=logp, value=value, reward=reward, comps=comps)
        return self.traj

    def minibatch_grad(self, idx: np.ndarray, adv: torch.Tensor, target: torch.Tensor, adv_sums=None):
        L, T = self.L, self.T
        tr = self.traj
        pf = self.params.clone().requires_grad_(True)
        p = ON.unflatten(pf, self.H, self.D)
        ii = torch.as_tensor(idx, dtype=torch.long)
        tt = lambda a: torch.tensor(a, dtype=self.dt)
        done = tt(tr["aux"][:T, :, L.AUX["DONE"]])
        ca = [[x[ii] for x in l] for l in self.carry0[0]]
        cc = [[x[ii] for x in l] for l in self.carry0[1]]
        lp, v, en, *_ = ON.ppo_variables(p, self.cfg, self.jb, tt(tr["actor_obs"][:T])[:, ii], tt(tr["critic_obs"][:T])[:, ii],
                                         tt(tr["action"])[:, ii], done[:, ii], ca, cc, self.carry0[2][ii], self.D)
        loss, metrics = ON.ppo_loss(self.cfg, lp, v, en, tt(tr["logp"])[:, ii], tt(tr["value"])[:, ii], adv[:, ii], target[:, ii], adv_sums=adv_sums)
        loss.backward()
        return pf.grad.detach(), {k: float(x.detach()) for k, x in metrics.items()}

    def update(self, perms: list[np.ndarray], allreduce: str = "per_step", reduce=None):
        """perms: one env permutation per pass (the product draws them with torch.randperm on the host).
        allreduce / reduce mirror HumanoidWalkingTask.update: `reduce(g)` sums g over the data-parallel ranks in place and returns
        the 1/world scale; "per_pass" accumulates a pass's minibatch gradients and takes ONE step per pass."""
        L, T = self.L, self.T
        tr = self.traj
        tt = lambda a: torch.tensor(a, dtype=self.dt)
        adv, target = ON.gae(tt(tr["value"]), tt(tr["reward"]), tt(tr["aux"][:T, :, L.AUX["DONE"]]), self.cfg.gamma, self.cfg.lam)
        last = None
        reduce = reduce or (lambda g: 1.0)
        nmb = self.N // self.B
        for perm in perms:
            acc = torch.zeros_like(self.params)
            for mb in range(nmb):
                g, last = self.minibatch_grad(perm[mb * self.B:(mb + 1) * self.B], adv, target)
                if allreduce == "per_pass":
                    acc += g
                    if mb + 1 < nmb:
                        continue
                    g = acc
                scale = reduce(g)
                if allreduce == "per_pass":
                    scale /= nmb
                self.opt_step += 1
                ON.adamw_step(self.cfg, self.params, self.m, self.v, g, self.opt_step, grad_scale=scale)
        return last

    def train_iteration(self, perms: list[np.ndarray] | None = None, allreduce: str = "per_step", reduce=None):
        self.rollout()
        if perms is None:
            rng = np.random.default_rng(self.seed + self.iteration)
            perms = [rng.permutation(self.N) for _ in range(self.cfg.num_passes)]
        m = self.update(perms, allreduce, reduce)
        self.iteration += 1
        return m
