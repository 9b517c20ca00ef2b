"""Host-side mirror of the reference's task interface for the hot path.

The reference's boundary is ksim's Python Task API: `HumanoidWalkingTask(ksim.PPOTask[Config])`
(train.py:1058) driven by `HumanoidWalkingTask.launch(HumanoidWalkingTaskConfig(...))` (train.py:1759-1792).
This module keeps those names, argument meanings and error behaviour for the path that was rebuilt
(rollout + PPO update), and routes them to libkbj.so. What the reference delegates to ksim/xax outside the
hot path (viewer, TensorBoard, CLI parsing) is out of scope.
"""
from __future__ import annotations

import dataclasses
import math
import os
import time
from dataclasses import dataclass
from typing import Optional

import torch

from ..spec import compiler, constants, layout as L
from . import binding as B
from . import dist as dist_util
from .buffers import CarryBuffers, TrajBuffers


@dataclass
class HumanoidWalkingTaskConfig:
    """Same field names and defaults as the reference config (train.py:73-122 + ksim.PPOConfig fields used in
    train.py:1761-1791). Defaults are the dataclass defaults of the reference, NOT the launch overrides."""
    # model (train.py:78-93)
    hidden_size: int = 128
    depth: int = 2
    var_scale: float = 0.5
    cutoff_frequency: float = 10.0
    # optimizer (train.py:95-114)
    learning_rate: float = 5e-4
    adam_weight_decay: float = 1e-5
    use_lr_decay: bool = False
    lr_decay_steps: int = 19_200_000
    lr_final_multiplier: float = 0.01
    actor_mirror_loss_scale: float = 1.0
    critic_mirror_loss_scale: float = 0.01
    # ksim.PPOConfig fields set by the launch block (train.py:1763-1781)
    num_envs: int = 4096
    batch_size: int = 512
    num_passes: int = 3
    rollout_length_seconds: float = 2.0
    entropy_coef: float = 0.004
    gamma: float = 0.94
    lam: float = 0.94
    dt: float = 0.004
    ctrl_dt: float = 0.02
    iterations: int = 8
    ls_iterations: int = 8
    action_latency_range: tuple = (0.003, 0.01)
    drop_action_prob: float = 0.05
    save_every_n_seconds: Optional[float] = 60
    # build-specific
    robot: str = "kbot"                # train.py:1080 loads robot/kbot; BASELINE configs use kbot-headless
    seed: int = 0
    fixed_command: Optional[tuple] = None   # BASELINE configs[1]: flat-ground fixed joystick velocity command
    terrain: str = "flat"                   # "flat" | "sine" (train.py:1081 loads the "sine" scene; BASELINE configs[4])
    terrain_amplitude: float = 0.05         # metres; the surface definition is this build's own
    terrain_wavelength: float = 2.0
    log_reward_components: bool = False     # keep the 12 unscaled reward terms of every rollout for logging (39 MB at 8192 x 100)

    def to_kbj(self, num_envs_local: int, env_id_offset: int = 0) -> L.Config:
        if self.batch_size <= 0 or num_envs_local % self.batch_size != 0:
            raise ValueError(f"batch_size {self.batch_size} must divide the per-GPU num_envs {num_envs_local}")
        if self.use_lr_decay and self.adam_weight_decay == 0.0:
            # train.py:1074-1075 chains scale_by_adam with scale_by_schedule and no sign flip (gradient ascent as written)
            raise NotImplementedError("use_lr_decay with adam_weight_decay == 0 is not supported")
        T = int(round(self.rollout_length_seconds / self.ctrl_dt))
        kw = dict(num_envs=num_envs_local, env_id_offset=env_id_offset, rollout_len=T, substeps=int(round(self.ctrl_dt / self.dt)),
                  solver_iterations=self.iterations, ls_iterations=self.ls_iterations, hidden_size=self.hidden_size, depth=self.depth,
                  batch_size=self.batch_size, num_passes=self.num_passes, dt=self.dt, ctrl_dt=self.ctrl_dt,
                  latency_lo=self.action_latency_range[0], latency_hi=self.action_latency_range[1],
                  drop_action_prob=self.drop_action_prob, var_scale=self.var_scale, entropy_coef=self.entropy_coef, gamma=self.gamma,
                  lam=self.lam, learning_rate=self.learning_rate, weight_decay=self.adam_weight_decay, switch_prob=self.ctrl_dt / 5,
                  actor_mirror_loss_scale=self.actor_mirror_loss_scale, critic_mirror_loss_scale=self.critic_mirror_loss_scale,
                  lpf_alpha=self.ctrl_dt / (self.ctrl_dt + 1.0 / (2.0 * math.pi * self.cutoff_frequency)))
        if self.terrain not in ("flat", "sine"):
            raise ValueError(f"unknown terrain {self.terrain!r} (flat | sine)")
        if self.terrain == "sine":
            kw.update(terrain_amp=self.terrain_amplitude, terrain_wavelength=self.terrain_wavelength)
        if self.fixed_command is not None:
            cmd = list(self.fixed_command) + [0.0] * (L.NCMD - len(self.fixed_command))
            kw.update(command_mode=1, fixed_command=cmd)
        return L.default_config(**kw)


def cosine_decay_lr(config: HumanoidWalkingTaskConfig, count: int) -> float:
    """optax.cosine_decay_schedule(init_value, decay_steps, alpha) at optimizer-update count `count` (train.py:1068-1072)."""
    frac = min(max(count, 0), config.lr_decay_steps) / config.lr_decay_steps
    cosine = 0.5 * (1.0 + math.cos(math.pi * frac))
    return config.learning_rate * ((1.0 - config.lr_final_multiplier) * cosine + config.lr_final_multiplier)


def launch_config(**overrides) -> HumanoidWalkingTaskConfig:
    """The reference's launch block (train.py:1761-1791)."""
    kw = dict(num_envs=4096, batch_size=512, num_passes=3, rollout_length_seconds=2.0, entropy_coef=0.004, learning_rate=5e-4, gamma=0.94,
              lam=0.94, actor_mirror_loss_scale=0.0, critic_mirror_loss_scale=0.0, hidden_size=256, dt=0.004, ctrl_dt=0.02, iterations=8,
              ls_iterations=8, action_latency_range=(0.003, 0.01), drop_action_prob=0.05, save_every_n_seconds=60)
    kw.update(overrides)
    return HumanoidWalkingTaskConfig(**kw)


class HumanoidWalkingTask:
    """Rollout + PPO update of the K-Bot joystick task on one GPU of a data-parallel job.

    Environments are sharded over ranks (rank r owns global env ids [r*N, (r+1)*N)); the only exchange step is
    the gradient all-reduce before each optimizer step.
    """

    def __init__(self, config: HumanoidWalkingTaskConfig, device: Optional[torch.device] = None, rank: int = 0, world_size: int = 1):
        if not torch.cuda.is_available():
            raise B.KbjError("HumanoidWalkingTask needs a HIP device (no CPU fallback)")
        self.config = config
        self.rank, self.world_size = rank, world_size
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.N, env_off = dist_util.env_shard(config.num_envs, rank, world_size)
        self.kcfg = config.to_kbj(self.N, env_id_offset=env_off)
        self.T, self.H, self.B = self.kcfg.rollout_len, self.kcfg.hidden_size, self.kcfg.batch_size
        self.model_blob = self.get_mujoco_model()
        with torch.cuda.device(self.device):
            self.ctx = B.Context(self.model_blob, self.kcfg, self.device.index or 0, torch.cuda.current_stream().cuda_stream)
        self.P = self.ctx.param_count()
        self.params = self.get_model(config.seed)
        self.opt_m = torch.zeros_like(self.params)
        self.opt_v = torch.zeros_like(self.params)
        self.grad = torch.zeros_like(self.params)
        self.metrics = torch.zeros(10, device=self.device)
        self.mirror = config.actor_mirror_loss_scale != 0.0 or config.critic_mirror_loss_scale != 0.0
        self.carry = self.get_initial_model_carry()
        self.traj = TrajBuffers(self.T, self.N, self.H, self.kcfg.depth, self.device, mirror=self.mirror, reward_comps=config.log_reward_components)
        self.opt_step = 0
        self.iteration = 0
        self._perm_gen = torch.Generator(device="cpu")
        self.ctx.env_reset_all(config.seed, self.traj.actor_obs[self.T], self.traj.critic_obs[self.T], self.traj.aux[self.T])

    # ---- reference API names (train.py:1059-1327, 1510-1572) ----
    def get_mujoco_model(self) -> L.Model:
        """train.py:1079-1081: the compiled robot (a kbj_model blob instead of mujoco.MjModel)."""
        return compiler.load_model(self.config.robot)

    def get_model(self, seed: int) -> torch.Tensor:
        """train.py:1278-1327: Model(actor, critic) as one flat fp32 vector in equinox leaf order."""
        p = torch.zeros(self.P, device=self.device)
        self.ctx.init_params(seed, p)
        return p

    def get_initial_model_carry(self) -> CarryBuffers:
        """train.py:1526-1543: zero LSTM carries and low-pass filter state."""
        return CarryBuffers(self.N, self.H, self.kcfg.depth, self.device, mirror=self.mirror)

    def sample_action(self, actor_obs, critic_obs, step_index: int, argmax: bool = False):
        """train.py:1545-1572 for all envs at one control step; returns (action, log_prob, value)."""
        a = torch.empty(self.N, L.NU, device=self.device)
        lp, v = torch.empty(self.N, device=self.device), torch.empty(self.N, device=self.device)
        self.ctx.policy_step(self.params, actor_obs, critic_obs, self.carry.c, self.config.seed, step_index, argmax, a, lp, v)
        return a, lp, v

    # ---- the hot path ----
    def rollout(self):
        """T control steps of all envs, trajectory + rewards on the device."""
        self.ctx.rollout(self.params, self.carry.c, self.config.seed, self.iteration * self.T, self.traj.c)

    def update(self):
        """GAE, then num_passes x (N / B) minibatch steps: BPTT gradient, all-reduce, AdamW."""
        self.ctx.gae(self.traj.c, self.traj.adv, self.traj.target)
        for p in range(self.kcfg.num_passes):
            self._perm_gen.manual_seed((self.config.seed * 1000003 + self.iteration * 97 + p) & 0x7FFFFFFF)
            perm = torch.randperm(self.N, generator=self._perm_gen).int().to(self.device)
            for mb in range(self.N // self.B):
                idx = perm[mb * self.B:(mb + 1) * self.B].contiguous()
                self.ctx.ppo_grad(self.params, self.traj.c, idx, self.B, self.traj.adv, self.traj.target, self.grad, self.metrics)
                scale = dist_util.allreduce_grad_(self.grad, self.world_size)   # RCCL over xGMI: the one exchange step
                if self.config.use_lr_decay:
                    self.ctx.set_learning_rate(cosine_decay_lr(self.config, self.opt_step))
                self.opt_step += 1
                self.ctx.adamw_step(self.params, self.opt_m, self.opt_v, self.grad, self.opt_step, scale)

    def train_iteration(self):
        self.rollout()
        self.update()
        self.iteration += 1
        self.ctx.synchronize()   # one sync per iteration: surfaces device-side errors (e.g. a recurrence hand-off timeout) as KbjError

    def env_steps_per_iteration(self) -> int:
        return self.N * self.T

    # ---- checkpointing: numpy archive of params/optimizer/env state (xax ckpt.bin layout is "next") ----
    def save_checkpoint(self, path: str):
        import numpy as np
        ep, es = self.ctx.env_get_state()
        extra = {}
        if self.mirror:
            extra = dict(actor_mirror_hc=self.carry.actor_mirror_hc.cpu().numpy(), critic_mirror_hc=self.carry.critic_mirror_hc.cpu().numpy(),
                         lpf_mirror=self.carry.lpf_mirror.cpu().numpy())
        np.savez(path, params=self.params.cpu().numpy(), opt_m=self.opt_m.cpu().numpy(), opt_v=self.opt_v.cpu().numpy(),
                 opt_step=self.opt_step, iteration=self.iteration, ep=ep, es=es, actor_hc=self.carry.actor_hc.cpu().numpy(),
                 critic_hc=self.carry.critic_hc.cpu().numpy(), lpf=self.carry.lpf.cpu().numpy(),
                 config=str(dataclasses.asdict(self.config)), **extra)

    def load_checkpoint(self, path: str):
        import numpy as np
        if not os.path.exists(path):
            raise FileNotFoundError(path)           # convert.py:33-34 error behaviour
        z = np.load(path, allow_pickle=False)
        self.params.copy_(torch.from_numpy(z["params"]))
        self.opt_m.copy_(torch.from_numpy(z["opt_m"])); self.opt_v.copy_(torch.from_numpy(z["opt_v"]))
        self.opt_step, self.iteration = int(z["opt_step"]), int(z["iteration"])
        self.ctx.env_set_state(z["ep"], z["es"])
        self.carry.actor_hc.copy_(torch.from_numpy(z["actor_hc"])); self.carry.critic_hc.copy_(torch.from_numpy(z["critic_hc"]))
        self.carry.lpf.copy_(torch.from_numpy(z["lpf"]))
        if self.mirror:
            self.carry.actor_mirror_hc.copy_(torch.from_numpy(z["actor_mirror_hc"]))
            self.carry.critic_mirror_hc.copy_(torch.from_numpy(z["critic_mirror_hc"]))
            self.carry.lpf_mirror.copy_(torch.from_numpy(z["lpf_mirror"]))

    def export_actor(self, path: str):
        """convert.py's input: the actor's leaves, joint/command order and the flat carry size (host/export.py)."""
        from . import export
        c = self.kcfg
        export.export_actor(path, self.params.cpu().numpy(), self.H, c.depth, c.ctrl_dt, self.config.cutoff_frequency, c.min_std, c.max_std,
                            c.var_scale, list(self.model_blob.joint_bias))

    def reward_components(self):
        """Mean of every unscaled reward term over the last rollout (train.py:1224-1256 order, spec/constants.REWARD_NAMES);
        needs `log_reward_components=True` in the config (kbj_rollout then also writes the [T][N][12] terms)."""
        if self.traj.comps is None:
            raise B.KbjError("reward_components() needs HumanoidWalkingTaskConfig.log_reward_components=True")
        return dict(zip(constants.REWARD_NAMES, self.traj.comps.mean(dim=(0, 1)).cpu().tolist()))

    @classmethod
    def launch(cls, config: HumanoidWalkingTaskConfig, num_iterations: int = 10, log_every: int = 1):
        """train.py:1760: build the task and run the training loop (single process; use bench.py / torchrun for N GPUs)."""
        task = cls(config)
        t0 = time.time()
        for it in range(num_iterations):
            task.train_iteration()
            if (it + 1) % log_every == 0:
                torch.cuda.synchronize()
                m = task.metrics.cpu().tolist()
                rew = float(task.traj.reward.mean())
                print(f"iter {it + 1}: reward/step {rew:.4f} loss {m[0]:.4f} value_loss {m[2]:.4f} entropy {m[3]:.3f} "
                      f"clipfrac {m[4]:.3f} | {task.env_steps_per_iteration() * (it + 1) / (time.time() - t0):.3e} env-steps/s")
        return task
