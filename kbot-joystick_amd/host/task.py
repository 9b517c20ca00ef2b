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

import numpy as np
import torch

from ..spec import compiler, constants, layout as L
from . import binding as B
from . import ckpt as ckpt_io
from . import dist as dist_util
from . import wiring
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
    valid_every_n_steps: Optional[int] = 100      # train.py:1789: deterministic (argmax) validation rollout every n iterations
    valid_every_n_seconds: Optional[float] = None # train.py:1790
    render_length_seconds: float = 10.0           # train.py:1785: length of a validation / view rollout
    max_values_per_plot: int = 50                 # train.py:1783 (plot thinning of the reference's logger; kept for name compatibility)
    render_track_body_id: int = 0                 # train.py:1784: the body the view's camera follows (0, the world, = the base)
    run_mode: str = "train"                       # README.md:66-70 `run_mode=view`: launch() plays the checkpointed policy instead of training (host/view.py)
    # the editable part of get_rewards() / get_commands() (train.py:1206-1256): overrides by reward name
    reward_scales: Optional[dict] = None          # e.g. {"feet_airtime": 2.0, "torque": 0.0}
    reward_params: Optional[dict] = None          # e.g. {"base_height": {"standard_height": 0.85}}
    command_ranges: Optional[dict] = None         # e.g. {"vx_range": (-0.5, 1.5)}; keys as UnifiedCommand's (train.py:1211-1217)
    # the editable part of get_terminations() (train.py:1258-1269): {"bad_z": {"unhealthy_z": 0.4}, "not_upright": {"max_radians": 0.785},
    # "episode_length": {"max_length_sec": 12}}; a threshold of -inf / +inf switches the built-in term off (a Python term may replace it)
    termination_params: Optional[dict] = None
    # build-specific
    robot: str = "kbot"                # train.py:1080 loads robot/kbot; BASELINE configs use kbot-headless
    seed: int = 0
    fixed_command: Optional[tuple] = None   # BASELINE configs[1]: flat-ground fixed joystick velocity command
    # a25: the in-tree samplers (UnifiedCommand train.py:725-752, 782-785; PlaneXYPositionReset train.py:834-836) derive their draws from the key
    # they are called with exactly as jax.random does (split = threefry over a counter, uniform = mantissa fill, bernoulli, randint:
    # kbj_config.command_mode = 2). The key each call RECEIVES is still this build's own (ksim's split tree is un-vendored), and no JAX is in the
    # image to produce fixtures: pinned by jax.random's public known answers only, UNVERIFIED against a live JAX. Off by default.
    jax_random_keys: bool = False
    terrain: str = "flat"                   # "flat" | "sine" (train.py:1081 loads the "sine" scene; BASELINE configs[4])
    terrain_amplitude: float = 0.05         # metres; the surface definition is this build's own
    terrain_wavelength: float = 2.0
    log_reward_components: bool = False     # keep the 12 unscaled reward terms of every rollout for logging (39 MB at 8192 x 100)
    # keep qpos / qvel of every env-step (kbj_traj.qstate_d, 262 MB at 8192 x 100): the Python reward terms (extra_rewards) then see a
    # ksim-shaped `host.trajectory.Trajectory` - trajectory.qpos / .qvel / .xpos / .xquat / .obs[...] / .command["unified_command"] as the
    # reference's reward classes read them (train.py:138-506) - instead of the narrower TrajectoryView of the aux record
    record_state: bool = False
    # data-parallel exchange: "per_step" = all-reduce the gradient before every optimizer step (the
    # single-GPU-equivalent default); "per_pass" = north_star's "once per update" variant: accumulate the minibatch gradients of
    # a pass locally, ONE all-reduce and ONE optimizer step per pass (fewer, larger steps - a different algorithm). KBJ_ALLREDUCE
    # in the environment overrides the default.
    allreduce: str = dataclasses.field(default_factory=lambda: os.environ.get("KBJ_ALLREDUCE", "per_step"))
    # per_step exchange only: all-reduce the actor's gradient slice on a second stream under the critic's tail of kbj_ppo_grad
    # (kbj_stream_wait_actor_grad), the critic's slice behind the call. Same result; pays only where the exchange is visible (N > 1)
    overlap_allreduce: bool = dataclasses.field(default_factory=lambda: os.environ.get("KBJ_OVERLAP_ALLREDUCE", "0") not in ("0", ""))
    # bit-reproducible update: fixed-order reductions instead of fp32 / fp64 atomics in the gradient (kbj_config.deterministic); the
    # reference's XLA program is deterministic by default, here it costs a few percent and is off unless asked for
    deterministic: bool = dataclasses.field(default_factory=lambda: os.environ.get("KBJ_DETERMINISTIC", "0") not in ("0", ""))
    # NOT the default, never the headline: the update's large backward GEMMs through the exact three-way bf16 operand split (kbj_config.gemm_bf16x3,
    # DESIGN.md section 10b): fp32 in, fp32 accumulate, fp32 out, every product exact up to 2^-23; faster and measured more accurate than the fp32 MFMA chain
    gemm_bf16x3: bool = dataclasses.field(default_factory=lambda: os.environ.get("KBJ_GEMM_X3", "0") not in ("0", ""))
    # user observations INTO the network rows (the reference's user appends terms to the lists run_actor / run_critic
    # concatenate, train.py:1351-1433): that many floats are appended behind the reference's 65 / 475 columns of every observation row, the input
    # projections and the parameter vector grow accordingly (kbj_config.extra_obs_*). Which terms fill them: HumanoidWalkingTask(
    # extra_observations={name: (term, "actor" | "critic" | "both")}); the widths must add up. 0..64 each.
    extra_actor_obs: int = 0
    extra_critic_obs: int = 0
    # data-parallel variant: normalise a minibatch's advantages with the mean / variance of the GLOBAL minibatch (all
    # ranks' minibatches of that step: three scalars all-reduced per step) instead of each rank's own. With it the averaged gradient equals the
    # single-process gradient over the union of the ranks' minibatches. Off by default (what ksim does across devices is not visible).
    global_advantage_stats: bool = dataclasses.field(default_factory=lambda: os.environ.get("KBJ_GLOBAL_ADV_STATS", "0") not in ("0", ""))
    # use_lr_decay with adam_weight_decay == 0 is train.py:1074-1075 AS WRITTEN: optax.chain(scale_by_adam(), scale_by_schedule(cosine)) has
    # no sign flip, i.e. gradient ASCENT. Served only with this explicit opt-in (recorded in the checkpoint's config member); without it
    # that combination is an error.
    reproduce_reference_lr_sign: bool = False

    def to_kbj(self, num_envs_local: int, env_id_offset: int = 0) -> L.Config:
        if self.batch_size <= 0 or num_envs_local % self.batch_size != 0:
            raise ValueError(f"batch_size {self.batch_size} must divide the per-GPU num_envs {num_envs_local}")
        # use_lr_decay with adam_weight_decay == 0 is train.py:1074-1075: optax.chain(scale_by_adam(), scale_by_schedule(cosine_schedule)).
        # AS WRITTEN that chain has no sign flip (optax.adam ends in scale_by_learning_rate = scale(-lr); scale_by_schedule does not), so
        # the parameters move ALONG the Adam direction: p += lr_t * m / (sqrt(v) + eps). It is served as written - kbj_adamw_step with
        # weight_decay 0 and the learning rate -cosine_decay_lr (update()) - behind reproduce_reference_lr_sign only; every rank says so at
        # construction and update() warns once, because it is gradient ascent.
        if self.use_lr_decay and self.adam_weight_decay == 0.0 and not self.reproduce_reference_lr_sign:
            raise ValueError("use_lr_decay with adam_weight_decay == 0 is train.py:1074-1075: optax.chain(scale_by_adam, scale_by_schedule) "
                             "has no sign flip, so the update ASCENDS the loss. Set reproduce_reference_lr_sign=True to run it as written, "
                             "or use a non-zero adam_weight_decay (the adamw branch, train.py:1076-1077)")
        T = int(round(self.rollout_length_seconds / self.ctrl_dt))
        kw = dict(num_envs=num_envs_local, env_id_offset=env_id_offset, rollout_len=T, substeps=int(round(self.ctrl_dt / self.dt)),
                  solver_iterations=self.iterations, ls_iterations=self.ls_iterations, hidden_size=self.hidden_size, depth=self.depth,
                  batch_size=self.batch_size, num_passes=self.num_passes, dt=self.dt, ctrl_dt=self.ctrl_dt,
                  latency_lo=self.action_latency_range[0], latency_hi=self.action_latency_range[1],
                  drop_action_prob=self.drop_action_prob, var_scale=self.var_scale, entropy_coef=self.entropy_coef, gamma=self.gamma,
                  lam=self.lam, learning_rate=self.learning_rate, weight_decay=self.adam_weight_decay, switch_prob=self.ctrl_dt / 5,
                  actor_mirror_loss_scale=self.actor_mirror_loss_scale, critic_mirror_loss_scale=self.critic_mirror_loss_scale,
                  lpf_alpha=self.ctrl_dt / (self.ctrl_dt + 1.0 / (2.0 * math.pi * self.cutoff_frequency)), deterministic=int(bool(self.deterministic)),
                  extra_obs_actor=int(self.extra_actor_obs), extra_obs_critic=int(self.extra_critic_obs), gemm_bf16x3=int(bool(self.gemm_bf16x3)))
        if not (0 <= self.extra_actor_obs <= L.MAX_EXTRA_OBS and 0 <= self.extra_critic_obs <= L.MAX_EXTRA_OBS):
            raise ValueError(f"extra_actor_obs / extra_critic_obs must be in 0..{L.MAX_EXTRA_OBS}")
        if self.allreduce not in ("per_step", "per_pass"):
            raise ValueError(f"unknown allreduce mode {self.allreduce!r} (per_step | per_pass)")
        if self.terrain not in ("flat", "sine"):
            raise ValueError(f"unknown terrain {self.terrain!r} (flat | sine)")
        if self.terrain == "sine":
            kw.update(terrain_amp=self.terrain_amplitude, terrain_wavelength=self.terrain_wavelength)
        if self.jax_random_keys:
            kw.update(command_mode=2)           # (a fixed command below overrides the sampler, not the reset's key handling... which command_mode 1 switches off too)
        if self.fixed_command is not None:
            if self.jax_random_keys:
                raise ValueError("jax_random_keys selects the UnifiedCommand SAMPLER with jax.random's key handling; it cannot be combined with fixed_command")
            cmd = list(self.fixed_command) + [0.0] * (L.NCMD - len(self.fixed_command))
            kw.update(command_mode=1, fixed_command=cmd)
        for k, (lo, hi) in (self.command_ranges or {}).items():
            if k not in ("vx_range", "vy_range", "wz_range", "bh_range", "rx_range", "ry_range"):
                raise KeyError(f"unknown command range {k!r}")
            kw[k[:2] + "_lo"], kw[k[:2] + "_hi"] = float(lo), float(hi)
        for name, kv in (self.termination_params or {}).items():
            for k, v in kv.items():
                v = float(v)
                if (name, k) == ("bad_z", "unhealthy_z"):
                    kw["unhealthy_z"] = max(v, -3.0e38)
                elif (name, k) == ("not_upright", "max_radians"):
                    kw["max_tilt_rad"] = min(v, math.pi)
                elif (name, k) == ("episode_length", "max_length_sec"):
                    kw["max_episode_steps"] = int(min(round(v / self.ctrl_dt), 2 ** 30))     # round like T and substeps: 2.3 / 0.02 is 115, not 114
                else:
                    raise KeyError(f"unknown termination parameter {name}.{k} (bad_z.unhealthy_z | not_upright.max_radians | episode_length.max_length_sec)")
        kcfg = L.default_config(**kw)
        wiring.apply_reward_overrides(kcfg, self.reward_scales, self.reward_params)
        return kcfg


def cosine_decay_lr(config: HumanoidWalkingTaskConfig, count: int) -> float:
    """optax.cosine_decay_schedule(init_value, decay_steps, alpha) at optimizer-update count `count` (train.py:1068-1072)."""
    frac = min(max(count, 0), config.lr_decay_steps) / config.lr_decay_steps
    cosine = 0.5 * (1.0 + math.cos(math.pi * frac))
    return config.learning_rate * ((1.0 - config.lr_final_multiplier) * cosine + config.lr_final_multiplier)


def launch_config(**overrides) -> HumanoidWalkingTaskConfig:
    """The reference's launch block (train.py:1761-1791)."""
    kw = dict(num_envs=4096, batch_size=512, num_passes=3, rollout_length_seconds=2.0, entropy_coef=0.004, learning_rate=5e-4, gamma=0.94,
              lam=0.94, actor_mirror_loss_scale=0.0, critic_mirror_loss_scale=0.0, hidden_size=256, dt=0.004, ctrl_dt=0.02, iterations=8,
              ls_iterations=8, action_latency_range=(0.003, 0.01), drop_action_prob=0.05, render_track_body_id=0, render_length_seconds=10,
              max_values_per_plot=50, save_every_n_seconds=60, valid_every_n_steps=100, valid_every_n_seconds=None)
    kw.update(overrides)
    return HumanoidWalkingTaskConfig(**kw)


_NO_PREFETCH = os.environ.get("KBJ_NO_PREFETCH", "0") not in ("0", "")     # A/B switch: kbj_ppo_prefetch is a pure scheduling hint


class HumanoidWalkingTask:
    """Rollout + PPO update of the K-Bot joystick task on one GPU of a data-parallel job.

    Environments are sharded over ranks (rank r owns global env ids [r*N, (r+1)*N)); the only exchange step is
    the gradient all-reduce before each optimizer step.
    """

    def __init__(self, config: HumanoidWalkingTaskConfig, device: Optional[torch.device] = None, rank: int = 0, world_size: int = 1,
                 extra_rewards: Optional[dict] = None, extra_terminations: Optional[dict] = None, extra_observations: Optional[dict] = None,
                 command=None, extra_resets: Optional[list] = None):
        """extra_rewards: {name: term} of Python reward terms in ksim's Reward protocol (`scale`, `get_reward(trajectory)` or the stateful
        pair), evaluated on `TrajectoryView` after every rollout and added to the built-in stack's reward (host/traj_view.py).
        extra_terminations: {name: term}, `term(state, curriculum_level) -> [N] in {-1, 0, 1}` (train.py:817) on a `StepView` after every
        control step, OR-ed into the kernel's own terminations: the env is reset (kbj_env_reset_where), the model carries with it, GAE is
        cut there. extra_observations: {name: term}, `term.observe(state, curriculum_level, rng)` or a plain callable -> [N, d]
        (train.py:635, 682, 706), evaluated per control step and kept as [T + 1, N, d] tensors for the Python reward / termination terms
        (`TrajectoryView.extra_observations[name]`); the networks' input rows stay the reference's 65 / 475 floats.
        command: ONE term in the reference's Command protocol (train.py:724, 768) that replaces the built-in UnifiedCommand sampler:
        `command.initial_command(state, curriculum_level, rng) -> [N, 16]` for the envs that have just been reset and
        `command(prev_command, state, curriculum_level, rng) -> [N, 16]` for the running ones, evaluated on the device after every
        control step and written into the env state and the next observation rows by kbj_env_set_command (`rng` is a torch.Generator of
        the task's device, seeded from (seed, step index)). The library then runs with command_mode = 1 so its own switch draw stays off.
        extra_resets: a LIST of terms in the reference's Reset protocol (train.py:833-844: `term(data, curriculum_level, rng) -> data` with
        `data.qpos` / `data.qvel`), applied in list order - behind the built-in resets of train.py:1146-1153, as further entries of that list would
        be - to the envs that have just been re-initialised; the result is written back by kbj_env_set_qstate, which also rewrites their next
        observation rows (host/traj_view.ResetData; `rng` is a torch.Generator seeded from (seed, step index)).
        With any of these the rollout runs step by step from the host (policy step, env step, user terms, carry reset: the same calls
        kbj_rollout fuses, bit-identical when no user term fires) instead of as one kbj_rollout call."""
        self.extra_rewards = dict(extra_rewards or {})
        self.extra_terminations = dict(extra_terminations or {})
        # an observation term may be given as (term, into) with into in {"actor", "critic", "both"}: its output is then ALSO written into the
        # networks' input rows, behind the reference's columns (config.extra_actor_obs / extra_critic_obs floats, in dictionary order)
        self.extra_observations, self.extra_obs_into = {}, {}
        for name, term in (extra_observations or {}).items():
            into = None
            if isinstance(term, tuple):
                term, into = term
            into = into if into is not None else getattr(term, "into", None)
            if into not in (None, "actor", "critic", "both"):
                raise ValueError(f"observation {name!r}: into must be 'actor', 'critic', 'both' or None, not {into!r}")
            self.extra_observations[name], self.extra_obs_into[name] = term, into
        self._obs_started = False
        self.command_term = command
        self._command_started = False
        self.extra_resets = list(extra_resets or [])
        self._resets_started = False
        self.extra_obs_buffers: dict = {}
        self._extra_carries: dict = {}
        self.extra_reward_means: dict = {}
        if not torch.cuda.is_available():
            raise B.KbjError("HumanoidWalkingTask needs a HIP device (no CPU fallback)")
        self.config = config
        self.rank, self.world_size = rank, world_size
        if config.run_mode not in ("train", "view"):
            raise ValueError(f"run_mode must be 'train' or 'view' (README.md:66-70), not {config.run_mode!r}")
        if config.use_lr_decay and config.adam_weight_decay == 0.0 and config.reproduce_reference_lr_sign:
            import sys                       # every rank, at construction: a single warnings.warn is easy to lose in a multi-rank log
            print(f"[kbj rank {rank}] reproduce_reference_lr_sign: optax.chain(scale_by_adam, scale_by_schedule) as written in "
                  "train.py:1074-1075 has no sign flip - this run ASCENDS the loss", file=sys.stderr, flush=True)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.N, env_off = dist_util.env_shard(config.num_envs, rank, world_size)
        self.kcfg = config.to_kbj(self.N, env_id_offset=env_off)
        if command is not None:
            self.kcfg.command_mode = 1          # resets write fixed_command (zeros unless configured); the term overwrites it every step
        self.T, self.H, self.B = self.kcfg.rollout_len, self.kcfg.hidden_size, self.kcfg.batch_size
        self.model_blob = self.get_mujoco_model()
        torch.cuda.set_device(self.device)     # torch ops of this task and the library's launches must target the same GPU
        self.ctx = B.Context(self.model_blob, self.kcfg, self.device.index or 0, torch.cuda.current_stream().cuda_stream)
        self.P = self.ctx.param_count()
        self.params = self.get_model(config.seed)
        self.opt_m = torch.zeros_like(self.params)
        self.opt_v = torch.zeros_like(self.params)
        self.grad = torch.zeros_like(self.params)
        self.grad_acc = torch.zeros_like(self.params) if config.allreduce == "per_pass" else None
        self.metrics = torch.zeros(10, device=self.device)
        self.mirror = config.actor_mirror_loss_scale != 0.0 or config.critic_mirror_loss_scale != 0.0
        self.carry = self.get_initial_model_carry()
        self.nobs_actor, self.nobs_critic, self.ld_actor, self.ld_critic = L.obs_widths(self.kcfg)
        self.extra_obs = (int(self.kcfg.extra_obs_actor), int(self.kcfg.extra_obs_critic))
        if any(self.extra_obs) and not any(v for v in self.extra_obs_into.values()):
            raise ValueError("config.extra_actor_obs / extra_critic_obs reserve network inputs, but no observation term is routed into them "
                             "(extra_observations={name: (term, 'actor' | 'critic' | 'both')})")
        self.traj = TrajBuffers(self.T, self.N, self.H, self.kcfg.depth, self.device, mirror=self.mirror, reward_comps=config.log_reward_components,
                                ld_actor=self.ld_actor, ld_critic=self.ld_critic, record_state=config.record_state)
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
        if self.extra_terminations or self.extra_observations or self.command_term is not None or self.extra_resets:
            self._rollout_stepwise()
        else:
            self.ctx.rollout(self.params, self.carry.c, self.config.seed, self.iteration * self.T, self.traj.c)
        if self.extra_rewards:
            from .traj_view import apply_extra_rewards
            self.extra_reward_means = apply_extra_rewards(self.extra_rewards, self._extra_carries, self.trajectory(), self.traj.reward)

    def trajectory(self):
        """The last rollout as the Python reward terms see it: with `config.record_state` a ksim-shaped `host.trajectory.Trajectory`
        (trajectory.qpos / .qvel / .xpos / .xquat / .ctrl / .done / .obs[...] / .command["unified_command"], train.py:138-506), else the
        `TrajectoryView` of the aux record. User observation terms appear under their names in `.obs` / `.extra_observations`."""
        extra = {k: v[:self.T] for k, v in self.extra_obs_buffers.items()}
        if self.config.record_state:
            from .trajectory import Trajectory
            view = Trajectory(self.traj, self.T, self.model_blob, extra_observations=extra)
        else:
            from .traj_view import TrajectoryView
            view = TrajectoryView(self.traj, self.T)
        view.extra_observations = extra
        return view

    def _apply_command(self, ctx, tr, row: int, view, fresh, step_index: int, all_fresh: bool = False):
        """The user's Command term for observation row `row` of `tr` (train.py:724, 768): `initial_command` for the envs whose episode
        starts there (`fresh` [N] bool), `__call__(prev_command, ...)` for the others, written by kbj_env_set_command. `all_fresh`: the
        caller KNOWS every env starts an episode (rows written by a reset of all envs); otherwise both are evaluated and selected on the
        device - asking `fresh.all()` would drain the queue once per control step."""
        term, n = self.command_term, tr.aux[row].shape[0]
        g = torch.Generator(device=self.device)
        g.manual_seed((self.config.seed * 2654435761 + step_index * 40503 + self.rank) & 0x7FFFFFFFFFFFFFFF)
        prev = tr.aux[row][:, L.AUX["CMD"]:L.AUX["CMD"] + L.NCMD].clone()
        new = term.initial_command(view, 1.0, g).reshape(n, L.NCMD).to(torch.float32)
        if not all_fresh:
            new = torch.where(fresh[:, None], new, term(prev, view, 1.0, g).reshape(n, L.NCMD).to(torch.float32))
        ctx.env_set_command(None, new.contiguous(), tr.actor_obs[row], tr.critic_obs[row], tr.aux[row])

    def _apply_resets(self, ctx, tr, row: int, fresh, step_index: int):
        """The user's Reset terms (train.py:833-844) for observation row `row` of `tr`: evaluated on every env's (qpos, qvel), kept for the envs
        whose episode starts there (`fresh` [N] bool, None = all), written back - with the rows of the next observation - by kbj_env_set_qstate."""
        from .traj_view import ResetData
        n = tr.aux[row].shape[0]
        qpos, qvel = torch.empty(n, L.NQ, device=self.device), torch.empty(n, L.NV, device=self.device)
        ctx.env_get_qstate(qpos, qvel)
        g = torch.Generator(device=self.device)
        g.manual_seed((self.config.seed * 2246822519 + step_index * 3266489917 + self.rank) & 0x7FFFFFFFFFFFFFFF)
        data = ResetData(qpos, qvel)
        for term in self.extra_resets:
            data = term(data, 1.0, g)
            # a Reset term returns the data it was given with fields replaced (train.py:833-844): same shapes, finite numbers. Checked here, by name,
            # on the host (this is the step-wise path of user terms, not the fused rollout): the kernel would carry a NaN into the observation
            # rows without a word, and a broadcastable wrong shape would only surface as a reshape error further down.
            for field, width in (("qpos", L.NQ), ("qvel", L.NV)):
                v = getattr(data, field, None)
                if not torch.is_tensor(v) or tuple(v.shape) != (n, width):
                    raise B.KbjError(f"Reset term {type(term).__name__} returned {field} of shape {tuple(v.shape) if torch.is_tensor(v) else type(v).__name__}, "
                                     f"expected ({n}, {width})")
                if not bool(torch.isfinite(v).all()):
                    raise B.KbjError(f"Reset term {type(term).__name__} returned non-finite values in {field}")
        new_q = data.qpos.to(torch.float32).contiguous()
        new_v = data.qvel.to(torch.float32).contiguous()
        self._reset_keep = (new_q, new_v)          # alive until the kernel has run (stream-ordered)
        mask = None if fresh is None else fresh.to(torch.float32)
        ctx.env_set_qstate(mask, new_q, new_v, tr.actor_obs[row], tr.critic_obs[row], tr.aux[row])

    def _observe_into_rows(self, tr, row: int, view) -> dict:
        """Evaluate the user's Observation terms (train.py:635, 682, 706 protocol) on `view` and write those routed into the networks behind the
        reference's columns of observation row `row` (actor: from column 65, critic: from 475, dictionary order). Returns {name: [N, d]}."""
        off = {"actor": L.NOBS_ACTOR, "critic": L.NOBS_CRITIC}
        rows = {"actor": tr.actor_obs[row], "critic": tr.critic_obs[row]}
        lim = {"actor": self.nobs_actor, "critic": self.nobs_critic}
        out = {}
        for name, term in self.extra_observations.items():
            v = term.observe(view, 1.0, None) if hasattr(term, "observe") else term(view)
            v = v.reshape(view.N, -1).to(torch.float32)
            out[name] = v
            into = self.extra_obs_into.get(name)
            for net in (("actor", "critic") if into == "both" else (into,) if into else ()):
                if off[net] + v.shape[1] > lim[net]:
                    raise B.KbjError(f"observation {name!r} ({v.shape[1]} floats) does not fit the {net} row: config.extra_{net}_obs reserves "
                                     f"{lim[net] - (L.NOBS_ACTOR if net == 'actor' else L.NOBS_CRITIC)} floats")
                rows[net][:, off[net]:off[net] + v.shape[1]] = v
                off[net] += v.shape[1]
        for net in ("actor", "critic"):
            if any(i in (net, "both") for i in self.extra_obs_into.values()) and off[net] != lim[net]:
                raise B.KbjError(f"the observation terms routed into the {net} fill {off[net] - (L.NOBS_ACTOR if net == 'actor' else L.NOBS_CRITIC)} "
                                 f"floats, config.extra_{net}_obs says {lim[net] - (L.NOBS_ACTOR if net == 'actor' else L.NOBS_CRITIC)}")
        return out

    def _rollout_stepwise(self):
        """kbj_rollout's steps as separate ABI calls with the user's Termination / Observation terms between the env step and the carry
        reset (train.py:817, 635 protocols on host/traj_view.StepView)."""
        from .traj_view import StepView, combine_terminations
        T, tr, c = self.T, self.traj, self.ctx
        done_col = 4 * L.AUX["DONE"]
        tr.actor_obs[0].copy_(tr.actor_obs[T]); tr.critic_obs[0].copy_(tr.critic_obs[T]); tr.aux[0].copy_(tr.aux[T])   # row T of the last rollout is row 0 of this one
        tr.carry0_actor_hc.copy_(self.carry.actor_hc); tr.carry0_critic_hc.copy_(self.carry.critic_hc); tr.carry0_lpf.copy_(self.carry.lpf)
        if self.mirror:
            tr.carry0_actor_mirror_hc.copy_(self.carry.actor_mirror_hc); tr.carry0_critic_mirror_hc.copy_(self.carry.critic_mirror_hc)
            tr.carry0_lpf_mirror.copy_(self.carry.lpf_mirror)
        first = self.iteration * T

        def observe(row: int, view):
            vals = self._observe_into_rows(tr, row, view)
            for name, v in vals.items():
                if name not in self.extra_obs_buffers:
                    self.extra_obs_buffers[name] = torch.zeros(T + 1, self.N, v.shape[1], device=self.device)
                self.extra_obs_buffers[name][row].copy_(v)

        def update_command(row: int, view, fresh, all_fresh: bool = False):
            self._apply_command(c, tr, row, view, fresh, first + row, all_fresh)

        if self.extra_resets and not self._resets_started:                     # the state env_reset_all drew: every env starts an episode
            self._apply_resets(c, tr, 0, None, first)
            self._resets_started = True
        if self.command_term is not None and not self._command_started:      # the rows env_reset_all wrote: every env starts an episode
            view0 = StepView(tr.aux[0], tr.actor_obs[0], tr.critic_obs[0], tr.aux[0], self.model_blob)
            update_command(0, view0, torch.ones(self.N, dtype=torch.bool, device=self.device), all_fresh=True)
            self._command_started = True
        for name, buf in self.extra_obs_buffers.items():
            buf[0].copy_(buf[T])
        if self.extra_observations and not self._obs_started:                 # the rows env_reset_all wrote carry zeros in the user columns
            observe(0, StepView(tr.aux[0], tr.actor_obs[0], tr.critic_obs[0], tr.aux[0], self.model_blob))
            self._obs_started = True
        for t in range(T):
            c.policy_step(self.params, tr.actor_obs[t], tr.critic_obs[t], self.carry.c, self.config.seed, first + t, False, tr.action[t], tr.logp[t], tr.value[t])
            c.env_step(tr.action[t], tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1], tr.qstate[t] if tr.qstate is not None else None)
            qs_t = tr.qstate[t] if tr.qstate is not None else None
            view = StepView(tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1], self.model_blob, qs_t)
            if self.extra_terminations:
                user = combine_terminations(self.extra_terminations, view)
                fire = (user != 0) & (tr.aux[t][:, L.AUX["DONE"]] == 0)          # the kernel's own terminations already reset their envs
                c.env_reset_where(fire.to(torch.float32), tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1])
                tr.aux[t][:, L.AUX["DONE"]] = torch.where(fire, user, tr.aux[t][:, L.AUX["DONE"]])
                view = StepView(tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1], self.model_blob, qs_t)
            if self.extra_resets:                                   # the envs this step finished carry their fresh episode's state: the user's Reset terms on top
                self._apply_resets(c, tr, t + 1, tr.aux[t][:, L.AUX["DONE"]] != 0, first + t + 1)
                view = StepView(tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1], self.model_blob, qs_t)
            if self.command_term is not None:
                update_command(t + 1, view, tr.aux[t][:, L.AUX["DONE"]] != 0)
            if self.extra_observations:
                observe(t + 1, view)
            c.carry_reset(self.carry.c, tr.aux[t].data_ptr() + done_col, L.AUX["SIZE"])
        c.rewards(tr.aux, T, tr.reward, tr.comps)

    def update(self):
        """GAE, then num_passes x (N / B) minibatch steps: BPTT gradient, all-reduce, AdamW."""
        # the passes' permutations: drawn on the host (same streams as the oracle trainer), staged in pinned memory and uploaded with
        # stream-ordered copies BEFORE anything of the update is enqueued - a pageable `.to(device)` per pass blocks the host until the
        # queued work has drained and leaves the GPU idle (~0.3 ms) while the next minibatch is being enqueued
        npass = self.kcfg.num_passes
        if getattr(self, "_perm_pinned", None) is None or self._perm_pinned.shape != (npass, self.N):
            self._perm_pinned = torch.empty(npass, self.N, dtype=torch.int32).pin_memory()
            self._perm_dev = torch.empty(npass, self.N, dtype=torch.int32, device=self.device)
            self._perm_event = torch.cuda.Event()
        else:
            self._perm_event.synchronize()       # the previous upload has left the staging buffer (it has, unless the caller never syncs)
        for p in range(npass):
            self._perm_gen.manual_seed((self.config.seed * 1000003 + self.iteration * 97 + p) & 0x7FFFFFFF)
            self._perm_pinned[p].copy_(torch.randperm(self.N, generator=self._perm_gen))
        self._perm_dev.copy_(self._perm_pinned, non_blocking=True)
        self._perm_event.record()
        self.ctx.gae(self.traj.c, self.traj.adv, self.traj.target)
        for p in range(npass):
            perm = self._perm_dev[p]
            nmb = self.N // self.B
            per_pass = self.config.allreduce == "per_pass"
            if per_pass:
                self.grad_acc.zero_()
            for mb in range(nmb):
                idx = perm[mb * self.B:(mb + 1) * self.B].contiguous()
                if self.config.global_advantage_stats:
                    self._adv_sums = dist_util.global_advantage_sums(self.traj.adv.index_select(1, idx.long()), self.world_size)   # kept alive until the call has run
                    self.ctx.set_advantage_sums(self._adv_sums)
                self.ctx.ppo_grad(self.params, self.traj.c, idx, self.B, self.traj.adv, self.traj.target, self.grad, self.metrics)
                # next-minibatch hint: its parameter-independent gathers run under this minibatch's exchange + optimizer step
                nxt = perm[(mb + 1) * self.B:(mb + 2) * self.B] if mb + 1 < nmb else (self._perm_dev[p + 1][:self.B] if p + 1 < npass else None)
                if nxt is not None and not _NO_PREFETCH:
                    self.ctx.ppo_prefetch(self.traj.c, nxt.contiguous())
                if per_pass:
                    self.grad_acc.add_(self.grad)          # kbj_ppo_grad overwrites `grad`; the pass total lives in grad_acc
                    if mb + 1 < nmb:
                        continue
                g = self.grad_acc if per_pass else self.grad
                if self.config.overlap_allreduce and not per_pass:
                    if getattr(self, "_comm_stream", None) is None:
                        self._comm_stream = torch.cuda.Stream(device=self.device)
                    scale = dist_util.allreduce_grad_overlapped_(self.ctx, g, self.ctx.actor_param_count(), self.world_size, self._comm_stream)
                else:
                    scale = dist_util.allreduce_grad_(g, self.world_size)   # RCCL over xGMI: the one exchange step
                if per_pass:
                    scale /= nmb                                     # mean over the pass's minibatches (and ranks)
                if self.config.use_lr_decay:
                    lr = cosine_decay_lr(self.config, self.opt_step)
                    if self.config.adam_weight_decay == 0.0:      # train.py:1074-1075 as written: no sign flip behind scale_by_adam
                        if not getattr(self, "_warned_ascent", False):
                            import warnings
                            warnings.warn("use_lr_decay with adam_weight_decay == 0 reproduces train.py:1074-1075 as written: "
                                          "optax.chain(scale_by_adam, scale_by_schedule) has no sign flip, the update ASCENDS the loss")
                            self._warned_ascent = True
                        lr = -lr
                    self.ctx.set_learning_rate(lr)
                self.opt_step += 1
                self.ctx.adamw_step(self.params, self.opt_m, self.opt_v, g, self.opt_step, scale)

    def train_iteration(self):
        self.rollout()
        self.update()
        self.iteration += 1
        self.ctx.synchronize()   # one sync per iteration: surfaces device-side errors (e.g. a recurrence hand-off timeout) as KbjError

    def env_steps_per_iteration(self) -> int:
        return self.N * self.T

    # ---- read-only views of the compiled task wiring, under the reference's method names (train.py:1059-1276) ----
    def get_optimizer(self) -> wiring.OptimizerSpec:
        """train.py:1059-1077: adam / adamw, optionally under a cosine decay schedule; executed by kbj_adamw_step."""
        return wiring.optimizer(self.config, self.kcfg)

    def get_mujoco_model_metadata(self, mj_model: Optional[L.Model] = None) -> dict:
        """train.py:1083-1089: per-joint kp / kd / soft torque limit of metadata.json (compiled into the model blob)."""
        m = mj_model or self.model_blob
        return {n: dict(kp=m.kp[i], kd=m.kd[i], soft_torque_limit=m.tau_limit[i]) for i, n in enumerate(constants.JOINT_NAMES)}

    def get_actuators(self) -> wiring.PositionActuatorsSpec:
        return wiring.actuators(self.model_blob, self.kcfg)

    def get_physics_randomizers(self) -> dict:
        return wiring.physics_randomizers(self.kcfg)

    def get_events(self) -> dict:
        return wiring.events(self.kcfg)

    def get_resets(self) -> list:
        return wiring.resets(self.kcfg)

    def get_observations(self) -> dict:
        return wiring.observations(self.kcfg)

    def get_commands(self) -> dict:
        return wiring.commands(self.model_blob, self.kcfg)

    def get_rewards(self) -> dict:
        return wiring.rewards(self.kcfg)

    def get_terminations(self) -> dict:
        return wiring.terminations(self.kcfg)

    def get_curriculum(self) -> wiring.CurriculumSpec:
        return wiring.CurriculumSpec()

    def get_ppo_variables(self, trajectory: Optional[TrajBuffers] = None, params: Optional[torch.Tensor] = None) -> dict:
        """train.py:1510-1524. Without arguments: the variables of the LAST rollout as its own policy steps produced them (ksim recomputes
        them with the pre-update model in a separate on-policy pass: same parameters, same observations). With `trajectory` (any
        TrajBuffers of this task's shape - e.g. a reloaded one, or the last rollout after the parameters changed) and optionally
        `params`: that separate pass itself (kbj_ppo_forward: both nets through the T steps from the trajectory's start carries, carry
        reset where done, no gradients), minibatch by minibatch. Returns PPOVariables' per-step fields as [T][N](x20) tensors."""
        if trajectory is None:
            return dict(log_probs=self.traj.logp, values=self.traj.value, action=self.traj.action)
        p = self.params if params is None else params
        T, N, B, dev = self.T, self.N, self.B, self.device
        out = dict(log_probs=torch.empty(T, N, device=dev), values=torch.empty(T, N, device=dev), entropy=torch.empty(T, N, device=dev),
                   action_std=torch.empty(T, N, L.NU, device=dev), action_mean=torch.empty(T, N, L.NU, device=dev))
        lp, v, en = (torch.empty(T, B, device=dev) for _ in range(3))
        sd, mu = torch.empty(T, B, L.NU, device=dev), torch.empty(T, B, L.NU, device=dev)
        for mb in range(N // B):
            idx = torch.arange(mb * B, (mb + 1) * B, device=dev, dtype=torch.int32)
            self.ctx.ppo_forward(p, trajectory.c, idx, B, lp, v, en, sd, mu)
            sl = slice(mb * B, (mb + 1) * B)
            out["log_probs"][:, sl], out["values"][:, sl], out["entropy"][:, sl] = lp, v, en
            out["action_std"][:, sl], out["action_mean"][:, sl] = sd, mu
        out["action"] = trajectory.action
        return out

    def run_actor(self, actor_obs: torch.Tensor, critic_obs: torch.Tensor, step_index: int = 0):
        """train.py:1351-1379 + Actor.forward (:913-941) for all envs: returns the distribution's mode (filtered mean incl. biases).
        Both nets advance their carries (kbj_policy_step is the fused actor + critic step)."""
        a, _, _ = self.sample_action(actor_obs, critic_obs, step_index, argmax=True)
        return a

    def run_critic(self, actor_obs: torch.Tensor, critic_obs: torch.Tensor, step_index: int = 0):
        """train.py:1381-1433 + Critic.forward (:993-1004): the value estimate (carries advance as in run_actor)."""
        _, _, v = self.sample_action(actor_obs, critic_obs, step_index, argmax=True)
        return v

    # ---- checkpointing in the xax `ckpt.bin` layout (host/ckpt.py; convert.sh:4, train.py:1788) ----
    def save_checkpoint(self, path: str, background: bool = False):
        """Everything a bit-exact resume needs: parameters, optimizer, counters (upstream members) + env rows, reward carries, model
        carries and the pending observation rows (kbj_* members). The device arrays are copied to the host here, synchronously (a
        consistent snapshot: ~150 MB at 8192 envs); with `background` the container (npy blobs, gzip, fsync, rename) is written by a
        thread while training goes on - `wait_for_checkpoint()` joins it, and a new save waits for the previous one first. launch() saves
        this way: written in line a save costs seconds of a loop that runs an iteration in a third of one."""
        self.wait_for_checkpoint()
        T = self.T
        ep, es = self.ctx.env_get_state()
        extras = dict(ep=ep, es=es, rc=self.ctx.env_get_reward_carry(), actor_hc=self.carry.actor_hc.cpu().numpy(), critic_hc=self.carry.critic_hc.cpu().numpy(),
                      lpf=self.carry.lpf.cpu().numpy(), actor_obs_T=self.traj.actor_obs[T].cpu().numpy(), critic_obs_T=self.traj.critic_obs[T].cpu().numpy(),
                      aux_T=self.traj.aux[T].cpu().numpy())
        if self.mirror:
            extras.update(actor_mirror_hc=self.carry.actor_mirror_hc.cpu().numpy(), critic_mirror_hc=self.carry.critic_mirror_hc.cpu().numpy(),
                          lpf_mirror=self.carry.lpf_mirror.cpu().numpy())
        # carries of Python StatefulReward terms (extra_rewards): tensors (or tuples / lists of tensors) per term name
        import numpy as np
        for name, carry in self._extra_carries.items():
            leaves = list(carry) if isinstance(carry, (tuple, list)) else [carry]
            for i, leaf in enumerate(leaves):
                extras[f"extra_{name}_{i}"] = leaf.detach().cpu().numpy() if hasattr(leaf, "detach") else np.asarray(leaf)
            extras[f"extra_{name}_n"] = np.asarray(len(leaves) if isinstance(carry, (tuple, list)) else 0, np.int32)   # 0: a bare tensor
        cfg = dataclasses.asdict(self.config)
        cfg["action_latency_range"] = list(cfg["action_latency_range"])
        if cfg.get("fixed_command") is not None:
            cfg["fixed_command"] = list(cfg["fixed_command"])
        state = dict(num_steps=self.iteration, opt_step=self.opt_step, num_samples=self.iteration * self.N * self.T * self.world_size,
                     rank=self.rank, world_size=self.world_size)
        args = (path, self.params.cpu().numpy(), self.opt_m.cpu().numpy(), self.opt_v.cpu().numpy(), self.opt_step, self.H, self.kcfg.depth, state, cfg, extras)
        kw = dict(schedule_count=self.opt_step if self.config.use_lr_decay else None, extra_obs=self.extra_obs)
        if not background:
            ckpt_io.save_ckpt(*args, **kw)
            return
        import threading

        def write():
            try:
                ckpt_io.save_ckpt(*args, **kw)
            except BaseException as e:      # surfaced by wait_for_checkpoint(): a failed save must not pass silently
                self._save_error = e
        self._save_error = None
        self._save_thread = threading.Thread(target=write, name="kbj-ckpt-writer", daemon=False)
        self._save_thread.start()

    def wait_for_checkpoint(self):
        """Join a background save_checkpoint(); re-raises what the writer raised."""
        th = getattr(self, "_save_thread", None)
        if th is not None:
            th.join()
            self._save_thread = None
            err, self._save_error = getattr(self, "_save_error", None), None
            if err is not None:
                raise err

    def load_checkpoint(self, path: str):
        """Resume from save_checkpoint(): the next train_iteration() is bit-identical to the one the saved run would have made."""
        z = ckpt_io.load_ckpt(path, "all", hidden_size=self.H, depth=self.kcfg.depth)
        dev = self.device
        self.params.copy_(torch.from_numpy(z["model"]))
        opt, st = z["opt_state"], z["state"]
        if opt is not None:
            self.opt_m.copy_(torch.from_numpy(opt["mu"])); self.opt_v.copy_(torch.from_numpy(opt["nu"]))
            # `opt_step` is this build's key; an upstream checkpoint only has the optax `count` leaf (and xax's num_steps)
            self.opt_step = int(st.get("opt_step", opt["count"]))
        else:
            # no optimizer member, or one this build cannot map onto (mu, nu): the moments restart from zero, and so must the step count -
            # Adam's bias correction at the OLD count on empty moments would be ~1 instead of 1 / (1 - beta^t) and shrink the first steps
            self.opt_m.zero_(); self.opt_v.zero_()
            self.opt_step = 0
            if ckpt_io.has_member(path, "opt_state_0"):
                import warnings
                warnings.warn(f"{path}: opt_state_0 is present but does not map onto (count, mu, nu) of this model - "
                              "resuming with FRESH Adam moments and optimizer step 0 (parameters were loaded)")
        self.iteration = int(st.get("num_steps", 0))
        self._obs_started = self.iteration > 0          # a resumed run's row 0 already carries the user observation columns
        self._command_started = self.iteration > 0      # a resumed run's row 0 already carries the user command term's commands
        self._resets_started = self.iteration > 0       # ... and its env rows the user Reset terms' state
        x = z["extras"]
        if "es" not in x:
            return            # a model-only checkpoint (e.g. written by the reference): parameters and optimizer only
        if x["es"].shape[0] != self.N:
            raise B.KbjError(f"checkpoint holds {x['es'].shape[0]} envs, this task has {self.N}")
        self.ctx.env_set_state(x["ep"], x["es"])
        self.ctx.env_set_reward_carry(x["rc"])
        T = self.T
        self.carry.actor_hc.copy_(torch.from_numpy(x["actor_hc"])); self.carry.critic_hc.copy_(torch.from_numpy(x["critic_hc"]))
        self.carry.lpf.copy_(torch.from_numpy(x["lpf"]))
        self.traj.actor_obs[T].copy_(torch.from_numpy(x["actor_obs_T"])); self.traj.critic_obs[T].copy_(torch.from_numpy(x["critic_obs_T"]))
        self.traj.aux[T].copy_(torch.from_numpy(x["aux_T"]))
        if self.mirror:
            self.carry.actor_mirror_hc.copy_(torch.from_numpy(x["actor_mirror_hc"]))
            self.carry.critic_mirror_hc.copy_(torch.from_numpy(x["critic_mirror_hc"]))
            self.carry.lpf_mirror.copy_(torch.from_numpy(x["lpf_mirror"]))
        self._extra_carries = {}
        for name in self.extra_rewards:      # Python StatefulReward carries; a term the checkpoint does not know starts from initial_carry
            if f"extra_{name}_n" in x:
                n = int(x[f"extra_{name}_n"])
                leaves = [torch.from_numpy(np.array(x[f"extra_{name}_{i}"])).to(dev) for i in range(max(n, 1))]
                self._extra_carries[name] = tuple(leaves) if n > 0 else leaves[0]

    @classmethod
    def load_task(cls, ckpt_path: str, device: Optional[torch.device] = None) -> "HumanoidWalkingTask":
        """convert.py:36 `HumanoidWalkingTask.load_task(ckpt_path)`: rebuild the task from the checkpoint's config member and
        load its state."""
        cfg = ckpt_io.load_ckpt(ckpt_path, "config")
        fields = {f.name for f in dataclasses.fields(HumanoidWalkingTaskConfig)}
        kw = {k: v for k, v in cfg.items() if k in fields}
        for k in ("action_latency_range", "fixed_command"):
            if kw.get(k) is not None:
                kw[k] = tuple(kw[k])
        task = cls(HumanoidWalkingTaskConfig(**kw), device=device)
        task.load_checkpoint(ckpt_path)
        return task

    def load_ckpt(self, path: str, init_params=None, part: str = "model"):
        """convert.py:39 `task.load_ckpt(ckpt_path, init_params=..., part="model")[0]`: returns a 1-tuple-like list whose first element is
        the requested part; for "model" a ModelView with `.actor` (named leaves) as convert.py:44-46 reads it."""
        if part == "model":
            flat = ckpt_io.load_ckpt(path, "model", hidden_size=self.H, depth=self.kcfg.depth)
            return [ModelView(flat, self.H, self.kcfg.depth, self.extra_obs)]
        return [ckpt_io.load_ckpt(path, part, hidden_size=self.H, depth=self.kcfg.depth)]

    # ---- validation (train.py:1564 argmax=True; valid_every_n_steps train.py:1789) ----
    def validate(self, num_envs: int = 64, seconds: Optional[float] = None, seed_offset: int = 7919, _capture=None, _stepwise: bool = False) -> dict:
        """Deterministic validation rollout: a separate small env set (its own context, carries and buffers, so training state is
        untouched), actions = the distribution's mode, `render_length_seconds` long. Returns scalar statistics.
        User Reset / Command / Observation terms are applied as in training; user TERMINATION terms (`extra_terminations`) are NOT - validation
        episodes end by the built-in terminations (bad z, tilt, episode length) only, on the fused and on the step-wise path alike.
        `_capture(ctx, frame)` (view()): called after the reset (frame 0) and after every control step."""
        seconds = self.config.render_length_seconds if seconds is None else seconds
        T = max(1, int(round(seconds / self.config.ctrl_dt)))
        key = (num_envs, T)
        if getattr(self, "_valid", None) is None or self._valid[0] != key:
            vcfg = self.config.to_kbj(num_envs) if num_envs % self.config.batch_size == 0 else dataclasses.replace(self.config, batch_size=num_envs).to_kbj(num_envs)
            vcfg.rollout_len = T
            if self.command_term is not None:
                vcfg.command_mode = 1
            vctx = B.Context(self.model_blob, vcfg, self.device.index or 0, torch.cuda.current_stream().cuda_stream)
            self._valid = (key, vctx, CarryBuffers(num_envs, self.H, self.kcfg.depth, self.device, mirror=self.mirror),
                           TrajBuffers(T, num_envs, self.H, self.kcfg.depth, self.device, mirror=self.mirror, reward_comps=True,
                                       ld_actor=self.ld_actor, ld_critic=self.ld_critic))
        _, vctx, carry, tr = self._valid
        seed = self.config.seed + seed_offset
        carry.zero_()
        vctx.env_reset_all(seed, tr.actor_obs[0], tr.critic_obs[0], tr.aux[0])
        if self.extra_resets:                   # the user's Reset terms shape the validation episodes as they shape the training ones
            self._apply_resets(vctx, tr, 0, None, seed_offset)
        if self.command_term is not None:       # the user's Command term drives the validation envs as it drives the training ones
            from .traj_view import StepView
            self._apply_command(vctx, tr, 0, StepView(tr.aux[0], tr.actor_obs[0], tr.critic_obs[0], tr.aux[0], self.model_blob),
                                torch.ones(num_envs, dtype=torch.bool, device=self.device), seed_offset, all_fresh=True)
        feeds = any(self.extra_obs)            # user observation terms that are network inputs drive the validation policy too
        if feeds:
            from .traj_view import StepView
            self._observe_into_rows(tr, 0, StepView(tr.aux[0], tr.actor_obs[0], tr.critic_obs[0], tr.aux[0], self.model_blob))
        if _capture is not None:
            _capture(vctx, 0)
        fused = _capture is None and not feeds and self.command_term is None and not self.extra_resets and not _stepwise
        if fused:
            # no user term and nothing to capture between the steps: ONE kbj_rollout call in argmax mode (the same kernels, bit-identical to the
            # step-by-step loop below, ~3 x fewer host calls: 0.05 instead of 0.15 s for 64 envs x 500 steps). kbj_rollout takes its first
            # observation from row T (the previous rollout's last), so the reset rows move there first.
            tr.actor_obs[T].copy_(tr.actor_obs[0]); tr.critic_obs[T].copy_(tr.critic_obs[0]); tr.aux[T].copy_(tr.aux[0])
            vctx.set_rollout_argmax(True)
            try:
                vctx.rollout(self.params, carry.c, seed, 0, tr.c)
            finally:
                vctx.set_rollout_argmax(False)      # the cached context samples again whatever happened (the flag is context state, not a call argument)
        for t in range(0 if not fused else T, T):
            vctx.policy_step(self.params, tr.actor_obs[t], tr.critic_obs[t], carry.c, seed, t, True, tr.action[t], tr.logp[t], tr.value[t])
            vctx.env_step(tr.action[t], tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1])
            if self.extra_resets:
                self._apply_resets(vctx, tr, t + 1, tr.aux[t][:, L.AUX["DONE"]] != 0, seed_offset + t + 1)
            if feeds:
                self._observe_into_rows(tr, t + 1, StepView(tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1], self.model_blob))
            if self.command_term is not None:
                self._apply_command(vctx, tr, t + 1, StepView(tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1], self.model_blob),
                                    tr.aux[t][:, L.AUX["DONE"]] != 0, seed_offset + t + 1)
            vctx.carry_reset(carry.c, tr.aux[t].data_ptr() + 4 * L.AUX["DONE"], L.AUX["SIZE"])
            if _capture is not None:
                _capture(vctx, t + 1)
        if not fused:
            vctx.rewards(tr.aux, T, tr.reward, tr.comps)
        vctx.synchronize()
        done = tr.done
        fails, succ = float((done < 0).sum()), float((done > 0).sum())
        out = {"valid/reward_per_step": float(tr.reward.mean()), "valid/failures_per_step": fails / (T * num_envs),
               "valid/episode_length_s": float(T * num_envs / max(1.0, fails + succ + num_envs) * self.config.ctrl_dt), "valid/value_mean": float(tr.value.mean())}
        for name, v in zip(constants.REWARD_NAMES, tr.comps.mean(dim=(0, 1)).cpu().tolist()):
            out[f"valid/reward/{name}"] = v
        return out

    def close_validation(self):
        """Drop the cached validation / view context (its env rows, workspace and lanes). validate() builds it again on demand. Measured on
        MI355X (tools/validate_cliff.py, 8192 envs): keeping it costs the training iterations nothing (365.2 ms before the first validation,
        366.1 / 365.3 / 364.8 ms after a validation, a second one and a view), re-creating it costs 0.15 s per validation - so it stays cached."""
        v = getattr(self, "_valid", None)
        if v is not None:
            v[1].close()
            self._valid = None

    def close(self):
        """Release the library contexts of this task (training and validation)."""
        self.wait_for_checkpoint()
        self.close_validation()
        self.ctx.close()

    def view(self, path: Optional[str] = None, num_envs: int = 4, seconds: Optional[float] = None, seed_offset: int = 7919):
        """`run_mode=view` (reference README.md:66-70; train.py:1783-1784 render_track_body_id / render_length_seconds): the validation
        rollout above on `num_envs` envs with the generalised positions recorded after every control step (kbj_env_get_state), returned
        as a `host.view.Recording`; with `path` (a stem) also written as `<path>.npz` and a self-contained `<path>.html` player."""
        from .view import Recording
        frames, eps = [], []

        def capture(vctx, frame):
            ep, es = vctx.env_get_state()
            frames.append(es[:, L.ES["QPOS"]:L.ES["QPOS"] + int(self.model_blob.nq)].copy())
            if frame == 0:
                eps.append(ep.copy())
        stats = self.validate(num_envs, seconds, seed_offset, _capture=capture)
        tr = self._valid[3]
        T = len(frames) - 1
        rec = Recording(self.model_blob, np.stack(frames), eps[0], tr.aux[:T + 1, :, L.AUX["CMD"]:L.AUX["CMD"] + 16].cpu().numpy(), tr.reward[:T].cpu().numpy(),
                        tr.done[:T].cpu().numpy(), self.config.ctrl_dt, self.config.render_track_body_id,
                        (self.kcfg.terrain_amp, self.kcfg.terrain_wavelength))
        rec.stats = stats
        if path is not None:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            rec.save_npz(path + ".npz")
            rec.save_html(path + ".html", title=f"kbot-joystick policy, iteration {self.iteration}: reward/step {stats['valid/reward_per_step']:.3f}")
        return rec

    def scalars(self) -> dict:
        """The scalars the reference's logger plots after each iteration (train.py:1783-1790): loss terms, reward, terminations."""
        m = self.metrics.cpu().tolist()
        names = ("loss", "policy_loss", "value_loss", "entropy", "clip_fraction", "approx_kl", "adv_mean", "adv_std", "action_mirror_loss", "value_mirror_loss")
        out = {f"train/{n}": v for n, v in zip(names, m)}
        done = self.traj.done
        st = torch.stack([self.traj.reward.mean(), (done < 0).float().mean(), (done > 0).float().mean(), self.traj.value.mean(), self.traj.logp.mean()]).cpu().tolist()   # one sync
        out["train/reward_per_step"], out["train/failures_per_step"], out["train/truncations_per_step"], out["train/value_mean"], out["train/action_std_logp"] = st
        if self.traj.comps is not None:
            for name, v in self.reward_components().items():
                out[f"reward/{name}"] = v
        return out

    def export_actor(self, path: str):
        """convert.py's input: the actor's leaves, joint/command order and the flat carry size (host/export.py)."""
        from . import export
        c = self.kcfg
        export.export_actor(path, self.params.cpu().numpy(), self.H, c.depth, c.ctrl_dt, self.config.cutoff_frequency, c.min_std, c.max_std,
                            c.var_scale, list(self.model_blob.joint_bias), extra_obs=self.extra_obs)

    def reward_components(self):
        """Mean of every unscaled reward term over the last rollout (train.py:1224-1256 order, spec/constants.REWARD_NAMES);
        needs `log_reward_components=True` in the config (kbj_rollout then also writes the [T][N][12] terms)."""
        if self.traj.comps is None:
            raise B.KbjError("reward_components() needs HumanoidWalkingTaskConfig.log_reward_components=True")
        return dict(zip(constants.REWARD_NAMES, self.traj.comps.mean(dim=(0, 1)).cpu().tolist()))

    @classmethod
    def launch(cls, config: HumanoidWalkingTaskConfig, num_iterations: int = 10, log_every: int = 1, run_dir: Optional[str] = None, quiet: bool = False):
        """train.py:1760: build the task and run the training loop (single process; bench.py / torchrun drive N GPUs).
        With `run_dir` (the reference's `humanoid_walking_task/run_N`): scalars go to `run_dir/logs` (CSV + TensorBoard event file),
        `run_dir/checkpoints/ckpt.bin` is rewritten every `save_every_n_seconds` (train.py:1788, convert.sh:4) and at the end, and a
        deterministic validation rollout runs every `valid_every_n_steps` iterations (train.py:1789)."""
        from .scalars import ScalarLogger
        task = cls(config)
        if getattr(config, "run_mode", "train") == "view":      # README.md:66-70 `python -m train run_mode=view`: no training, play the checkpointed policy
            # the mode exists to look at the TRAINED policy: without the run directory's checkpoint it would record the freshly initialised
            # one under a title that says "policy, iteration 0" - a plausible-looking, meaningless rollout. The reference's view mode loads the
            # run's checkpoint as well (README.md:66-70).
            ck = os.path.join(run_dir, "checkpoints", "ckpt.bin") if run_dir is not None else None
            if ck is None or not os.path.exists(ck):
                task.close()
                raise FileNotFoundError(f"run_mode=view plays the checkpointed policy: {ck or '<run_dir>/checkpoints/ckpt.bin'} does not exist "
                                        "(pass run_dir= of a finished or running training run; task.view() records the CURRENT parameters of a live task)")
            task.load_checkpoint(ck)
            out = os.path.join(run_dir if run_dir is not None else ".", "view", f"rollout_{task.iteration}")
            rec = task.view(out)
            if not quiet:
                print(f"run_mode=view: {out}.html / .npz ({rec.qpos.shape[0]} frames x {rec.qpos.shape[1]} envs, reward/step {rec.stats['valid/reward_per_step']:.4f})")
            return task
        logger, ckpt_path = None, None
        if run_dir is not None:
            logger = ScalarLogger(os.path.join(run_dir, "logs"))
            os.makedirs(os.path.join(run_dir, "checkpoints"), exist_ok=True)
            ckpt_path = os.path.join(run_dir, "checkpoints", "ckpt.bin")
        t0 = last_save = last_valid = time.time()
        stats = dict(iterations=0, validations=0, validation_seconds=0.0, checkpoints=0, checkpoint_seconds=0.0)
        for it in range(num_iterations):
            task.train_iteration()
            now = time.time()
            if it == 0:
                stats["first_iteration_seconds"] = now - t0      # one-time costs live here: code objects loaded at first launch, pinned staging buffers
            if (it + 1) % log_every == 0:
                sc = task.scalars()
                sc["perf/env_steps_per_s"] = task.env_steps_per_iteration() * (it + 1) / (now - t0)
                due = (config.valid_every_n_steps and (it + 1) % config.valid_every_n_steps == 0) or \
                      (config.valid_every_n_seconds and now - last_valid >= config.valid_every_n_seconds)
                if due:
                    tv = time.time()
                    sc.update(task.validate())
                    last_valid = now
                    stats["validations"] += 1; stats["validation_seconds"] += time.time() - tv
                if logger:
                    logger.log(task.iteration, sc)
                if not quiet:
                    print(f"iter {it + 1}: reward/step {sc['train/reward_per_step']:.4f} loss {sc['train/loss']:.4f} value_loss {sc['train/value_loss']:.4f} "
                          f"entropy {sc['train/entropy']:.3f} clipfrac {sc['train/clip_fraction']:.3f} | {sc['perf/env_steps_per_s']:.3e} env-steps/s")
            if ckpt_path and config.save_every_n_seconds and now - last_save >= config.save_every_n_seconds:
                tc = time.time()
                task.save_checkpoint(ckpt_path, background=True)     # snapshot now, container written under the next iterations
                last_save = now
                stats["checkpoints"] += 1; stats["checkpoint_seconds"] += time.time() - tc
            stats["iterations"] = it + 1
        # the loop as its user sees it: training iterations + scalar logging + validations + periodic checkpoints, up to here
        stats["loop_seconds"] = time.time() - t0
        stats["env_steps_per_s"] = task.env_steps_per_iteration() * stats["iterations"] / max(stats["loop_seconds"], 1e-9)
        # steady state: everything behind the first iteration (its logging included), as a benchmark's warm-up step is outside its timed region
        if stats["iterations"] > 1:
            stats["steady_env_steps_per_s"] = task.env_steps_per_iteration() * (stats["iterations"] - 1) / max(stats["loop_seconds"] - stats["first_iteration_seconds"], 1e-9)
        if ckpt_path:
            tc = time.time()
            task.save_checkpoint(ckpt_path)      # the final one in line: the file is complete when launch() returns
            stats["final_checkpoint_seconds"] = time.time() - tc
        if logger:
            logger.close()
        task.loop_stats = stats
        return task


class ModelView:
    """What convert.py:39-46 takes out of a checkpoint: `model.actor` with the leaves Actor.forward uses (train.py:847-941), as
    numpy arrays under their equinox names, plus the carry layout convert.py:71-78 flattens."""

    class _Net:
        def __init__(self, leaves: dict, prefix: str, depth: int):
            g = lambda k: leaves[f"{prefix}.{k}"]
            self.input_proj = _Leaf(weight=g("input_proj.weight"), bias=g("input_proj.bias"))
            self.rnns = tuple(_Leaf(weight_ih=g(f"rnns.{l}.weight_ih"), weight_hh=g(f"rnns.{l}.weight_hh"), bias=g(f"rnns.{l}.bias")) for l in range(depth))
            self.output_proj = _Leaf(weight=g("output_proj.weight"), bias=g("output_proj.bias"))

    def __init__(self, flat, hidden_size: int, depth: int = 2, extra_obs=(0, 0)):
        self.hidden_size, self.depth = hidden_size, depth
        leaves = dict(ckpt_io.split_leaves(flat, hidden_size, depth, extra_obs))
        self.leaves = leaves
        self.actor = ModelView._Net(leaves, "actor", depth)
        self.critic = ModelView._Net(leaves, "critic", depth)
        self.carry_size = depth * 2 * hidden_size + L.NU     # convert.py:71: flat (depth, 2, H) LSTM carry + 20 low-pass floats


class _Leaf:
    def __init__(self, **kw):
        self.__dict__.update(kw)
