"""Python-side reward terms on the stored trajectory (the reference's "edit train.py" workflow).

The twelve built-in terms are kernels (`rewards_kernel`, scales and constructor arguments in `kbj_config`). A user who wants a NEW term
writes it the way the reference does - an object with a `scale` and `get_reward(trajectory) -> [T, N]` (ksim's Reward protocol,
train.py:161-165), or `initial_carry()` / `get_reward_stateful(trajectory, carry)` for stateful ones (train.py:135-154) - against
`TrajectoryView`, which exposes the per-step record the built-in stack itself reads (`kbj_model.h` KBJ_AUX_*: state AFTER the step, the
observation-derived touch / com-distance the policy saw, the command, done flags) plus actions and the packed observations, as
torch tensors on the GPU. `HumanoidWalkingTask(config, extra_rewards={...})` adds `scale * term` to the rollout's reward before GAE.
Nothing here touches the hot path's kernels: it is a plain torch epilogue after `kbj_rollout`.
"""
from __future__ import annotations

from typing import Dict

import torch

from ..spec import layout as L


class TrajectoryView:
    """Named views (no copies) of one rollout: every field is [T, N, ...] on the device."""

    def __init__(self, traj, T: int):
        aux = traj.aux[:T]
        A = L.AUX
        self.T, self.N = T, aux.shape[1]
        self.base_qvel = aux[:, :, A["QVEL"]:A["QVEL"] + 6]            # qvel[0:6] after the step (world linear, local angular)
        self.base_quat = aux[:, :, A["BQUAT"]:A["BQUAT"] + 4]          # xquat[base], w first
        self.base_z = aux[:, :, A["BASEZ"]]
        self.left_foot_z = aux[:, :, A["LFZ"]]
        self.right_foot_z = aux[:, :, A["RFZ"]]
        self.left_foot_quat = aux[:, :, A["LFQUAT"]:A["LFQUAT"] + 4]
        self.right_foot_quat = aux[:, :, A["RFQUAT"]:A["RFQUAT"] + 4]
        self.arm_qpos = aux[:, :, A["ARMQ"]:A["ARMQ"] + 10]
        self.ctrl = aux[:, :, A["CTRL"]:A["CTRL"] + L.NU]             # actuator torques of the last substep
        self.foot_touch = aux[:, :, A["TOUCH"]:A["TOUCH"] + 2]         # left, right (observation the policy saw)
        self.com_distance = aux[:, :, A["COMDIST"]]
        self.command = aux[:, :, A["CMD"]:A["CMD"] + L.NCMD]           # train.py:744-763 layout (vx, vy, wz, height, roll, pitch, 10 arm targets)
        self.done = aux[:, :, A["DONE"]]                               # -1 failure, 0 running, +1 episode-length truncation
        self.action = traj.action[:T]
        self.actor_obs = traj.actor_obs[:T, :, :L.NOBS_ACTOR]
        self.critic_obs = traj.critic_obs[:T, :, :L.NOBS_CRITIC]


def apply_extra_rewards(terms: Dict[str, object], carries: Dict[str, object], view: TrajectoryView, reward: torch.Tensor) -> Dict[str, float]:
    """reward [T, N] += scale * term for every user term; returns the unscaled means (for logging)."""
    means = {}
    for name, term in terms.items():
        scale = float(getattr(term, "scale", 1.0))
        if hasattr(term, "get_reward_stateful"):
            if name not in carries:
                carries[name] = term.initial_carry(view.N, reward.device) if hasattr(term, "initial_carry") else None
            r, carries[name] = term.get_reward_stateful(view, carries[name])
        else:
            r = term.get_reward(view)
        if r.shape != reward.shape:
            raise ValueError(f"reward term {name!r} returned shape {tuple(r.shape)}, expected {tuple(reward.shape)}")
        reward.add_(r.to(reward.dtype), alpha=scale)
        means[name] = float(r.mean())
    return means


class StepView:
    """What a user-written Termination / Observation term sees after one control step of all envs (the reference hands such terms the
    engine's `physics_data`, train.py:635, 682, 706, 817): the post-step record of the step (`KBJ_AUX_*`, the state BEFORE any reset) and
    the next observation rows (the state the policy will see: post-reset for the envs the kernel's own terminations finished). [N, ...]
    torch views on the device, no copies."""

    _joint_tables: dict = {}       # (device, joint tables) -> (bias, range) tensors: uploaded once, not twice per control step

    @classmethod
    def joint_tables(cls, model, device):
        key = (str(device), tuple(model.joint_bias), tuple(model.joint_lo), tuple(model.joint_hi))
        if key not in cls._joint_tables:
            bias = torch.tensor(list(model.joint_bias), device=device)
            rng = torch.tensor([max(b - lo, hi - b) for b, lo, hi in zip(model.joint_bias, model.joint_lo, model.joint_hi)], device=device)
            cls._joint_tables[key] = (bias, rng)
        return cls._joint_tables[key]

    def __init__(self, aux_t: torch.Tensor, actor_next: torch.Tensor, critic_next: torch.Tensor, aux_next: torch.Tensor, model, qstate_t: torch.Tensor = None):
        A, O = L.AUX, L.OBS
        self.N = aux_t.shape[0]
        self._model, self._qstate, self._kin = model, qstate_t, None
        # ---- post-step, pre-reset (what the kernel's own terminations and the reward stack read) ----
        self.base_qvel = aux_t[:, A["QVEL"]:A["QVEL"] + 6]
        self.base_quat = aux_t[:, A["BQUAT"]:A["BQUAT"] + 4]
        self.base_z = aux_t[:, A["BASEZ"]]                      # xpos[base].z
        self.left_foot_z = aux_t[:, A["LFZ"]]                   # xpos[left foot].z
        self.right_foot_z = aux_t[:, A["RFZ"]]
        self.left_foot_quat = aux_t[:, A["LFQUAT"]:A["LFQUAT"] + 4]
        self.right_foot_quat = aux_t[:, A["RFQUAT"]:A["RFQUAT"] + 4]
        self.arm_qpos = aux_t[:, A["ARMQ"]:A["ARMQ"] + 10]
        self.ctrl = aux_t[:, A["CTRL"]:A["CTRL"] + L.NU]
        self.command = aux_t[:, A["CMD"]:A["CMD"] + L.NCMD]
        self.done = aux_t[:, A["DONE"]]                         # the kernel's own terminations: -1 failure, +1 episode length
        # ---- the next observation (clean critic pieces, train.py:1381-1433 order; KBJ_OBS_* offsets) ----
        piece = lambda name: critic_next[:, O[name][0]:O[name][0] + O[name][1]]
        bias, rng = self.joint_tables(model, aux_t.device)
        self.joint_position = piece("JPOS") * rng + bias         # qpos[7:]
        self.joint_velocity = piece("JVEL") * L.OBS_JVEL_DIV     # qvel[6:]
        self.projected_gravity = critic_next[:, O["PG"][0] + 2:O["PG"][0] + 5]
        self.imu_gyro = piece("GYRO")
        self.foot_touch = piece("TOUCH")
        self.feet_position = piece("FEETPOS")
        self.base_position = piece("BASEPOS")                    # qpos[0:3]
        self.base_orientation = piece("BASEQUAT")                # qpos[3:7]
        self.center_of_mass_inertia = piece("CINERT").reshape(self.N, -1, 10)
        self.center_of_mass_velocity = piece("CVEL").reshape(self.N, -1, 6)
        self.base_linear_velocity = piece("LINVEL")
        self.base_angular_velocity = piece("ANGVEL")
        self.actuator_force = piece("ACTFRC") * L.OBS_ACTFRC_DIV
        self.base_height = piece("HEIGHT")[:, 0]
        self.com_distance = aux_next[:, A["COMDIST"]]
        self.actor_obs, self.critic_obs = actor_next[:, :L.NOBS_ACTOR], critic_next[:, :L.NOBS_CRITIC]


    # ---- ksim's physics_data names (train.py:817-823 reads `state.xpos[body, 2]`): available when the task records the step state
    # (HumanoidWalkingTaskConfig.record_state -> kbj_env_record_state); [N, ...] here, [...] per env under per_env_state() ----
    def _need_state(self):
        if self._qstate is None:
            raise ValueError("state.qpos / .qvel / .xpos / .xquat need the per-step state record: HumanoidWalkingTaskConfig(record_state=True)")
        return self._qstate

    @property
    def qpos(self) -> torch.Tensor:
        q = self._need_state()
        return q[:, L.QSTATE["QPOS"]:L.QSTATE["QPOS"] + L.NQ]

    @property
    def qvel(self) -> torch.Tensor:
        q = self._need_state()
        return q[:, L.QSTATE["QVEL"]:L.QSTATE["QVEL"] + L.NV]

    def _kinematics(self):
        if self._kin is None:
            from .trajectory import forward_kinematics
            q = self._need_state()
            xp, xq = forward_kinematics(self._model, q[:, L.QSTATE["QPOS_KIN"]:L.QSTATE["QPOS_KIN"] + L.NQ])
            self._kin = (xp.to(torch.float32), xq.to(torch.float32))
        return self._kin

    @property
    def xpos(self) -> torch.Tensor:
        """[N, nbody, 3]: body positions of the step's last forward pass (what the kernel's own terminations read)."""
        return self._kinematics()[0]

    @property
    def xquat(self) -> torch.Tensor:
        return self._kinematics()[1]


def per_env_state(fn):
    """A Termination body written per env, exactly as the reference's (`state.xpos[self.base_idx, 2]`, train.py:817-823), over all envs of a StepView
    under torch.vmap: `per_env_state(body)(state, curriculum_level) -> [N]`."""
    import types

    def run(state: "StepView", curriculum_level=1.0) -> torch.Tensor:
        fields = dict(qpos=state.qpos, qvel=state.qvel, xpos=state.xpos, xquat=state.xquat)
        return torch.vmap(lambda f: fn(types.SimpleNamespace(**f), curriculum_level), in_dims=0)(fields)
    return run


def combine_terminations(terms: Dict[str, object], view: StepView, curriculum_level: float = 1.0) -> torch.Tensor:
    """[N] float in {-1, 0, 1}: the first non-zero answer of the user terms, in dictionary order (train.py:1258-1269 lists them likewise)."""
    out = torch.zeros(view.N, device=view.done.device)
    for name, term in terms.items():
        v = term(view, curriculum_level) if callable(term) else term.__call__(view, curriculum_level)
        v = v.to(out.dtype)
        if v.shape != out.shape:
            raise ValueError(f"termination {name!r} returned shape {tuple(v.shape)}, expected {tuple(out.shape)}")
        out = torch.where(out != 0, out, torch.sign(v))
    return out


class ResetData:
    """What a user-written Reset term receives and returns (train.py:833-844: `data.qpos`, `ksim.update_data_field(data, "qpos", qpos_j)`): the
    generalised positions / velocities of ALL envs as [N, 27] / [N, 26] device tensors ([27] / [26] per env under `per_env_reset`). The task keeps
    the term's result only for the envs that have just been re-initialised."""

    def __init__(self, qpos: torch.Tensor, qvel: torch.Tensor):
        self.qpos, self.qvel = qpos, qvel


def update_data_field(data: ResetData, name: str, value: torch.Tensor) -> ResetData:
    """ksim.update_data_field for the two fields a Reset may change: returns a new object (the terms are functional, as in the reference)."""
    if name not in ("qpos", "qvel"):
        raise KeyError(f"a Reset term may update 'qpos' or 'qvel', not {name!r}")
    return ResetData(value if name == "qpos" else data.qpos, value if name == "qvel" else data.qvel)


def per_env_reset(fn):
    """A Reset body written per env as the reference's (`qpos_j.at[0:1].set(new_x)` on a [27] vector), over all envs under torch.vmap. `rng` is the
    torch.Generator the task passes; draw per-env random numbers OUTSIDE the vmapped body (torch.vmap has no per-example generators) and pass
    them in through a closure or extra tensors of shape [N, ...] given as `extras`."""
    def run(data: ResetData, curriculum_level=1.0, rng=None, extras=()):
        out = torch.vmap(lambda q, v, *x: (lambda d: (d.qpos, d.qvel))(fn(ResetData(q, v), curriculum_level, rng, *x)), in_dims=0)(data.qpos, data.qvel, *extras)
        return ResetData(out[0], out[1])
    return run
