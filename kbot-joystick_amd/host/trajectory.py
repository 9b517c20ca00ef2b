"""A ksim-shaped `Trajectory` for user reward terms and the reference's twelve reward classes restated on it.

The reference's reward classes read a `ksim.Trajectory` by attribute (`/root/reference/train.py:138-154, 161-165, 197-213, 261-265, 274-292,
301-306, 316-334, 377-388, 418-457, 466-478, 487-494, 503-506`): `trajectory.qpos`, `.qvel`, `.xpos[:, body]`, `.xquat[:, body]`, `.ctrl`,
`.done`, `.obs["left_foot_touch"]`, `.obs["com_distance"]`, `.command["unified_command"]`. `Trajectory` below carries exactly those names
for one rollout of ALL envs - every field is `[T, N, ...]` where the reference's (vmapped over envs) is `[T, ...]`, so a term written with
`[..., idx]` / `axis=-1` as the reference's are runs unchanged; `[:, idx]` on the time axis becomes `[..., idx, :]` on the body axis - or the body
runs unchanged through `per_env(fn)`, which hands it ksim's per-env `[T, ...]` view under `torch.vmap` (as ksim does under `jax.vmap`).

Where the fields come from (nothing here is on the hot path; all of it is torch on the device):
  * `qpos [T,N,27]`, `qvel [T,N,26]`: the per-step state record the env kernel writes when asked to (`kbj_traj.qstate_d`,
    `HumanoidWalkingTaskConfig.record_state=True`): the state AFTER step t, before any reset - what a ksim Trajectory step holds.
  * `xpos [T,N,24,3]`, `xquat [T,N,24,4]`: forward kinematics (the model blob's body tree, fp64) over the positions the step's LAST forward
    pass ran on (`KBJ_QSTATE_QPOS_KIN`): as in `mj_step`, the derived quantities a step leaves behind are one integration behind `qpos`, and
    these are the poses the built-in reward stack reads (`KBJ_AUX_BASEZ`, `..._LFQUAT`, ...).
  * `ctrl`, `done`, `command`, the touch / com-distance observations: views of the aux record (`KBJ_AUX_*`); the other observation entries:
    views of the packed actor / critic rows (`KBJ_OBS_*`), de-normalised where the packing normalises (train.py:1351-1433).

`reference_rewards(model, config)` returns the reference's `get_rewards()` dictionary (train.py:1224-1256: same keys, classes, constructor
arguments) as torch terms in ksim's Reward / StatefulReward protocol. They are what a user copies and edits - the "edit train.py" workflow -
and `tests/test_gpu_host.py::test_reference_reward_classes_on_the_trajectory_reproduce_the_kernel` holds them against `rewards_kernel`
term by term. The built-in stack stays the kernel; these run only where a user passes them as `extra_rewards`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from ..spec import constants, layout as L
from .traj_view import TrajectoryView


# ---- xax geometry helpers (same formulas as csrc/kbj_env_core.h and the oracle), batched over leading axes ----
def quat_to_euler(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = q.unbind(-1)
    roll = torch.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    pitch = torch.asin(torch.clamp(2 * (w * y - z * x), -1.0, 1.0))
    yaw = torch.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return torch.stack([roll, pitch, yaw], dim=-1)


def euler_to_quat(e: torch.Tensor) -> torch.Tensor:
    r, p, y = (e[..., k] * 0.5 for k in range(3))
    cr, sr, cp, sp, cy, sy = torch.cos(r), torch.sin(r), torch.cos(p), torch.sin(p), torch.cos(y), torch.sin(y)
    return torch.stack([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy], dim=-1)


def quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def rotate_vector_by_quat(v: torch.Tensor, q: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    if inverse:
        q = q * torch.tensor([1.0, -1.0, -1.0, -1.0], device=q.device, dtype=q.dtype)
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + w * t + torch.cross(u, t, dim=-1)


def get_norm(x: torch.Tensor, norm: str) -> torch.Tensor:
    """xax.get_norm: ELEMENTWISE (no reduction, no square root): "l1" = |x|, "l2" = x^2."""
    if norm == "l1":
        return x.abs()
    if norm == "l2":
        return x * x
    raise ValueError(norm)


def forward_kinematics(model, qpos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """World positions [..., nbody, 3] and orientations [..., nbody, 4] (w first) of every body for generalised positions [..., nq]: the
    torch restatement of host/view.forward_kinematics (checked against the oracle's xpos / xquat in tests/test_host_cpu.py), fp64 inside."""
    q = qpos.to(torch.float64)
    lead, dev = q.shape[:-1], q.device
    nb = int(model.nbody)
    xp = [torch.zeros(lead + (3,), dtype=torch.float64, device=dev)]
    xq = [torch.zeros(lead + (4,), dtype=torch.float64, device=dev)]
    xq[0][..., 0] = 1.0
    for b in range(1, nb):
        p = int(model.body_parent[b])
        num, adr = int(model.body_dofnum[b]), int(model.body_dofadr[b])
        if num == 6:      # free joint: qpos holds the world pose
            bq = q[..., 3:7]
            xp.append(q[..., 0:3].clone())
            xq.append(bq / bq.norm(dim=-1, keepdim=True))
            continue
        bp = torch.tensor(list(model.body_pos[b][:]), dtype=torch.float64, device=dev)
        bq = torch.tensor(list(model.body_quat[b][:]), dtype=torch.float64, device=dev)
        pos = xp[p] + rotate_vector_by_quat(bp.expand(lead + (3,)), xq[p])
        quat = quat_mul(xq[p], bq.expand(lead + (4,)))
        if num == 1:      # hinge about jnt_axis through the body origin; qpos index = dof index + 1
            ang = q[..., adr + 1]
            ax = torch.tensor(list(model.jnt_axis[b][:]), dtype=torch.float64, device=dev)
            jq = torch.cat([torch.cos(0.5 * ang)[..., None], torch.sin(0.5 * ang)[..., None] * ax], dim=-1)
            quat = quat_mul(quat, jq)
        xp.append(pos)
        xq.append(quat / quat.norm(dim=-1, keepdim=True))
    return torch.stack(xp, dim=-2), torch.stack(xq, dim=-2)


class Trajectory(TrajectoryView):
    """ksim.Trajectory's field names for one rollout, `[T, N, ...]` on the device (module docstring). Also a `TrajectoryView`: terms written
    against the older vocabulary (`base_qvel`, `arm_qpos`, ...) keep working - except for `command` and `done`, which take ksim's form here (a dict
    keyed "unified_command"; bool, with the signed value in `done_signed`). `per_env(fn)` below gives a term ksim's per-env `[T, ...]` view."""

    def __init__(self, traj, T: int, model, extra_observations: Optional[Dict[str, torch.Tensor]] = None):
        super().__init__(traj, T)
        if getattr(traj, "qstate", None) is None:
            raise ValueError("Trajectory needs the per-step state record: build the task with HumanoidWalkingTaskConfig(record_state=True) "
                             "(TrajBuffers(record_state=True) + kbj_traj.qstate_d at the ABI)")
        Q, A, O = L.QSTATE, L.AUX, L.OBS
        qs = traj.qstate[:T]
        self.model = model
        self.qpos = qs[..., Q["QPOS"]:Q["QPOS"] + L.NQ]
        self.qvel = qs[..., Q["QVEL"]:Q["QVEL"] + L.NV]
        self._qpos_kin = qs[..., Q["QPOS_KIN"]:Q["QPOS_KIN"] + L.NQ]
        self._xpos = self._xquat = None
        aux = traj.aux[:T]
        self.done = aux[..., A["DONE"]] != 0                         # ksim: bool [T]; the signed value stays in `done_signed`
        self.done_signed = aux[..., A["DONE"]]
        self.command = {"unified_command": aux[..., A["CMD"]:A["CMD"] + L.NCMD]}
        self.action = traj.action[:T]
        self.reward = traj.reward[:T]
        critic, actor = traj.critic_obs[:T], traj.actor_obs[:T]
        piece = lambda rows, name: rows[..., O[name][0]:O[name][0] + O[name][1]]
        from .traj_view import StepView
        bias, rng = StepView.joint_tables(model, aux.device)
        # the reference's observation dictionary (train.py:1156-1204); `noisy_*` = what the actor row carries (train.py:1360-1363)
        self.obs = {
            "joint_position": piece(critic, "JPOS") * rng + bias,
            "joint_velocity": piece(critic, "JVEL") * L.OBS_JVEL_DIV,
            "noisy_biased_joint_position": piece(actor, "JPOS") * rng + bias,
            "noisy_joint_velocity": piece(actor, "JVEL") * L.OBS_JVEL_DIV,
            "actuator_force": piece(critic, "ACTFRC") * L.OBS_ACTFRC_DIV,
            "center_of_mass_inertia": piece(critic, "CINERT"),
            "center_of_mass_velocity": piece(critic, "CVEL"),
            "base_position": piece(critic, "BASEPOS"),
            "base_orientation": piece(critic, "BASEQUAT"),
            "base_linear_velocity": piece(critic, "LINVEL"),
            "base_angular_velocity": piece(critic, "ANGVEL"),
            "imu_gyro": piece(critic, "GYRO"),
            "noisy_imu_gyro": piece(actor, "GYRO"),
            "left_foot_touch": aux[..., A["TOUCH"]:A["TOUCH"] + 1],          # [T, N, 1] as the sensor observation is ([T, 1] per env)
            "right_foot_touch": aux[..., A["TOUCH"] + 1:A["TOUCH"] + 2],
            "feet_position": piece(critic, "FEETPOS"),
            "base_height": piece(critic, "HEIGHT"),
            "projected_gravity": critic[..., O["PG"][0] + 2:O["PG"][0] + 5],
            "noisy_imu_projected_gravity": actor[..., O["PG"][0] + 2:O["PG"][0] + 5],
            "com_distance": aux[..., A["COMDIST"]],                          # scalar per step (train.py:646-659): [T, N]
        }
        for k, v in (extra_observations or {}).items():
            self.obs[k] = v

    def _kinematics(self):
        if self._xpos is None:
            xp, xq = forward_kinematics(self.model, self._qpos_kin)
            self._xpos, self._xquat = xp.to(torch.float32), xq.to(torch.float32)
        return self._xpos, self._xquat

    @property
    def xpos(self) -> torch.Tensor:
        """[T, N, nbody, 3]: body positions of the step's last forward pass (module docstring); body ids = MuJoCo's (0 world, 1 base, ...)."""
        return self._kinematics()[0]

    @property
    def xquat(self) -> torch.Tensor:
        return self._kinematics()[1]


class EnvTrajectory:
    """One env's slice of a `Trajectory` as ksim hands it to a reward term: every field `[T, ...]` (train.py's `trajectory.xquat[:, 1, :]`,
    `traj.obs["left_foot_touch"][:, 0]`, `jnp.pad(trajectory.done, ((1, 0),))` index THIS shape). Built by `per_env`."""

    def __init__(self, fields: dict):
        self.qpos, self.qvel, self.xpos, self.xquat = fields["qpos"], fields["qvel"], fields["xpos"], fields["xquat"]
        self.ctrl, self.done, self.action, self.reward = fields["ctrl"], fields["done"], fields["action"], fields["reward"]
        self.obs, self.command = fields["obs"], fields["command"]


def per_env(fn):
    """ksim evaluates a reward term per env under `jax.vmap`: the term sees `[T, ...]` fields. `per_env(fn)(trajectory)` does the same with
    `torch.vmap` over the env axis of a `[T, N, ...]` Trajectory - a `get_reward` body written exactly as the reference's (time-first indexing,
    `axis=-1` reductions, `where` instead of Python branches on values, as under jax) runs unchanged and returns `[T, N]`:

        class MyReward:            # body as it would be in train.py, jnp -> torch
            scale = 0.1
            def get_reward(self, trajectory):
                return per_env(self._one)(trajectory)
            def _one(self, traj):  # traj.qvel is [T, 26], traj.xquat[:, 1, :] the base orientation per step
                return torch.exp(-traj.qvel[:, 5].abs())
    """
    def run(trajectory: "Trajectory") -> torch.Tensor:
        fields = dict(qpos=trajectory.qpos, qvel=trajectory.qvel, xpos=trajectory.xpos, xquat=trajectory.xquat, ctrl=trajectory.ctrl, done=trajectory.done,
                      action=trajectory.action, reward=trajectory.reward, obs=dict(trajectory.obs), command=dict(trajectory.command))
        out = torch.vmap(lambda f: fn(EnvTrajectory(f)), in_dims=1, out_dims=1)(fields)
        return out
    return run


# ---- the reference's reward classes (train.py:125-506) on `Trajectory`: attribute names, constructor arguments and arithmetic as there ----
def _zero_cmd(traj: Trajectory) -> torch.Tensor:
    return torch.linalg.norm(traj.command["unified_command"][..., :3], dim=-1) < 1e-3


class _Reward:
    def __init__(self, scale: float = 1.0, **kw):
        self.scale = scale
        for k, v in kw.items():
            setattr(self, k, v)


class LinearVelocityTrackingReward(_Reward):           # train.py:269-292
    def __init__(self, scale: float, error_scale: float = 0.25):
        super().__init__(scale, error_scale=error_scale)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        base_euler = quat_to_euler(trajectory.xquat[..., 1, :]).clone()
        base_euler[..., :2] = 0.0
        base_z_quat = euler_to_quat(base_euler)
        robot_vel_cmd = torch.nn.functional.pad(trajectory.command["unified_command"][..., :2], (0, 1))
        global_vel_cmd = rotate_vector_by_quat(robot_vel_cmd, base_z_quat, inverse=False)
        vel_error = torch.linalg.norm(trajectory.qvel[..., :2] - global_vel_cmd[..., :2], dim=-1)
        error = torch.where(_zero_cmd(trajectory), vel_error, vel_error.square())
        return torch.exp(-error / self.error_scale)


class AngularVelocityReward(_Reward):                  # train.py:296-306
    def __init__(self, scale: float, error_scale: float = 0.25):
        super().__init__(scale, error_scale=error_scale)

    def get_reward(self, traj: Trajectory) -> torch.Tensor:
        return torch.exp(-(traj.qvel[..., 5] - traj.command["unified_command"][..., 2]).abs() / self.error_scale)


class XYOrientationReward(_Reward):                    # train.py:310-334
    def __init__(self, scale: float, error_scale: float = 0.03, error_scale_zero_cmd: float = 0.003):
        super().__init__(scale, error_scale=error_scale, error_scale_zero_cmd=error_scale_zero_cmd)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        e = quat_to_euler(trajectory.xquat[..., 1, :]).clone()
        e[..., 2] = 0.0
        base_xy_quat = euler_to_quat(e)
        cmd = trajectory.command["unified_command"]
        base_xy_quat_cmd = euler_to_quat(torch.stack([cmd[..., 4], cmd[..., 5], torch.zeros_like(cmd[..., 5])], dim=-1))
        quat_error = 1 - (base_xy_quat_cmd * base_xy_quat).sum(dim=-1) ** 2
        scale = torch.where(_zero_cmd(trajectory), self.error_scale_zero_cmd, self.error_scale)
        return torch.exp(-quat_error / scale)


class TerrainBaseHeightReward(_Reward):                # train.py:338-388
    def __init__(self, base_idx: int, foot_left_idx: int, foot_right_idx: int, scale: float, error_scale: float = 0.25, standard_height: float = 0.9,
                 foot_origin_height: float = 0.0):
        super().__init__(scale, base_idx=base_idx, foot_left_idx=foot_left_idx, foot_right_idx=foot_right_idx, error_scale=error_scale,
                         standard_height=standard_height, foot_origin_height=foot_origin_height)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        left = trajectory.xpos[..., self.foot_left_idx, 2] - self.foot_origin_height
        right = trajectory.xpos[..., self.foot_right_idx, 2] - self.foot_origin_height
        current_height = trajectory.xpos[..., self.base_idx, 2] - torch.minimum(left, right)
        commanded_height = trajectory.command["unified_command"][..., 3] + self.standard_height
        return torch.exp(-(current_height - commanded_height).abs() / self.error_scale)


class ArmPositionReward(_Reward):                      # train.py:217-265
    def __init__(self, joint_indices, joint_biases, scale: float, error_scale: float = 0.1):
        super().__init__(scale, joint_indices=list(joint_indices), joint_biases=list(joint_biases), error_scale=error_scale)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        dev = trajectory.qpos.device
        qpos_sel = trajectory.qpos[..., torch.tensor(self.joint_indices, device=dev) + 7]
        target = trajectory.command["unified_command"][..., 6:16] + torch.tensor(self.joint_biases, device=dev, dtype=qpos_sel.dtype)
        error = get_norm(qpos_sel - target, "l2").sum(dim=-1)
        return torch.exp(-error / self.error_scale)


class SingleFootContactReward(_Reward):                # train.py:125-154 (StatefulReward)
    def __init__(self, scale: float, ctrl_dt: float = 0.02, grace_period: float = 0.2):
        super().__init__(scale, ctrl_dt=ctrl_dt, grace_period=grace_period)

    def initial_carry(self, num_envs: int, device) -> torch.Tensor:
        return torch.zeros(num_envs, device=device)

    def get_reward_stateful(self, traj: Trajectory, reward_carry: torch.Tensor):
        left = traj.obs["left_foot_touch"][..., 0] > 0.1
        right = traj.obs["right_foot_touch"][..., 0] > 0.1
        single = left ^ right
        is_zero = _zero_cmd(traj)
        t_since, out = reward_carry, []
        for t in range(single.shape[0]):          # jax.lax.scan over time (train.py:142-149)
            t_since = torch.where(single[t], torch.zeros_like(t_since), t_since + self.ctrl_dt)
            t_since = torch.where(is_zero[t], torch.full_like(t_since, self.grace_period), t_since)
            out.append(t_since)
        grace = torch.stack(out) < self.grace_period
        return torch.where(is_zero, torch.ones_like(grace, dtype=torch.float32), grace.to(torch.float32)), t_since


class NoContactPenalty(_Reward):                       # train.py:157-165
    def get_reward(self, traj: Trajectory) -> torch.Tensor:
        left = traj.obs["left_foot_touch"][..., 0] > 0.1
        right = traj.obs["right_foot_touch"][..., 0] > 0.1
        return torch.where(_zero_cmd(traj) | left | right, 0.0, 1.0)


class FeetAirtimeReward(_Reward):                      # train.py:168-213 (StatefulReward)
    def __init__(self, scale: float, ctrl_dt: float = 0.02, touchdown_penalty: float = 0.4):
        super().__init__(scale, ctrl_dt=ctrl_dt, touchdown_penalty=touchdown_penalty)

    def initial_carry(self, num_envs: int, device):
        return torch.zeros(num_envs, 2, device=device), torch.ones(num_envs, 2, dtype=torch.bool, device=device)

    def get_reward_stateful(self, traj: Trajectory, reward_carry):
        airtime_carry, contact_carry = reward_carry
        contact = torch.stack([traj.obs["left_foot_touch"][..., 0] > 0.1, traj.obs["right_foot_touch"][..., 0] > 0.1], dim=-1)     # [T, N, 2]
        contact_or_done = contact | traj.done[..., None]
        air, rows = airtime_carry, []
        for t in range(contact.shape[0]):         # _compute_airtime's scan (train.py:182-190)
            air = torch.where(contact_or_done[t], torch.zeros_like(air), air + self.ctrl_dt)
            rows.append(air)
        airtime = torch.stack(rows)
        prev_contact = torch.cat([contact_carry[None], contact[:-1]], dim=0)
        first_contact = contact & ~prev_contact & ~traj.done[..., None]
        shifted = torch.cat([airtime_carry[None], airtime], dim=0)[:-1]        # touchdowns meet the PREVIOUS step's airtime
        reward = ((shifted - self.touchdown_penalty) * first_contact.to(torch.float32)).sum(dim=-1)
        reward = torch.where(_zero_cmd(traj), torch.zeros_like(reward), reward)
        return reward, (air, contact[-1])


class FeetOrientationReward(_Reward):                  # train.py:391-457
    def __init__(self, foot_left_idx: int, foot_right_idx: int, scale: float, error_scale: float = 0.25):
        super().__init__(scale, foot_left_idx=foot_left_idx, foot_right_idx=foot_right_idx, error_scale=error_scale)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        base_yaw = quat_to_euler(trajectory.xquat[..., 1, :])[..., 2]
        z, hp = torch.zeros_like(base_yaw), torch.full_like(base_yaw, math.pi / 2)
        straight_foot_euler = torch.stack([torch.stack([-hp, z, base_yaw - math.pi], dim=-1), torch.stack([hp, z, base_yaw - math.pi], dim=-1)], dim=-2)   # [T, N, 2, 3]
        straight_foot_quat = euler_to_quat(straight_foot_euler)
        feet_quat = trajectory.xquat[..., [self.foot_left_idx, self.foot_right_idx], :]
        rpy_error = (1 - (straight_foot_quat * feet_quat).sum(dim=-1) ** 2).sum(dim=-1)
        feet_euler = quat_to_euler(feet_quat).clone()
        feet_euler[..., 2] = 0.0
        feet_quat0 = euler_to_quat(feet_euler)
        se0 = straight_foot_euler.clone()
        se0[..., 2] = 0.0
        rp_error = (1 - (euler_to_quat(se0) * feet_quat0).sum(dim=-1) ** 2).sum(dim=-1)
        is_rotating = trajectory.command["unified_command"][..., 2].abs() > 1e-3
        return torch.exp(-torch.where(is_rotating, rp_error, rpy_error) / self.error_scale)


class COMDistanceReward(_Reward):                      # train.py:460-478
    def __init__(self, scale: float, error_scale: float = 0.25):
        super().__init__(scale, error_scale=error_scale)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        d = trajectory.obs["com_distance"]
        return torch.where((d >= 0.0) & _zero_cmd(trajectory), torch.exp(-d / self.error_scale), torch.zeros_like(d))


class BaseAccelerationReward(_Reward):                 # train.py:481-494
    def __init__(self, scale: float, error_scale: float = 1.0):
        super().__init__(scale, error_scale=error_scale)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        base_vel = trajectory.qvel[..., :6]
        padded = torch.cat([base_vel[:1], base_vel], dim=0)            # jnp.pad(mode="edge") on the time axis
        done_padded = torch.cat([trajectory.done[:1], trajectory.done], dim=0)
        acc = torch.where(done_padded[:-1, ..., None], torch.zeros_like(base_vel), padded[1:] - padded[:-1])
        return torch.exp(-acc.abs().sum(dim=-1) / self.error_scale)


class TorqueReward(_Reward):                           # train.py:497-506
    def __init__(self, scale: float, error_scale: float = 1.0):
        super().__init__(scale, error_scale=error_scale)

    def get_reward(self, trajectory: Trajectory) -> torch.Tensor:
        r = torch.exp(-trajectory.ctrl.abs() / self.error_scale).mean(dim=-1)
        return torch.where(_zero_cmd(trajectory), r, torch.ones_like(r))


def build_reward(name: str, scale: float, params: Dict[str, float], model) -> _Reward:
    """One entry of `task.get_rewards()` (host/wiring.RewardSpec: the name, scale and constructor arguments COMPILED into kbj_config, user
    overrides included) as the executable torch term of the same class the reference builds for that key (train.py:1224-1256)."""
    base, lfoot, rfoot = int(model.base_body), int(model.lfoot_body), int(model.rfoot_body)
    p = dict(params)
    if name == "linvel":
        return LinearVelocityTrackingReward(scale=scale, **p)
    if name == "angvel":
        return AngularVelocityReward(scale=scale, **p)
    if name == "roll_pitch":
        return XYOrientationReward(scale=scale, **p)
    if name == "base_height":
        return TerrainBaseHeightReward(base_idx=base, foot_left_idx=lfoot, foot_right_idx=rfoot, scale=scale, **p)
    if name == "arm_pos":
        idx = list(range(10, 20))                 # the ten arm joints in joint order = the reference's joint_names (train.py:236-247)
        return ArmPositionReward(idx, [float(model.joint_bias[i]) for i in idx], scale=scale, **p)
    if name == "single_contact":
        return SingleFootContactReward(scale=scale, **p)
    if name == "no_contact_p":
        return NoContactPenalty(scale=scale)
    if name == "feet_airtime":
        return FeetAirtimeReward(scale=scale, **p)
    if name == "feet_orient":
        return FeetOrientationReward(foot_left_idx=lfoot, foot_right_idx=rfoot, scale=scale, **p)
    if name == "com_distance":
        return COMDistanceReward(scale=scale, **p)
    if name == "base_accel":
        return BaseAccelerationReward(scale=scale, **p)
    if name == "torque":
        return TorqueReward(scale=scale, **p)
    raise KeyError(f"unknown reward {name!r}; known: {constants.REWARD_NAMES}")


def reference_rewards(model, ctrl_dt: float = 0.02) -> Dict[str, _Reward]:
    """train.py:1224-1256 `get_rewards()`: same keys, order, classes and constructor arguments, built against the model blob instead of the
    mujoco model (body ids and joint order are MuJoCo's: kbj_model.h)."""
    base, lfoot, rfoot = int(model.base_body), int(model.lfoot_body), int(model.rfoot_body)
    arm_names = ("dof_right_shoulder_pitch_03", "dof_right_shoulder_roll_03", "dof_right_shoulder_yaw_02", "dof_right_elbow_02", "dof_right_wrist_00",
                 "dof_left_shoulder_pitch_03", "dof_left_shoulder_roll_03", "dof_left_shoulder_yaw_02", "dof_left_elbow_02", "dof_left_wrist_00")     # train.py:236-247
    idx = [constants.JOINT_NAMES.index(n) for n in arm_names]          # qpos index - 7 (train.py:251)
    biases = [float(model.joint_bias[i]) for i in idx]
    return {
        "linvel": LinearVelocityTrackingReward(scale=0.2, error_scale=0.2),
        "angvel": AngularVelocityReward(scale=0.1, error_scale=0.2),
        "roll_pitch": XYOrientationReward(scale=0.2, error_scale=0.03, error_scale_zero_cmd=0.01),
        "base_height": TerrainBaseHeightReward(base_idx=base, foot_left_idx=lfoot, foot_right_idx=rfoot, scale=0.2, error_scale=0.02, standard_height=0.80,
                                               foot_origin_height=0.06),
        "arm_pos": ArmPositionReward(idx, biases, scale=0.2, error_scale=0.1),
        "single_contact": SingleFootContactReward(scale=0.1, ctrl_dt=ctrl_dt, grace_period=2.0),
        "no_contact_p": NoContactPenalty(scale=0.1),
        "feet_airtime": FeetAirtimeReward(scale=1.5, ctrl_dt=ctrl_dt, touchdown_penalty=0.4),
        "feet_orient": FeetOrientationReward(foot_left_idx=lfoot, foot_right_idx=rfoot, scale=0.1, error_scale=0.02),
        "com_distance": COMDistanceReward(scale=0.05, error_scale=0.04),
        "base_accel": BaseAccelerationReward(scale=0.1, error_scale=5.0),
        "torque": TorqueReward(scale=0.1, error_scale=5.0),
    }
