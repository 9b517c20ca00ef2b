"""The reference's `get_rewards()` stack as torch terms on the ksim-shaped `Trajectory` (example / test material, NOT product code).

`/root/reference/train.py:125-506` defines twelve reward classes in ksim's Reward / StatefulReward protocol (`scale`, `get_reward(trajectory)` /
`initial_carry(rng)`, `get_reward_stateful(trajectory, carry)`) and `train.py:1224-1256` wires them up. This file restates them on
`kbot_joystick_amd.host.trajectory.Trajectory` - the same attribute names, every field `[T, N, ...]` - as the starting point for the
"edit train.py" workflow: copy a class, change it, pass it to `HumanoidWalkingTask(config, extra_rewards={...})` with the built-in term's
scale set to 0. The product's reward stack is the HIP `rewards_kernel`; nothing in `kbot-joystick_amd/` imports this file except the lazy
`RewardSpec.build()` convenience, and the tests hold these classes against the kernel (2e-4) and the oracle's reward scan.

`reference_rewards(model, ctrl_dt)` returns the reference's dictionary (same keys, order, classes, constructor arguments);
`build_reward(name, scale, params, model)` one entry with a configuration's own scale / arguments.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from kbot_joystick_amd.spec import constants
from kbot_joystick_amd.host.trajectory import Trajectory, get_norm, quat_to_euler, euler_to_quat, rotate_vector_by_quat


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
