"""A ksim-shaped `Trajectory` for user reward terms.

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

The reference's twelve reward classes restated as torch terms on this `Trajectory` are NOT part of the product package: they live in
`examples/reference_rewards.py` (what a user copies and edits - the "edit train.py" workflow), and
`tests/test_gpu_host.py::test_reference_reward_classes_on_the_trajectory_reproduce_the_kernel` holds them against `rewards_kernel` term by
term. The built-in stack stays the kernel; such terms run only where a user passes them as `extra_rewards`.
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
