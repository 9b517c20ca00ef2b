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
