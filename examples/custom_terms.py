"""Editing the task the way the reference is edited: user-written terms in ksim's element protocols, bodies written per env
with the reference's attribute names, plugged into `HumanoidWalkingTask`. Everything here runs on the GPU as torch code around the library's kernels.

    python examples/custom_terms.py [iterations]

  Reward            `get_reward(trajectory) -> [T]`                          train.py:161-165   -> extra_rewards     (needs record_state for qpos / xpos ...)
  StatefulReward    `initial_carry`, `get_reward_stateful(traj, carry)`      train.py:135-154   -> extra_rewards
  Termination       `__call__(state, curriculum_level) -> {-1, 0, 1}`        train.py:817-823   -> extra_terminations
  Reset             `__call__(data, curriculum_level, rng) -> data`          train.py:833-844   -> extra_resets
  Observation       `observe(state, curriculum_level, rng) -> [d]`           train.py:682-707   -> extra_observations (optionally a NETWORK input)
  Command           `initial_command(...)`, `__call__(prev_command, ...)`    train.py:724, 768  -> command
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
from kbot_joystick_amd.host.traj_view import per_env_reset, per_env_state, update_data_field
from kbot_joystick_amd.host.trajectory import per_env, quat_to_euler


class KneeBendReward:                      # a NEW reward, body per env on ksim's Trajectory fields
    scale = 0.05

    def get_reward(self, trajectory):
        return per_env(self._one)(trajectory)

    def _one(self, traj):                  # traj.qpos [T, 27]: 7 free-joint positions, then the 20 joints in train.py:24-45 order (knees: 3 and 8)
        knees = traj.qpos[:, [7 + 3, 7 + 8]]
        return torch.exp(-(knees.abs() - 0.4).square().sum(dim=-1) / 0.1)


class TorsoRollTermination:                # a second failure condition beside the built-in three
    def __call__(self, state, curriculum_level):
        return per_env_state(self._one)(state, curriculum_level)

    def _one(self, state, curriculum_level):
        roll = quat_to_euler(state.xquat[2])[0]            # body 2 = torso
        return torch.where(roll.abs() > 0.6, -1, 0)


class StartFacingForward:                  # a Reset behind the built-in list: episodes start with zero heading (undoes RandomHeadingReset)
    def __call__(self, data, curriculum_level, rng):
        return per_env_reset(self._one)(data, curriculum_level, rng)

    def _one(self, data, curriculum_level, rng):
        qpos = torch.cat([data.qpos[:3], torch.tensor([1.0, 0.0, 0.0, 0.0], device=data.qpos.device), data.qpos[7:]])
        return update_data_field(data, "qpos", qpos)


class FeetSpread:                          # an Observation fed INTO the critic (one extra input column)
    def observe(self, state, curriculum_level, rng):
        fp = state.feet_position                            # [N, 6]: left xyz, right xyz in the base's yaw frame
        return (fp[:, 1] - fp[:, 4])[:, None]


class ForwardRamp:                         # a Command: walk forward, faster with the episode's age
    def initial_command(self, state, curriculum_level, rng):
        cmd = torch.zeros(state.N, 16, device=state.done.device)
        cmd[:, 0] = 0.3
        return cmd

    def __call__(self, prev_command, state, curriculum_level, rng):
        out = prev_command.clone()
        out[:, 0] = torch.clamp(prev_command[:, 0] + 0.002, max=1.0)
        return out


def main(iterations: int = 3, num_envs: int = 256):
    cfg = launch_config(num_envs=num_envs, batch_size=min(64, num_envs), hidden_size=64, rollout_length_seconds=1.0, robot="kbot-headless", record_state=True,
                        extra_critic_obs=1, reward_scales={"torque": 0.05})
    task = HumanoidWalkingTask(cfg, extra_rewards={"knee_bend": KneeBendReward()}, extra_terminations={"torso_roll": TorsoRollTermination()},
                               extra_resets=[StartFacingForward()], extra_observations={"feet_spread": (FeetSpread(), "critic")}, command=ForwardRamp())
    for it in range(iterations):
        task.train_iteration()
        sc = task.scalars()
        print(f"iter {it + 1}: reward/step {sc['train/reward_per_step']:.4f}  knee_bend {task.extra_reward_means['knee_bend']:.4f}  "
              f"failures/step {sc['train/failures_per_step']:.4f}  loss {sc['train/loss']:.4f}", flush=True)
    tr = task.trajectory()
    fresh = tr.done[:-1]                                                  # step t finished the episode: the observation of step t + 1 is the fresh episode's first
    yaw = quat_to_euler(tr.obs["base_orientation"][1:][fresh])[..., 2]
    print(f"{int(fresh.sum())} episodes started inside the last rollout, all facing forward: max |yaw| {float(yaw.abs().max()) if yaw.numel() else 0.0:.2e}; "
          f"command vx in [{float(tr.command['unified_command'][..., 0].min()):.2f}, {float(tr.command['unified_command'][..., 0].max()):.2f}]")
    assert yaw.numel() == 0 or float(yaw.abs().max()) < 1e-3
    assert torch.isfinite(task.params).all()
    task.close()
    print("custom terms ok")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
