import sys, os
sys.path.insert(0, os.getcwd())
import torch
from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
from kbot_joystick_amd.spec import layout as L
cfg = launch_config(num_envs=8192, robot="kbot", terrain="sine", seed=3, actor_mirror_loss_scale=1.0, critic_mirror_loss_scale=0.01,
                    use_lr_decay=True, lr_decay_steps=48 * 60, log_reward_components=True)
task = HumanoidWalkingTask(cfg)
import time; t0 = time.time()
for it in range(60):
    task.train_iteration()
    if (it + 1) % 20 == 0:
        torch.cuda.synchronize()
        m = task.metrics.cpu().tolist()
        assert all(x == x for x in m), m
        print(f"iter {it+1} reward/step {float(task.traj.reward.mean()):.4f} loss {m[0]:.4f} mirror {m[8]:.4f}/{m[9]:.5f} entropy {m[3]:.2f} "
              f"{8192*100*(it+1)/(time.time()-t0):.3e} env-steps/s", flush=True)
print({k: round(v, 4) for k, v in task.reward_components().items()})
assert torch.isfinite(task.params).all() and torch.isfinite(task.traj.critic_obs).all()
print("stress ok")
