"""Short training run (sanity: the reward per step should rise and episodes should get longer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
from kbot_joystick_amd.spec import layout as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
x3 = len(sys.argv) > 2 and sys.argv[2] == "gemm_bf16x3"   # the same run on the flagged bf16 x3 GEMM path
print("GEMM path:", "bf16 x3 (kbj_config.gemm_bf16x3)" if x3 else "plain fp32 MFMA (default)")
task = HumanoidWalkingTask(launch_config(num_envs=8192, robot="kbot-headless", seed=1, gemm_bf16x3=x3))
for it in range(iters):
    task.train_iteration()
    if (it + 1) % 10 == 0 or it == 0:
        torch.cuda.synchronize()
        m = task.metrics.cpu().tolist()
        done = task.traj.aux[: task.T, :, L.AUX["DONE"]]
        fails = float((done < 0).float().sum()) / (task.N * task.T)
        print(f"iter {it + 1:4d}  reward/step {float(task.traj.reward.mean()):.4f}  failures/step {fails:.5f}  value_loss {m[2]:.4f}  entropy {m[3]:.2f}  "
              f"clipfrac {m[4]:.3f}  kl {m[5]:.4f}", flush=True)
