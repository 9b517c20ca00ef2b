import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from kbot_joystick_amd.host.task import HumanoidWalkingTask, HumanoidWalkingTaskConfig, launch_config
for name, cfg in (("launch block (4096 envs, H 256)", launch_config()),
                  ("dataclass defaults (H 128, mirror losses on)", HumanoidWalkingTaskConfig()),
                  ("H 64, 1024 envs, batch 256", launch_config(num_envs=1024, batch_size=256, hidden_size=64))):
    task = HumanoidWalkingTask(cfg)
    task.train_iteration(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(2): task.train_iteration()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 2
    m = task.metrics.cpu().tolist()
    assert all(x == x for x in m) and torch.isfinite(task.params).all()
    print(f"{name}: {task.N * task.T / dt:.3e} env-steps/s, loss {m[0]:.4f}", flush=True)
    task.ctx.close()
