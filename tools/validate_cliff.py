"""Does a validation / view context slow the training iterations that follow it? (round-4 review, item 1)

One process, BASELINE configs[1] (8192 envs): K x train_iteration() timed, task.validate(), K more timed, task.view(), K more timed,
then the validation context dropped (task.close_validation()) and K more. Prints ms per iteration of every leg and the ratios.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192


def leg(task, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        task.train_iteration()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k


def main():
    cfg = launch_config(num_envs=N, robot="kbot-headless", fixed_command=(0.5, 0.0, 0.0))
    task = HumanoidWalkingTask(cfg)
    for _ in range(3):
        task.train_iteration()
    out = {"num_envs": N, "iterations_per_leg": K}
    out["before_ms"] = leg(task, K)
    t0 = time.perf_counter(); task.validate(); torch.cuda.synchronize(); out["validate_first_s"] = time.perf_counter() - t0
    out["after_validate_ms"] = leg(task, K)
    t0 = time.perf_counter(); task.validate(); torch.cuda.synchronize(); out["validate_second_s"] = time.perf_counter() - t0
    out["after_second_validate_ms"] = leg(task, K)
    t0 = time.perf_counter(); task.view(); torch.cuda.synchronize(); out["view_s"] = time.perf_counter() - t0
    out["after_view_ms"] = leg(task, K)
    if hasattr(task, "close_validation"):
        task.close_validation()
        out["after_close_ms"] = leg(task, K)
    for k in ("after_validate_ms", "after_second_validate_ms", "after_view_ms", "after_close_ms"):
        if k in out:
            out[k.replace("_ms", "_ratio")] = out[k] / out["before_ms"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
