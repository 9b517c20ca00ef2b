"""Two data-parallel ranks of the PRODUCT path on one GPU (both processes on cuda:0, gloo moving the gradient through the host):
the N > 1 host logic — env sharding by global env id, gradient all-reduce, 1/world scaling — driven through libkbj.so.
The 8-GPU job runs the same code with the nccl (= RCCL) backend over xGMI."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, overlap=False):
    import torch.distributed as dist
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = launch_config(num_envs=128, batch_size=32, hidden_size=64, rollout_length_seconds=0.2, robot="kbot-headless", seed=6, num_passes=1,
                        overlap_allreduce=overlap, deterministic=True)
    task = HumanoidWalkingTask(cfg, device=torch.device("cuda", 0), rank=rank, world_size=world)
    assert task.N == 64 and task.kcfg.env_id_offset == 64 * rank
    task.train_iteration()
    torch.cuda.synchronize()
    ep, es = task.ctx.env_get_state()
    out[rank] = dict(params=task.params.cpu().numpy(), es=es, reward=task.traj.reward.cpu().numpy(), steps=task.opt_step)
    task.ctx.close()
    dist.destroy_process_group()


def test_two_ranks_share_one_gpu():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, 29531, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["steps"] == r1["steps"] == 2                                   # 64 envs / batch 32, one pass
    assert np.isfinite(r0["params"]).all() and np.array_equal(r0["params"], r1["params"])      # replicated parameters stay identical
    assert not np.array_equal(r0["es"][:, :27], r1["es"][:, :27])            # different env shards (global env ids 0..63 / 64..127)
    # the sharded rollout equals the single-process rollout of all 128 envs (RNG keyed by global env id; first rollout, same init)
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    cfg = launch_config(num_envs=128, batch_size=32, hidden_size=64, rollout_length_seconds=0.2, robot="kbot-headless", seed=6, num_passes=1)
    single = HumanoidWalkingTask(cfg)
    single.rollout()
    torch.cuda.synchronize()
    both = np.concatenate([r0["reward"], r1["reward"]], axis=1)
    assert np.allclose(single.traj.reward.cpu().numpy(), both, atol=1e-5)
    single.ctx.close()


def test_overlapped_actor_slice_exchange_gives_identical_parameters():
    """overlap_allreduce: the actor's gradient slice is all-reduced on a second stream as soon as kbj_ppo_grad has finished it
    (kbj_stream_wait_actor_grad), the critic's slice behind the call. Two ranks on one GPU, deterministic reductions: the parameters
    equal those of the plain exchange bit for bit."""
    mgr = mp.Manager()
    plain, over = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, 29537, plain, False), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, 29539, over, True), nprocs=2, join=True)
    assert np.array_equal(over[0]["params"], over[1]["params"])
    assert np.array_equal(plain[0]["params"], over[0]["params"]) and plain[0]["steps"] == over[0]["steps"] == 2
