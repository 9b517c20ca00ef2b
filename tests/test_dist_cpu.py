"""world_size-2 CPU (gloo) coverage of the N>1 path: env sharding with global env ids, the gradient all-reduce and the
1/world scaling of the optimizer step. The GPU job runs the same host code over RCCL."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from kbot_joystick_amd.spec import compiler, layout as L


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from kbot_joystick_amd.host import dist as D
    from oracle.trainer import OracleTrainer
    from oracle import nn as ON
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = compiler.load_model("kbot-headless")
    n_local, off = D.env_shard(8, rank, world)
    cfg = L.default_config(num_envs=n_local, env_id_offset=off, batch_size=n_local, rollout_len=4, hidden_size=16, num_passes=1)
    rng = np.random.default_rng(0)
    params = (rng.uniform(-1, 1, ON.param_count(16)) / 4).astype(np.float32)
    tr = OracleTrainer(model, cfg, seed=3, params=params, precision="f64")
    tr.rollout()
    adv, tgt = ON.gae(torch.tensor(tr.traj["value"], dtype=torch.float64), torch.tensor(tr.traj["reward"], dtype=torch.float64),
                      torch.tensor(tr.traj["aux"][:4, :, L.AUX["DONE"]], dtype=torch.float64), cfg.gamma, cfg.lam)
    g, _ = tr.minibatch_grad(np.arange(n_local), adv, tgt)
    g_local = g.clone()
    scale = D.allreduce_grad_(g, world)
    ON.adamw_step(cfg, tr.params, tr.m, tr.v, g, 1, grad_scale=scale)
    out[rank] = dict(es=tr.env.es.copy(), g_local=g_local.numpy(), g_sum=g.numpy(), params=tr.params.numpy(), scale=scale)
    dist.destroy_process_group()


def test_two_rank_sharding_and_allreduce():
    from oracle import oracle as O
    O.build()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, 29517, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["scale"] == 0.5
    assert np.allclose(r0["g_sum"], r0["g_local"] + r1["g_local"]) and np.array_equal(r0["g_sum"], r1["g_sum"])
    assert np.array_equal(r0["params"], r1["params"])                     # replicated parameters stay in lock-step
    # the shards' env streams equal those of a single 8-env run (RNG keyed by global env id)
    model = compiler.load_model("kbot-headless")
    from oracle import nn as ON
    from oracle.trainer import OracleTrainer
    cfg = L.default_config(num_envs=8, batch_size=8, rollout_len=4, hidden_size=16, num_passes=1)
    params = (np.random.default_rng(0).uniform(-1, 1, ON.param_count(16)) / 4).astype(np.float32)
    single = OracleTrainer(model, cfg, seed=3, params=params, precision="f64")
    single.rollout()
    both = np.concatenate([r0["es"], r1["es"]])
    assert np.array_equal(single.env.es[:, 80:], both[:, 80:])            # commands, counters, episode bookkeeping: exact
    assert np.allclose(single.env.es[:, :54], both[:, :54], atol=1e-6)    # qpos/qvel: batch-size dependent BLAS rounding only


def _worker_per_pass(rank, world, port, out):
    import torch.distributed as dist
    from kbot_joystick_amd.host import dist as D
    from oracle.trainer import OracleTrainer
    from oracle import nn as ON
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = compiler.load_model("kbot-headless")
    n_local, off = D.env_shard(8, rank, world)
    cfg = L.default_config(num_envs=n_local, env_id_offset=off, batch_size=2, rollout_len=4, hidden_size=16, num_passes=2)
    params = (np.random.default_rng(0).uniform(-1, 1, ON.param_count(16)) / 4).astype(np.float32)
    tr = OracleTrainer(model, cfg, seed=3, params=params, precision="f64")
    perms = [np.arange(n_local), np.arange(n_local)[::-1].copy()]
    calls = []

    def reduce(g):
        calls.append(1)
        return D.allreduce_grad_(g, world)

    tr.train_iteration(perms, allreduce="per_pass", reduce=reduce)
    out[rank] = dict(params=tr.params.numpy(), steps=tr.opt_step, calls=len(calls))
    dist.destroy_process_group()


def test_two_rank_per_pass_accumulate():
    """KBJ_ALLREDUCE=per_pass (north_star's "once per update"): ONE all-reduce and ONE optimizer step per pass; the result equals a
    single process that averages the gradients of all (rank, minibatch) pairs of the pass."""
    from oracle import oracle as O
    from oracle import nn as ON
    from oracle.trainer import OracleTrainer
    O.build()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_per_pass, args=(2, 29519, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["steps"] == r1["steps"] == 2 and r0["calls"] == r1["calls"] == 2      # 2 passes -> 2 exchanges, 2 steps
    assert np.array_equal(r0["params"], r1["params"])
    # single-process restatement: per pass, mean of the four (shard, minibatch) gradients, one AdamW step
    model = compiler.load_model("kbot-headless")
    params = (np.random.default_rng(0).uniform(-1, 1, ON.param_count(16)) / 4).astype(np.float32)
    shards = []
    for rank in range(2):
        cfg = L.default_config(num_envs=4, env_id_offset=4 * rank, batch_size=2, rollout_len=4, hidden_size=16, num_passes=2)
        tr = OracleTrainer(model, cfg, seed=3, params=params, precision="f64")
        tr.rollout()
        adv, tgt = ON.gae(torch.tensor(tr.traj["value"], dtype=torch.float64), torch.tensor(tr.traj["reward"], dtype=torch.float64),
                          torch.tensor(tr.traj["aux"][:4, :, L.AUX["DONE"]], dtype=torch.float64), cfg.gamma, cfg.lam)
        shards.append((tr, adv, tgt, cfg))
    p = torch.tensor(params, dtype=torch.float64)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step, perm in enumerate([np.arange(4), np.arange(4)[::-1].copy()], start=1):
        g = torch.zeros_like(p)
        for tr, adv, tgt, cfg in shards:
            tr.params = p.clone()
            for mb in range(2):
                gi, _ = tr.minibatch_grad(perm[2 * mb:2 * mb + 2], adv, tgt)
                g += gi
        ON.adamw_step(shards[0][3], p, m, v, g, step, grad_scale=1.0 / 4)
    assert np.allclose(r0["params"], p.numpy(), rtol=0, atol=1e-12)


def test_env_shard_helper():
    from kbot_joystick_amd.host import dist as D
    assert D.env_shard(65536, 3, 8) == (8192, 24576)
    with pytest.raises(ValueError):
        D.env_shard(10, 0, 4)
    g = torch.ones(4)
    assert D.allreduce_grad_(g, 1) == 1.0 and torch.equal(g, torch.ones(4))


def _worker_global_adv(rank, world, port, out):
    import torch.distributed as dist
    from kbot_joystick_amd.host import dist as D
    from oracle.trainer import OracleTrainer
    from oracle import nn as ON
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = compiler.load_model("kbot-headless")
    n_local, off = D.env_shard(8, rank, world)
    cfg = L.default_config(num_envs=n_local, env_id_offset=off, batch_size=n_local, rollout_len=4, hidden_size=16, num_passes=1)
    params = (np.random.default_rng(0).uniform(-1, 1, ON.param_count(16)) / 4).astype(np.float32)
    tr = OracleTrainer(model, cfg, seed=3, params=params, precision="f64")
    tr.rollout()
    adv, tgt = ON.gae(torch.tensor(tr.traj["value"], dtype=torch.float64), torch.tensor(tr.traj["reward"], dtype=torch.float64),
                      torch.tensor(tr.traj["aux"][:4, :, L.AUX["DONE"]], dtype=torch.float64), cfg.gamma, cfg.lam)
    idx = np.arange(n_local)
    sums = D.global_advantage_sums(adv[:, idx], world)                   # the exchange of the variant: three scalars
    g, met = tr.minibatch_grad(idx, adv, tgt, adv_sums=sums)
    g_own, _ = tr.minibatch_grad(idx, adv, tgt)                           # default: this rank's own statistics
    scale = D.allreduce_grad_(g, world)
    out[rank] = dict(sums=sums.numpy(), g_avg=(g * scale).numpy(), g_own=g_own.numpy(), adv=adv.numpy(), adv_mean=met["adv_mean"], adv_std=met["adv_std"])
    dist.destroy_process_group()


def test_global_advantage_statistics_equal_the_union():
    """SURVEY section 8e's optional exchange: all-reduce (sum adv, sum adv^2, count) per minibatch. Two gloo ranks with 4 envs each: the sums
    every rank ends up with are those of the union of the two minibatches, and the AVERAGED gradient under that normalisation equals the
    single-process gradient of the 8-env minibatch - which the default (per-rank statistics) does not."""
    from oracle import oracle as O
    from oracle import nn as ON
    from oracle.trainer import OracleTrainer
    O.build()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_global_adv, args=(2, 29523, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    union = np.concatenate([r0["adv"], r1["adv"]], axis=1)
    assert np.array_equal(r0["sums"], r1["sums"]) and r0["sums"][2] == union.size
    assert np.allclose(r0["sums"][:2], [union.sum(), (union ** 2).sum()], rtol=1e-12)
    assert abs(r0["adv_mean"] - union.mean()) < 1e-12 and abs(r0["adv_std"] - union.std()) < 1e-12
    model = compiler.load_model("kbot-headless")
    cfg = L.default_config(num_envs=8, batch_size=8, rollout_len=4, hidden_size=16, num_passes=1)
    params = (np.random.default_rng(0).uniform(-1, 1, ON.param_count(16)) / 4).astype(np.float32)
    single = OracleTrainer(model, cfg, seed=3, params=params, precision="f64")
    single.rollout()
    adv, tgt = ON.gae(torch.tensor(single.traj["value"], dtype=torch.float64), torch.tensor(single.traj["reward"], dtype=torch.float64),
                      torch.tensor(single.traj["aux"][:4, :, L.AUX["DONE"]], dtype=torch.float64), cfg.gamma, cfg.lam)
    g_single, _ = single.minibatch_grad(np.arange(8), adv, tgt)
    g_single = g_single.numpy()
    scale = np.abs(g_single).max()
    assert np.abs(r0["g_avg"] - g_single).max() < 1e-6 * scale              # (qpos / qvel differ by batch-size dependent BLAS rounding, 1e-6)
    assert np.abs(0.5 * (r0["g_own"] + r1["g_own"]) - g_single).max() > 1e-4 * scale   # the default is a different (per-rank) normalisation
