"""The one exchange step of the hot path: sum-all-reduce of the flat fp32 gradient over the data-parallel ranks
(RCCL over xGMI on the GPU job, gloo in the CPU tests). The 1/world scale is applied inside kbj_adamw_step."""
from __future__ import annotations

import torch


def env_shard(num_envs_total: int, rank: int, world_size: int) -> tuple[int, int]:
    """(num_envs_local, env_id_offset): rank r owns global env ids [r*N, (r+1)*N) so that every env's RNG streams
    (threefry key = (seed, global env id)) are independent of the number of GPUs."""
    if num_envs_total % world_size != 0:
        raise ValueError("num_envs must be divisible by the number of ranks")
    n = num_envs_total // world_size
    return n, rank * n


FORCE_COLLECTIVE = False      # run the collective even at world_size 1 (tests: the RCCL leg on a one-GPU box)
TIMING = None                 # set to a list to collect (start, end) CUDA event pairs around every gradient all-reduce (bench.py)


def allreduce_grad_overlapped_(ctx, grad: torch.Tensor, n_actor: int, world_size: int, comm_stream) -> float:
    """The same exchange in two pieces: the actor's slice grad[:n_actor] on `comm_stream` as soon as kbj_ppo_grad has finished it
    (kbj_stream_wait_actor_grad: ~0.5 ms before the critic's), the critic's slice on the current stream behind the whole call; the
    current stream then waits for `comm_stream`. Same sums, same result as allreduce_grad_."""
    if world_size > 1 or FORCE_COLLECTIVE:
        import torch.distributed as dist
        ctx.stream_wait_actor_grad(comm_stream.cuda_stream)
        with torch.cuda.stream(comm_stream):
            dist.all_reduce(grad[:n_actor], op=dist.ReduceOp.SUM)
        dist.all_reduce(grad[n_actor:], op=dist.ReduceOp.SUM)
        torch.cuda.current_stream().wait_stream(comm_stream)
    return 1.0 / world_size


def allreduce_grad_(grad: torch.Tensor, world_size: int) -> float:
    """In-place SUM all-reduce on the current stream; returns the scale (1/world) the optimizer step must apply."""
    if world_size > 1 or FORCE_COLLECTIVE:
        import torch.distributed as dist
        if TIMING is not None and grad.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
            e1.record()
            TIMING.append((e0, e1))
        else:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


def global_advantage_sums(adv_minibatch: torch.Tensor, world_size: int) -> torch.Tensor:
    """(sum adv, sum adv^2, count) of the GLOBAL minibatch as three float64 on the tensor's device: this rank's sums all-reduced over the
    data-parallel ranks. Fed to kbj_set_advantage_sums (product) / ppo_loss(adv_sums=...) (oracle): every rank then
    normalises its advantages with the same mean and variance, and the averaged gradient equals the single-process gradient of the union."""
    a = adv_minibatch.to(torch.float64)
    sums = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), dtype=torch.float64, device=a.device)])
    if world_size > 1 or FORCE_COLLECTIVE:
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums
