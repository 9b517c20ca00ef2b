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


def allreduce_grad_(grad: torch.Tensor, world_size: int) -> float:
    """In-place SUM all-reduce; returns the scale (1/world) the optimizer step must apply."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return 1.0 / world_size
