"""Data parallelism by molecule: one process per GPU, one RCCL all-reduce of the flat gradient buffer.

The reference is single-device.  Molecules never interact (block-diagonal graph,
loss = mean over molecules), so the batch is dealt to the ranks after sorting by size (round-robin,
balances tuples per GPU), every rank runs the same kernels on its shard with the loss scaled by
1/B_global, and the gradients are summed with ONE collective per step over xGMI.
"""
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: str = None) -> int:
    """torchrun-style env (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, LOCAL_RANK) -> world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return world


def shard_indices(sizes: Sequence[int], world_size: int, rank: int) -> List[int]:
    """indices of the molecules of `rank`: sort by size (descending, stable) and deal round-robin."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    return sorted(order[rank::world_size])


def all_reduce_gradients(flat_grad: torch.Tensor) -> None:
    """sum over ranks; each rank's loss is already scaled by 1/B_global (MolwiseLoss.global_batch_size)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
