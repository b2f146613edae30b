"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, over xGMI).

The hot path has exactly ONE exchange per training step: a sum all-reduce of the flat gradient
arena; the 1/world average, the global-norm clip and Adam then run identically on every rank.  Parameters and BN
buffers are broadcast from rank 0 once at start-up.  Device-agnostic on purpose (the CPU/gloo tests exercise it)."""
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def exchange_gradients(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """sum-all-reduce ONE flat bucket in place (no bucketing logic needed: the arena is already contiguous)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def broadcast_state(flat_params: torch.Tensor, buffers: Iterable[torch.Tensor], src: int = 0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src, group=group)
        for b in buffers:
            dist.broadcast(b, src, group=group)


def shard_seed(base_seed: int, rank: Optional[int] = None) -> int:
    """per-rank synthetic-data seed (1234 + rank)"""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return base_seed + rank
