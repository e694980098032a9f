"""Multi-GPU plumbing for the hot path.

Utterances are independent (InstanceNorm / LayerNorm / LSTM state are per
sample), so inference shards the batch over ranks with NO data-path
collective; the only collectives are a barrier around the timed region, a MAX
over ranks of the elapsed time and -- mirroring the reference's evaluate()
(train_distributed.py:119-120) -- a SUM/world_size of a scalar loss.  One
process per GPU; backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as td


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, device=None) -> bool:
    """Join the process group if WORLD_SIZE > 1.  Returns True when distributed."""
    rank, world, _ = env_rank()
    if world <= 1:
        return False
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    td.init_process_group(backend, rank=rank, world_size=world, **kw)
    return True


def shard(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of range(n_items): sizes differ by at most one,
    every item belongs to exactly one rank."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def max_over_ranks(value: float, device="cpu") -> float:
    if not (td.is_available() and td.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def mean_over_ranks(x: torch.Tensor) -> torch.Tensor:
    """all_reduce(SUM) / world_size, as evaluate() does with the validation loss."""
    if not (td.is_available() and td.is_initialized()):
        return x
    y = x.clone()
    td.all_reduce(y)
    return y / td.get_world_size()


def barrier() -> None:
    if td.is_available() and td.is_initialized():
        td.barrier()
