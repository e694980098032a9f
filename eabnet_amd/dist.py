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
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple

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


def gather_over_ranks(value: float, device="cpu") -> List[float]:
    """The value of every rank, in rank order (a list of one without a process group)."""
    if not (td.is_available() and td.is_initialized()):
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(td.get_world_size())]
    td.all_gather(out, t)
    return [float(o.item()) for o in out]


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_local(n: int, argv: Sequence[str], env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Start ``n`` fresh copies of ``python argv...`` on this node, one per GPU, as the reference's
    ``torch.multiprocessing.spawn(main, nprocs=world_size)`` does (train_distributed.py:363-366): RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment (what
    ``torch.distributed.run`` would set).  The caller must not have touched the GPU: the children are new
    processes, nothing is forked or exec'd from an initialised one.  Rank 0 inherits stdout (its one JSON
    line is the job's); every rank inherits stderr.  Returns the worst exit code; a failing rank ends the rest."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    # this image's host driver only supports dmabuf IPC: without it RCCL's intra-node transport (and any CUDA-tensor sharing
    # across processes) fails with `hipIpcGetMemHandle: invalid argument`.  The launching shell normally exports it already.
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, *argv], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    import time

    def reap(ps, grace: float = 10.0) -> None:
        """SIGTERM, a grace period, then SIGKILL -- and always wait(): no zombies, and no rank still tearing its GPU
        context down when the caller starts the next job (a rank stuck inside an RCCL collective can ignore SIGTERM)."""
        for q in ps:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + grace
        for q in ps:
            try:
                q.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    t0, rc = time.monotonic(), 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                reap(alive)                          # one rank failed: the others would wait at a barrier forever
                alive = []
                break
        if alive and timeout is not None and time.monotonic() - t0 > timeout:
            reap(alive, grace=2.0)
            return rc or 124
        time.sleep(0.05)
    return rc
