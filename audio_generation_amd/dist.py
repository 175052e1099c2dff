"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU rehearsal tests).

The codec forward shards over the batch with **no data-path collective**: every
item is independent through encoder, RVQ search and decoder.  The
only exchange step of the whole system is the gradient all-reduce of a training
step, provided here as one flattened bucket (``allreduce_mean_``) so that the
~200 small weight-norm tensors travel as a single RCCL call.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str = "nccl") -> Tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for one rank)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` batch items owned by ``rank``
    (sizes differ by at most one; the first ``n_items % world`` ranks get the extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX all-reduce of a scalar (the timing rule of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_mean_(tensors: Iterable[torch.Tensor]) -> None:
    """In-place mean over ranks of a list of same-dtype tensors through ONE
    flattened bucket (one collective instead of one per tensor)."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not dist.is_initialized():
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def gather_index_shards(index: torch.Tensor) -> List[torch.Tensor]:
    """All ranks' code indices on every rank (for a codec bitstream writer)."""
    if not dist.is_initialized():
        return [index]
    out = [torch.empty_like(index) for _ in range(dist.get_world_size())]
    dist.all_gather(out, index.contiguous())
    return out
