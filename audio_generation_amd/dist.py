"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU rehearsal tests).

The codec forward shards over the batch with **no data-path collective**: every
item is independent through encoder, RVQ search and decoder.  The
exchange steps of the whole system belong to a training step: the gradient
all-reduce, one flat bucket allocated once (``GradBucket``: the ~200 small
weight-norm tensors travel as a single in-place RCCL call), and -- with
``update_codebook=True`` -- one all-reduce of the RVQ's per-code counts and sums
(``allreduce_sum_``, called from ``quantizer._ema_update``).
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


# ``force_collective``: run the exchange steps through the backend even when the group has ONE rank (they are identities
# there).  Off by default -- a single-GPU run pays nothing --; switched on by ``init(..., force=True)`` or
# ``AGX_FORCE_COLLECTIVE=1`` so that the in-place device path through librccl (flat gradient bucket, EMA statistics) can be
# executed and checked on a one-GPU box before an 8-GPU node ever sees it (tests/test_gpu_rccl_world1.py).
_force_collective = os.environ.get("AGX_FORCE_COLLECTIVE", "0") == "1"


def force_collective(on: bool = True) -> None:
    global _force_collective
    _force_collective = bool(on)


def _collective_needed() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _force_collective)


def init(backend: str = "nccl", force: bool = False) -> Tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for one rank unless ``force``: then a
    one-rank group is created on 127.0.0.1 and every exchange step really calls the backend)."""
    rank, local_rank, world = env_world()
    if force:
        force_collective(True)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and torch.cuda.is_available():
            # bind the communicator to this rank's device up front (no lazy guess from the first tensor; barrier() needs it)
            kw["device_id"] = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` batch items owned by ``rank``
    (sizes differ by at most one; the first ``n_items % world`` ranks get the extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier() -> None:
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])     # RCCL: name the device, never let it guess
        else:
            dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX all-reduce of a scalar (the timing rule of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device="cpu") -> List[float]:
    """Every rank's scalar, in rank order, on every rank (bench.py prints per-rank step times with it)."""
    if not dist.is_initialized():
        return [float(value)]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def sum_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_sum_(t: torch.Tensor) -> None:
    """In-place sum over ranks (no-op for one rank unless ``force_collective``).  Used for the RVQ's per-code counts and sums."""
    if _collective_needed():
        if t.is_cuda and dist.get_backend() == "gloo":
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)


def broadcast_(tensors: Iterable[torch.Tensor], src: int = 0) -> None:
    """In-place broadcast of a list of tensors from ``src`` (no-op for one rank)."""
    if not _collective_needed():
        return
    for t in tensors:
        if t.is_cuda and dist.get_backend() == "gloo":
            host = t.cpu()
            dist.broadcast(host, src=src)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src)


def allreduce_mean_(tensors: Iterable[torch.Tensor]) -> None:
    """In-place mean over ranks of a list of same-dtype tensors through ONE
    flattened bucket (one collective instead of one per tensor).  One-off form: it
    concatenates and copies back; a training loop uses ``GradBucket`` instead."""
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


class GradBucket:
    """The gradient exchange step of a data-parallel training step (SURVEY 5 / 8e;
    ``training.py:380-390`` is the step it sits in): ONE flat fp32 buffer allocated at
    init, every parameter's ``.grad`` a view into it, so the all-reduce is a single
    in-place RCCL call on memory the backward kernels wrote directly -- no ``cat``,
    no copy-back, nothing allocated per step.

    ``allreduce_mean_()`` after the backward; ``zero_()`` instead of
    ``optimizer.zero_grad()`` (which must not replace the views: call it with
    ``set_to_none=False`` or not at all).  Several buckets (generator, each
    discriminator) can be reduced on their own as soon as their backward is done.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameter")
        dev, dt = self.params[0].device, self.params[0].dtype
        offs, n = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("GradBucket: parameters must share device and dtype")
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4       # 16-byte aligned views (float4 stores of the dW kernels)
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        for p, o in zip(self.params, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def intact(self) -> bool:
        """True while every ``.grad`` still aliases the bucket (an optimizer's ``zero_grad(set_to_none=True)``
        or a ``p.grad = ...`` assignment breaks it)."""
        lo = self.flat.data_ptr()
        hi = lo + self.flat.numel() * self.flat.element_size()
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params)

    def zero_(self) -> None:
        self.flat.zero_()

    def allreduce_mean_(self) -> None:
        if not self.intact():
            raise RuntimeError("GradBucket: a parameter's .grad no longer aliases the bucket "
                               "(use bucket.zero_() or zero_grad(set_to_none=False))")
        if _collective_needed():
            if self.flat.is_cuda and dist.get_backend() == "gloo":      # CPU rehearsal of the RCCL step
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def replica_checksums(module: torch.nn.Module) -> Tuple[float, float]:
    """(min, max) over ranks of a checksum of ALL state -- parameters and buffers (the RVQ's EMA codebooks,
    sums and frequencies are buffers).  Replicas are in sync iff min == max."""
    chk = torch.zeros((), dtype=torch.float64)
    state = list(module.parameters()) + list(module.buffers())
    for t in state:
        chk = chk + t.detach().double().abs().sum().cpu()
    if not dist.is_initialized():
        return float(chk), float(chk)
    dev = state[0].device if state and dist.get_backend() == "nccl" else "cpu"
    lo, hi = chk.clone().to(dev).reshape(1), chk.clone().to(dev).reshape(1)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def gather_index_shards(index: torch.Tensor) -> List[torch.Tensor]:
    """All ranks' code indices on every rank (for a codec bitstream writer)."""
    if not dist.is_initialized():
        return [index]
    out = [torch.empty_like(index) for _ in range(dist.get_world_size())]
    dist.all_gather(out, index.contiguous())
    return out
