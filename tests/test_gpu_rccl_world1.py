"""The RCCL branch of the two exchange steps, executed once on the device.

Every multi-rank rehearsal so far ran on gloo (CPU tensors, or a host copy of the bucket), and at world size 1 the
exchange steps skip the collective, so ``dist.all_reduce`` on a DEVICE tensor through librccl had never executed anywhere.
Here a one-rank ``nccl`` group is created on ``cuda:0`` (``dist.init("nccl", force=True)``: ``force_collective``) and the
training step's exchange steps -- ``GradBucket.allreduce_mean_`` (one in-place all-reduce of the flat gradient bucket,
training.py:380-390 is the step it belongs to) and ``quantizer._ema_update``'s fixed-shape all-reduce of the per-code
statistics -- run on device memory: values unchanged (sum over one rank, mean over one rank), the bucket intact and
in place, no host copy.  Not a scaling measurement: it proves that librccl loads and that the in-place device path runs.
"""
import os

import pytest
import torch
import torch.distributed as dist

from audio_generation_amd import dist as agx_dist

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None


@pytest.fixture(scope="module")
def nccl_world1():
    assert not dist.is_initialized(), "another test left a process group behind"
    keep = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    torch.cuda.set_device(0)
    rank, local_rank, world = agx_dist.init("nccl", force=True)
    assert (rank, local_rank, world) == (0, 0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    yield
    agx_dist.force_collective(False)
    dist.destroy_process_group()
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_rccl_is_loaded_and_barrier_runs(nccl_world1):
    agx_dist.barrier()                      # dist.barrier(device_ids=[current device]) on RCCL
    torch.cuda.synchronize()
    maps = open("/proc/self/maps").read()
    assert "librccl" in maps or "libnccl" in maps, "the nccl backend is up but no RCCL library is mapped"
    assert agx_dist.max_over_ranks(3.25, device=DEV) == 3.25
    assert agx_dist.gather_floats(1.5, device=DEV) == [1.5]
    assert agx_dist.sum_over_ranks(2.0, device=DEV) == 2.0


def test_grad_bucket_allreduce_in_place_on_device(nccl_world1):
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n, device=DEV)) for n in (7, 1024, 3, 65537, 12)]
    bucket = agx_dist.GradBucket(params)
    assert bucket.flat.is_cuda and bucket.intact()
    gen = torch.Generator().manual_seed(1)
    for p in params:                         # what the backward kernels do: write straight into the views
        p.grad.copy_(torch.randn(p.shape, generator=gen).to(DEV))
    before = bucket.flat.clone()
    ptr = bucket.flat.data_ptr()
    bucket.allreduce_mean_()                 # ONE in-place all-reduce through RCCL, then / world
    torch.cuda.synchronize()
    assert bucket.flat.data_ptr() == ptr and bucket.intact()
    assert torch.equal(bucket.flat, before), "sum over one rank / 1 must leave every gradient bit-identical"
    # a second round on the same bucket (a training loop reuses it every step)
    bucket.zero_()
    params[1].grad.fill_(2.0)
    bucket.allreduce_mean_()
    torch.cuda.synchronize()
    assert float(params[1].grad.sum()) == 2048.0 and float(bucket.flat.sum()) == 2048.0


def test_ema_statistics_allreduce_on_device(nccl_world1):
    """``quantizer._ema_update``: the (num_quantizers, K, D+1) statistics travel through ONE all-reduce on the device; with
    one rank the update must equal the update without any collective, bit for bit."""
    from audio_generation_amd.quantizer import ResidualQuantizer
    torch.manual_seed(3)
    q, k, d = 3, 64, 32
    frames = torch.randn(2, 40, d, device=DEV)

    def updated(force):
        torch.manual_seed(4)
        m = ResidualQuantizer(num_quantizers=q, dim=d, codebook_sizes=k).to(DEV).train()
        agx_dist.force_collective(force)
        try:
            with torch.no_grad():
                m(frames, 2, update_codebook=True)      # codebook_n = 2 < Q: the collective still has the full fixed shape
        finally:
            agx_dist.force_collective(True)
        return m

    a, b = updated(True), updated(False)
    torch.cuda.synchronize()
    for name in ("codebooks", "cluster_frequency", "ema_sum"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    lo, hi = agx_dist.replica_checksums(a)
    assert lo == hi
    a.sync_from_rank0()                      # broadcast from rank 0 through RCCL (identity on one rank)
    torch.cuda.synchronize()
    assert torch.equal(a.codebooks, b.codebooks)
