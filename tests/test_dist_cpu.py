"""world_size-2 rehearsal of the multi-GPU plumbing on CPU (gloo)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audio_generation_amd import dist as agx_dist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = agx_dist.init("gloo")
    assert (r, lr, w) == (rank, rank, world)
    # batch sharding: disjoint, covering, balanced
    lo, hi = agx_dist.shard_range(7, rank, world)
    # timing rule: the slowest rank defines the step time; throughput sums over ranks
    slow = agx_dist.max_over_ranks(1.0 + rank)
    total = agx_dist.sum_over_ranks(float(hi - lo))
    # gradient exchange: one flattened bucket, mean over ranks
    grads = [torch.full((3, 2), float(rank + 1)), torch.full((5,), 10.0 * (rank + 1))]
    agx_dist.allreduce_mean_(grads)
    shards = agx_dist.gather_index_shards(torch.full((2, 3), rank, dtype=torch.int64))
    agx_dist.barrier()
    out.put((rank, lo, hi, slow, total, grads[0][0, 0].item(), grads[1][0].item(),
             [int(s[0, 0]) for s in shards]))
    dist.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, slow0, tot0, g0a, g0b, sh0), (r1, lo1, hi1, slow1, tot1, g1a, g1b, sh1) = results
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)
    assert slow0 == slow1 == 2.0 and tot0 == tot1 == 7.0
    assert g0a == g1a == 1.5 and g0b == g1b == 15.0
    assert sh0 == sh1 == [0, 1]


def test_shard_range_properties():
    for n in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [agx_dist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert agx_dist.max_over_ranks(3.5) == 3.5  # single process: identity
