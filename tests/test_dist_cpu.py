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


def _ema_worker(rank, world, port, out):
    """update_codebook=True under data parallelism (training.py:305-308, 326): every rank
    quantises its OWN shard, the per-code counts / sums are all-reduced, codebooks must stay bit-identical."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    agx_dist.init("gloo")
    from audio_generation_amd.quantizer import ResidualQuantizer
    from oracle import rvq
    torch.manual_seed(0)                                   # same initial codebooks on every rank
    rq = ResidualQuantizer(num_quantizers=3, dim=8, codebook_sizes=(16, 16, 12), quantizer_class="ema").train()
    if rank == 1:                                          # a rank that initialised differently ...
        rq.codebooks.add_(1.0)
    rq.sync_from_rank0()                                   # ... is brought back by the broadcast
    gen = torch.Generator().manual_seed(100 + rank)        # every rank its own shard
    for _ in range(3):
        x = torch.randn(2, 20, 8, generator=gen)
        _, index, _ = rvq.residual_quantize(x, rq.codebooks, sizes=rq.codebook_sizes)
        fr, ix = x.reshape(-1, 8), index.reshape(-1, 3)
        rq._ema_update(fr, ix, stats=rvq.ema_assignment_stats(fr, rq.codebooks, ix))   # the kernel's CPU statement
    lo, hi = agx_dist.replica_checksums(rq)
    # GradBucket: grads are views of one flat buffer, mean over ranks in place, views stay intact
    lin = torch.nn.Linear(5, 3)
    with torch.no_grad():
        for p in lin.parameters():
            p.fill_(0.5)
    bucket = agx_dist.GradBucket(lin.parameters())
    ptrs = [p.grad.data_ptr() for p in lin.parameters()]
    lin(torch.full((2, 5), float(rank + 1))).sum().backward()
    bucket.allreduce_mean_()
    ok = bucket.intact() and ptrs == [p.grad.data_ptr() for p in lin.parameters()]
    g = lin.weight.grad[0, 0].item()                       # d/dw = sum over the 2 rows of x = 2 (rank+1) -> mean 3
    agx_dist.barrier()
    out.put((rank, rq.codebooks.clone().numpy(), rq.ema_sum.clone().numpy(), rq.cluster_frequency.clone().numpy(),
             lo, hi, ok, g, rq._packed is None))
    dist.destroy_process_group()


def test_two_rank_codebook_update_and_grad_bucket():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_ema_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res
    for i in (1, 2, 3):                                     # codebooks, sums, frequencies: bit-identical
        assert np.array_equal(a[i], b[i])
    assert a[4] == a[5] == b[4] == b[5]                     # min == max of the state checksum, on both ranks
    assert a[6] and b[6] and a[7] == b[7] == 3.0
    assert a[8] and b[8]                                    # the packed search image was dropped by the update
    # the padding rows of the 12-codeword stage stay zero, and the update moved the codebooks
    assert np.all(a[1][2, 12:] == 0.0)


def test_ema_update_uses_pre_update_codewords_and_invalidates_pack():
    """the next stage's residual must be formed with the codeword the forward saw (pre-update),
    and any codebook write must drop the packed image."""
    from audio_generation_amd.quantizer import ResidualQuantizer
    from oracle import rvq
    torch.manual_seed(1)
    rq = ResidualQuantizer(num_quantizers=2, dim=4, codebook_sizes=8, quantizer_class="ema", ema_decay=0.5).train()
    cb0 = rq.codebooks.clone()
    x = torch.randn(1, 30, 4)
    _, index, _ = rvq.residual_quantize(x, cb0)
    rq._packed, v0 = torch.zeros(1), rq.codebooks._version
    rq._ema_update(x.reshape(-1, 4), index.reshape(-1, 2),
                   stats=rvq.ema_assignment_stats(x.reshape(-1, 4), cb0, index.reshape(-1, 2)))
    assert rq._packed is None and rq.codebooks._version > v0
    frames = x.reshape(-1, 4)
    idx0, idx1 = index.reshape(-1, 2).T
    r1 = frames - cb0[0][idx0]                              # residual against the PRE-update stage-0 codewords
    sums1 = torch.zeros(8, 4).index_add_(0, idx1, r1)
    cnt1 = torch.bincount(idx1, minlength=8).float()
    want = (0.5 * cb0[1] + 0.5 * sums1) / (0.5 + 0.5 * cnt1).clamp_min(1e-5).unsqueeze(1)
    assert torch.allclose(rq.codebooks[1], want, atol=1e-6)
    rq._packed = torch.zeros(1)
    rq.init_randn(0.3)
    assert rq._packed is None


# ------------------------------------------------------------------------------------------------ world 4
def _spawn(target, world, timeout=240):
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=timeout) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _different_codebook_n_worker(rank, world, port, out):
    """the reference draws ``codebook_n`` per PROCESS (``training.py:294``), so ranks may run a
    different number of stages in the same step; the statistics all-reduce must have one shape regardless."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    agx_dist.init("gloo")
    from audio_generation_amd.quantizer import ResidualQuantizer
    from oracle import rvq
    torch.manual_seed(0)
    rq = ResidualQuantizer(num_quantizers=4, dim=8, codebook_sizes=16, quantizer_class="ema").train()
    cb_start = rq.codebooks.clone()
    gen = torch.Generator().manual_seed(100 + rank)
    for step in range(3):
        n = (1, 3)[rank] if step < 2 else (2, 2)[rank]          # ranks disagree on codebook_n; stage 3 never runs
        x = torch.randn(2, 20, 8, generator=gen)
        _, index, _ = rvq.residual_quantize(x, rq.codebooks, codebook_n=n)
        fr, ix = x.reshape(-1, 8), index.reshape(-1, n)
        rq._ema_update(fr, ix, stats=rvq.ema_assignment_stats(fr, rq.codebooks, ix))
    lo, hi = agx_dist.replica_checksums(rq)
    agx_dist.barrier()
    out.put((rank, rq.codebooks.clone().numpy(), rq.cluster_frequency.clone().numpy(), lo, hi,
             bool(torch.equal(rq.codebooks[3], cb_start[3])), bool(torch.equal(rq.codebooks[2], cb_start[2]))))
    dist.destroy_process_group()


def test_two_ranks_with_different_codebook_n_stay_in_sync():
    import numpy as np
    a, b = _spawn(_different_codebook_n_worker, 2)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3] == a[4] == b[3] == b[4]
    assert a[5] and b[5]                # the stage no rank ran is untouched (no decay towards zero)
    assert not a[6] and not b[6]        # the stage only ONE rank ran was updated, from that rank's statistics, on both


def _train_worker(rank, world, port, out):
    """Three data-parallel training steps on CPU: the oracle's autograd statement of the generator step
    (``step.training_backward`` restated: encode -> RVQ straight-through + commit -> decode -> MSE + commit,
    ``training.py:325-347, 380``) on this rank's shard of a global batch of 256 = 8 x 32, gradients through
    ``GradBucket`` (one in-place all-reduce), Adam, EMA codebook statistics all-reduced, per-rank ``codebook_n``."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    agx_dist.init("gloo")
    from audio_generation_amd.vae import CausalVQAE
    from oracle import codec, rvq
    torch.manual_seed(0)
    kw = dict(in_channels=1, n_blocks=2, strides=(2, 4), first_block_channels=4, num_quantizers=3, codebook_size=16,
              codebook_dim=8, input_format="n c l", wavelet_decoders=False)
    model = CausalVQAE(**kw).train()
    spec = codec.CodecSpec(in_channels=1, n_blocks=2, strides=(2, 4), first_block_channels=4, codebook_dim=8,
                           wavelet_decoders=False, input_format="n c l")
    if rank:                                                            # replicas start apart ...
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01 * rank)
    agx_dist.broadcast_([p.data for p in model.parameters()] + [b for b in model.buffers()])   # ... and are synchronised
    lo, hi = agx_dist.shard_range(256, rank, world)
    bucket = agx_dist.GradBucket(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    data = torch.Generator().manual_seed(77)
    draw = torch.Generator().manual_seed(500 + rank)                    # every rank draws its OWN codebook_n
    losses = []
    for step in range(3):
        x_global = 0.1 * torch.randn(256, 1, 64, generator=data)        # same global batch on every rank
        x = x_global[lo:hi]
        n = int(torch.randint(2, 4, (1,), generator=draw))              # training.py:294: randint(2, Q + 1)
        sd = dict(model.state_dict(keep_vars=True))
        z = codec.encode_latents(x, sd, spec)
        zq, index, commit = rvq.residual_quantize_train(z, model.quantizer.codebooks.detach(), codebook_n=n)
        y = codec.decode_latents(zq, sd, spec)
        loss = ((x - y) ** 2).mean() + commit
        bucket.zero_()
        loss.backward()
        bucket.allreduce_mean_()
        opt.step()
        fr, ix = z.detach().reshape(-1, 8), index.reshape(-1, n)
        model.quantizer._ema_update(fr, ix, stats=rvq.ema_assignment_stats(fr, model.quantizer.codebooks.detach(), ix))
        losses.append(float(loss))
    cmin, cmax = agx_dist.replica_checksums(model)
    times = agx_dist.gather_floats(10.0 + rank)
    agx_dist.barrier()
    out.put((rank, lo, hi, cmin, cmax, bucket.intact(), times, losses))
    dist.destroy_process_group()


def test_four_rank_training_steps_keep_replicas_in_sync():
    res = _spawn(_train_worker, 4, timeout=300)
    spans = [(r[1], r[2]) for r in res]
    assert spans == [(0, 64), (64, 128), (128, 192), (192, 256)]        # 256 = 8 x 32 over 4 ranks: 64 each
    cmin = {r[3] for r in res} | {r[4] for r in res}
    assert len(cmin) == 1                                               # min == max of the state checksum, on every rank
    assert all(r[5] for r in res)
    assert all(r[6] == [10.0, 11.0, 12.0, 13.0] for r in res)           # per-rank timings visible on every rank
    assert all(all(v == v and v < 1e3 for v in r[7]) for r in res)
    # 8 ranks x 32 clips (the config-5 layout) without running them: the split itself
    assert [agx_dist.shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]


def test_quantizer_says_what_it_ignores():
    """``use_som`` / ``prioritize_early`` are accepted (the shipped YAML sets ``use_som: True``) but the
    SOM update / early-stage prioritisation are not performed -- one warning each, never silence."""
    import warnings
    from audio_generation_amd import quantizer as qz
    from oracle import rvq
    qz._WARNED.clear()
    rq = qz.ResidualQuantizer(num_quantizers=2, dim=4, codebook_sizes=8, use_som=True, som_kernel_type="hard").train()
    x = torch.randn(1, 10, 4)

    class _Ops:                                                          # host logic only: the search itself needs the GPU
        @staticmethod
        def rvq_forward(xx, cb, packed, q_used, layout):
            xq, idx, commit = rvq.residual_quantize(xx, cb, codebook_n=q_used)
            return xq, idx, None, commit

        @staticmethod
        def rvq_ema_stats(frames, cb, index):
            return rvq.ema_assignment_stats(frames, cb, index)

    real_ops, real_pack = qz.ops, rq._packed_codebooks
    qz.ops, rq._packed_codebooks = _Ops, (lambda: None)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            rq(x)                                                        # eval-style call: nothing ignored yet
            assert not w
            rq(x, update_codebook=True)                                  # SOM update would have run here
            rq(x, update_codebook=True)                                  # ... once per process
            rq(x, prioritize_early=True)
            rq(x, prioritize_early=True)
        msgs = [str(m.message) for m in w]
    finally:
        qz.ops, rq._packed_codebooks = real_ops, real_pack
    assert len(msgs) == 2 and "SOM" in msgs[0] and "prioritize_early" in msgs[1], msgs
    assert all("som_quantizer" in m and "parity unpinned" in m for m in msgs)


def _force_worker(port, out):
    """One rank, gloo, ``force=True``: the exchange steps call the backend although the group has a single member."""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert agx_dist.init("gloo", force=True) == (0, 0, 1) and dist.is_initialized()
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.all_reduce = counting
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 64, 3)]
    bucket = agx_dist.GradBucket(params)
    for p in params:
        p.grad.copy_(torch.arange(p.numel(), dtype=torch.float32))
    before = bucket.flat.clone()
    bucket.allreduce_mean_()
    stats = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    agx_dist.allreduce_sum_(stats)
    forced = calls["n"]
    agx_dist.force_collective(False)
    bucket.allreduce_mean_()          # world 1, not forced: no collective
    dist.all_reduce = real
    agx_dist.barrier()
    out.put((forced, calls["n"], bool(torch.equal(bucket.flat, before)), bucket.intact(),
             bool(torch.equal(stats, torch.arange(12, dtype=torch.float32).reshape(3, 4)))))
    dist.destroy_process_group()


def test_force_collective_runs_the_exchange_steps_at_world_size_one():
    """The switch tests/test_gpu_rccl_world1.py uses to execute the RCCL branch on one GPU, rehearsed on gloo."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(_free_port(), out))
    p.start()
    forced, total, same, intact, stats_same = out.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert forced == 2 and total == 2 and same and intact and stats_same
