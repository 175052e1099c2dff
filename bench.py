#!/usr/bin/env python3
"""Headline benchmark: 24 kHz samples/s through encode -> RVQ -> decode.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N > 1 through torch.distributed.run;
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment).  A "step"
is one forward of the "Soundstream default" codec (BASELINE.json configs[1]:
8 x 1024 x 512 RVQ, strides 2,4,5,8, fp32) over one batch of 32 synthetic
72 000-sample clips resident in HBM; the batch dimension shards over ranks with
no data-path collective (weak scaling: 32 clips per GPU).  Rank 0 prints ONE
JSON line.

Besides the contract fields the line carries
  roofline     -- the dominant kernel (by time) of the forward: achieved vs peak,
                  per-launch durations measured with HIP events on the launch stream;
  cpu_baseline -- the CPU oracle (torch fp32 conv stacks + the exact RVQ) timed on
                  this box's host cores on a bounded sample of the same workload;
  parity       -- GPU vs oracle on that sample (index equality on identical
                  latents, waveform RMS), so every bench line is also a parity check.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (the bf16x3 kernels execute 6 bf16 flops per fp32 flop)
PEAK_HBM_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E spec (measured copy ceiling ~6300)
CLIP = 72000               # utils.py:149 collator clip length = 3 s @ 24 kHz
BATCH_PER_GPU = 32
MODEL_KW = dict(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
                codebook_dim=512, input_format="n c l", wavelet_decoders=False)


# ----------------------------------------------------------------------- work model
def conv_work(desc):
    """(executed MACs, reference-counted MACs, layer-boundary bytes) of one conv launch.

    executed  = what the polyphase form needs: B * Cin * J * (q*Cout) * Lt
    reference = the (2s+1)-tap count on the upsampled signal the reference executes
    bytes     = read the unpadded input once + write the output once (SURVEY 8d byte model)
    """
    from audio_generation_amd import _lib, ops
    b, cin, cout, lin, k, s = desc.batch, desc.c_in, desc.c_out, desc.l_in, desc.kernel, desc.stride
    lout = ops.conv_out_len(desc)
    if desc.kind == _lib.CONV_UPSAMPLE:
        pl = (k - 1) // 2
        jmin, jmax = (-pl) // s, (s - 1 + k - 1 - pl) // s
        executed = b * cin * (jmax - jmin + 1) * s * cout * lin
        reference = b * cin * cout * k * lout
    elif desc.kind == _lib.CONV_TRANSPOSED:
        executed = b * cin * (-(-k // s)) * s * cout * lin
        reference = b * cin * cout * k * lin
    else:
        executed = reference = b * cin * cout * k * lout
    nbytes = 4 * b * (cin * lin + cout * lout)
    if desc.epilogue & _lib.EPI_RESIDUAL:
        nbytes += 4 * b * cout * lout
    return executed, reference, nbytes


class LaunchTimer:
    """ops observer: HIP events (torch.cuda.Event on the launch stream) around every C-ABI call."""

    def __init__(self):
        self.records = []

    def begin(self, kind, info):
        from audio_generation_amd import ops
        if kind == "rvq":
            b, t, d, k, q = info
            name, work = "rvq_forward", (b * t * q * k * d, b * t * q * k * d, 4 * b * t * d * 2 + 8 * b * t * q)
        elif kind == "other":           # (name, algorithmic bytes[, MACs])
            macs = info[2] if len(info) > 2 else 0
            name, work = info[0], (macs, macs, info[1])
        elif kind == "resblock":
            name = ops.resblock_kernel_name(info)
            e1_, r1_, _ = conv_work(info)
            k1 = ops.conv_desc(info.kind, info.batch, info.c_out, info.c_out, info.l_in, 1)
            e2_, r2_, nb = conv_work(k1)            # bytes: one read + one write of the activation
            work = (e1_ + e2_, r1_ + r2_, nb)
        else:
            name, work = ops.conv_kernel_name(info), conv_work(info)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, work, e0, e1)

    def end(self, tok):
        tok[3].record()
        self.records.append(tok)

    def summary(self, steps):
        torch.cuda.synchronize()
        per = {}
        for name, (ex, ref, nb), e0, e1 in self.records:
            r = per.setdefault(name, dict(ms=0.0, launches=0, macs=0, ref_macs=0, bytes=0))
            r["ms"] += e0.elapsed_time(e1)
            r["launches"] += 1
            r["macs"] += ex
            r["ref_macs"] += ref
            r["bytes"] += nb
        for r in per.values():
            r["avg_us"] = 1e3 * r["ms"] / r["launches"]
            r["tflops"] = 2e-9 * r["macs"] / r["ms"] if r["ms"] > 0 else 0.0
            r["gbps"] = 1e-6 * r["bytes"] / r["ms"] if r["ms"] > 0 else 0.0
            r["launches_per_step"] = r["launches"] / steps
            r["ms_per_step"] = r["ms"] / steps
        return per


def encoder_roofline(model, x, bsz, enc_bytes, reps=10):
    """The conv encoder on its own (north_star: '>= 40 % HBM-bandwidth roofline on the conv encoder'): HIP events
    around ``reps`` eager encoder passes (30 convs in 18 launches), against BOTH ceilings -- the fp32 matrix pipe
    on the executed FLOPs (390.4 kFLOP/sample) and HBM on SURVEY 8(d)'s layer-boundary byte model (4 864 B/sample).
    In fp32 the FLOP ceiling binds first: 157.3 TFLOP/s / 390.4 kFLOP = 403 Msamples/s = 24.5 % of HBM on that model."""
    with torch.no_grad():
        for _ in range(2):
            model._run_encoders(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            model._run_encoders(x)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rate = bsz * CLIP / (1e-3 * ms)
    flop_per_sample = 2.0 * 195194          # 14 053.9 MMAC per 72 000-sample item
    tflops = rate * flop_per_sample * 1e-12
    return {"ms": ms, "samples_per_s": rate, "tflops": tflops, "frac_of_fp32_peak": tflops / PEAK_FP32_TFLOPS,
            "layer_boundary_GBps": rate * enc_bytes * 1e-9, "frac_of_hbm_peak": rate * enc_bytes * 1e-9 / PEAK_HBM_GBPS,
            "hbm_frac_ceiling_in_fp32": PEAK_FP32_TFLOPS * 1e12 / flop_per_sample * enc_bytes * 1e-9 / PEAK_HBM_GBPS,
            "bytes_per_sample_model": enc_bytes, "launch": "eager"}


# ----------------------------------------------------------------------------- main
def encoder_roofline_bf16x3(model, x, bsz):
    """The encoder on the bf16x3 kernels (NOT the supported configuration: it decides integers): same byte model, the
    FLOP ceiling is the bf16 matrix pipe at 6 bf16 flops per fp32-equivalent flop."""
    model.set_conv_arithmetic(decoders="fp32", encoders="bf16x3")
    try:
        r = encoder_roofline(model, x, bsz, 4864.0)
    finally:
        model.set_conv_arithmetic(decoders="fp32", encoders="fp32")
    r["arithmetic"] = "bf16x3"
    r["frac_of_bf16_peak_executed"] = 6 * r["tflops"] / PEAK_BF16_TFLOPS      # = tflops / (2500 / 6)
    r["peak_fp32_equivalent_tflops"] = PEAK_BF16_TFLOPS / 6
    r.pop("frac_of_fp32_peak"), r.pop("hbm_frac_ceiling_in_fp32")
    return r


def arithmetic_roofline(per, ms_per_step):
    """Roofline of a forward that mixes the two matrix pipes (``--arith mixed`` / all-bf16x3).  Kernels whose name ends in
    ``:bf16x3`` execute six bf16 MFMA flops per fp32-equivalent flop: their ceiling is 2500 / 6 = 416.7 fp32-equivalent
    TFLOP/s; the others run on the fp32-input MFMA (157.3).  Reported: the dominant bf16x3 kernel against ITS peak, and the
    step against the blended floor  t = flop_bf16x3 / 416.7 T + flop_fp32 / 157.3 T  (never a bf16x3 figure over the fp32 peak)."""
    peak_b3 = PEAK_BF16_TFLOPS / 6.0
    bf = {k: v for k, v in per.items() if k.endswith(":bf16x3")}
    fp = {k: v for k, v in per.items() if not k.endswith(":bf16x3")}
    steps = max((v["launches"] / v["launches_per_step"] for v in per.values()), default=1)
    flop_bf = 2.0 * sum(v["macs"] for v in bf.values()) / steps
    flop_fp = 2.0 * sum(v["macs"] for v in fp.values()) / steps
    floor_ms = 1e3 * (flop_bf / (peak_b3 * 1e12) + flop_fp / (PEAK_FP32_TFLOPS * 1e12))
    out = {"bound": "mfma", "unit": "TFLOP/s (fp32-equivalent)", "peak_bf16x3": peak_b3, "peak_fp32": PEAK_FP32_TFLOPS,
           "tflop_per_step_bf16_pipe_fp32_equivalent": flop_bf * 1e-12, "tflop_per_step_fp32_pipe": flop_fp * 1e-12,
           "blended_floor_ms": floor_ms, "ms_per_step": ms_per_step, "frac_of_blended_roofline": floor_ms / ms_per_step}
    if bf:
        name, dom = max(bf.items(), key=lambda kv: kv[1]["ms"])
        out.update({"kernel": name, "achieved": dom["tflops"], "peak": peak_b3, "frac": dom["tflops"] / peak_b3,
                    "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                    "bf16x3_kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "avg_us": round(v["avg_us"], 2),
                                           "tflops_fp32_equivalent": round(v["tflops"], 2),
                                           "frac_of_bf16x3_peak": round(v["tflops"] / peak_b3, 4)} for k, v in sorted(bf.items())}})
    return out


def make_inputs(batch, rank):
    gen = torch.Generator().manual_seed(1234 + rank)
    return (0.1 * torch.randn(batch, 1, CLIP, generator=gen)).clamp(-1, 1)


def build_model(dev):
    from audio_generation_amd.vae import CausalVQAE
    torch.manual_seed(0)
    model = CausalVQAE(**MODEL_KW).eval()
    return model.to(dev)


def calibrate_codebooks(model, x_small, recipe, n_clips=None):
    """Synthetic codebooks.  ``survey``: the letter of 8(d) -- ``randn(Q,K,D) * sigma``, seed 7, sigma =
    std of the encoder output measured in the same run.  ``latents``: stage 0 = latent frames of the first clips +
    noise, later stages shrinking randn (what a k-means-initialised quantiser looks like: the arg-min is spread
    over hundreds of codes instead of the few nearest to the latents' common offset)."""
    with torch.no_grad():
        # n_clips: x_small is the WHOLE batch and the first n_clips clips' latents are used -- bit-identical to encoding those clips
        # alone (batch items are independent, tests/test_gpu_fullsize.py) but every launch of the run then has the measured batch
        # size, so a rocprofv3 --stats average of a kernel is not polluted by small calibration launches
        z = model._run_encoders(x_small)
        if n_clips is not None:
            z = z[:n_clips].contiguous()
        if recipe == "survey":
            sigma = float(z.std())
            model.quantizer.init_randn(sigma, seed=7)
            return sigma
        return model.quantizer.init_from_latents(z, seed=7)


def pmc_entry(pmc, dom_name):
    """Entries of profiles/pmc_traffic.json for a kernel the observer names ``stem<a,b,c>[:bf16x3]``.  The PMC names
    carry the full template list.  First-round kernels (``resblock_mfma`` / ``conv_mfma``): ``stem<a,b,c,...,IMPL>`` whose
    LAST argument is the arithmetic (0 = fp32 MFMA, 1 = bf16x3) -- match the stem AND that suffix, never the first prefix
    hit.  Ring kernels (``resblock_p<MW,NW,CCH,D,NS>``: fp32 only): one entry per dilation, all of them are returned (the
    observer aggregates the three dilations under one name, so their traffic is averaged)."""
    bf = dom_name.endswith(":bf16x3")
    stem = dom_name.replace("2x:", "").replace(":bf16x3", "").rstrip(">")
    want = "1" if bf else "0"
    ring = stem.startswith(("resblock_p", "conv_p"))
    hits = []
    for k in pmc:
        if k == stem + ">":
            return [k]
        if k.startswith(stem + ",") and (ring or k.rstrip(">").rsplit(",", 1)[1] == want):
            hits.append(k)
    return hits if ring else hits[:1]


def parity_block(model, sd, spec, y_gpu, idx_gpu, z_gpu, cpu_run):
    """GPU vs oracle on the bounded sample (same inputs / weights / codebooks):
      (a) oracle RVQ on the GPU's latents must give the GPU's indices exactly;
      (b) waveform RMS vs the oracle decode of those codes;
      (c) the fully independent CPU path -- agreement, and the near-tie PROOF of every first disagreement
          (oracle/neartie.py: both top-2 margins <= 2 |delta| |c_a - c_b|, delta = measured latent difference).
    ``cpu_run`` = (z, idx, y) of the oracle's own forward on the same clips."""
    from oracle import codec, neartie, rvq
    cbs = model.quantizer.codebooks.detach().cpu()
    z, idx, y = cpu_run
    n, t = idx.shape[:2]
    frames_gpu = z_gpu.cpu().transpose(1, 2).contiguous()
    with torch.no_grad():
        zq_g, idx_g, _ = rvq.residual_quantize(frames_gpu, cbs)
        y_g = codec.decode_latents(zq_g, sd, spec)
    rep = neartie.explain_disagreements(frames_gpu.reshape(n * t, -1).numpy(), z.reshape(n * t, -1).numpy(),
                                        idx_gpu.cpu().reshape(n * t, -1).numpy(), idx.reshape(n * t, -1).numpy(), cbs.numpy())
    return {
        "distinct_stage0_codes_in_sample": int(idx_gpu[..., 0].unique().numel()),
        "distinct_codes_per_stage_in_sample": [int(idx_gpu[..., q].unique().numel()) for q in range(idx_gpu.shape[-1])],
        "clips": int(n), "frames": int(n * t),
        "index_bit_exact_on_same_latents": bool(torch.equal(idx_g, idx_gpu.cpu())),
        "waveform_rms_vs_oracle": float((y_gpu.cpu().double() - y_g.double()).pow(2).mean().sqrt()),
        "independent_path_index_agreement": float((idx == idx_gpu.cpu()).float().mean()),
        "independent_path_waveform_rms": float((y_gpu.cpu().double() - y.double()).pow(2).mean().sqrt()),
        "latent_rms_vs_oracle": float((frames_gpu.double() - z.double()).pow(2).mean().sqrt()),
        "independent_path_disagreements_proved_near_ties": rep["proved"],
        "near_tie_report": {k: rep[k] for k in ("frames_with_a_disagreement", "codes_disagreeing", "max_margin_over_bound",
                                                "max_relative_margin", "max_latent_error_relative", "unexplained",
                                                "negative_margins")},
    }


def cpu_baseline_and_parity(model, x_cpu, y_gpu, idx_gpu, z_gpu, n_items):
    """Time the oracle on host cores on the first ``n_items`` clips and check the GPU result."""
    from oracle import codec, rvq
    # host cores this process may use; the 1-GPU box's CPU share is 16 (a 256-thread pool on a
    # 16-core share only adds contention), so the pool is capped there
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    spec = codec.CodecSpec(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                           wavelet_decoders=False, input_format="n c l")
    cbs = sd["quantizer.codebooks"]
    xs = x_cpu[:n_items]

    def run(xin):
        """One CPU forward; returns the results and the (encoder, RVQ, decoder) seconds."""
        with torch.no_grad():
            t0 = time.perf_counter()
            z = codec.encode_latents(xin, sd, spec)
            t1 = time.perf_counter()
            zq, idx, _ = rvq.residual_quantize(z, cbs, score_dtype=torch.float32)
            t2 = time.perf_counter()
            y = codec.decode_latents(zq, sd, spec)
            t3 = time.perf_counter()
        return (z, zq, idx, y), (t1 - t0, t2 - t1, t3 - t2)

    # 1 warm-up + 3 timed forwards, median; all host cores of this process's share.  The share a box
    # really grants can be smaller than the affinity mask says (a CPU quota: 16 threads then run no faster than one), so
    # the pool size is picked by a short probe -- one clip through the encoder at 16 / 8 / 4 / 2 threads, fastest wins.
    probe = {}
    for nt in sorted({cores, max(cores // 2, 1), max(cores // 4, 1), max(cores // 8, 1)}, reverse=True):
        torch.set_num_threads(nt)
        with torch.no_grad():
            codec.encode_latents(xs[:1], sd, spec)
            t0 = time.perf_counter()
            codec.encode_latents(xs[:1], sd, spec)
            probe[nt] = time.perf_counter() - t0
    avail, cores = cores, min(probe, key=probe.get)
    torch.set_num_threads(cores)
    run(xs)  # warm-up (thread pools, page faults)
    runs = [run(xs) for _ in range(3)]
    order = sorted(range(3), key=lambda i: sum(runs[i][1]))
    (z, zq, idx, y), stage = runs[order[1]]
    sec = sum(stage)
    # The same clips ONE AT A TIME (all threads on one clip: its activations stay in cache where the 8-clip batch falls out of
    # it -- VERDICT r3: the encoder ran 4.6 x faster per clip that way).  1 warm-up + 3 timed passes over the sample, median.
    def run_clipwise():
        t = [0.0, 0.0, 0.0]
        for i in range(n_items):
            _, st = run(xs[i:i + 1])
            t = [a + b for a, b in zip(t, st)]
        return t
    run(xs[:1])
    clip_runs = sorted((run_clipwise() for _ in range(3)), key=sum)
    stage_c = clip_runs[1]
    sec_c = sum(stage_c)
    # ... and the per-core figure: the same forward on ONE clip with one thread (1 warm-up + 1 timed)
    torch.set_num_threads(1)
    run(xs[:1])
    _, stage1 = run(xs[:1])
    torch.set_num_threads(cores)
    parity = parity_block(model, sd, spec, y_gpu[:n_items], idx_gpu[:n_items], z_gpu[:n_items], (z, idx, y))
    nsmp = n_items * CLIP
    batched, clipwise = nsmp / sec, nsmp / sec_c
    mode = "clip by clip" if clipwise > batched else f"one batch of {n_items}"
    best_stage = stage_c if clipwise > batched else stage
    base = {"value": max(batched, clipwise), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_items} clips x {CLIP} samples (same inputs/weights as the GPU run), "
                      f"oracle conv stacks (torch fp32 CPU, {cores} threads) + exact RVQ, median of 3 after 1 warm-up; "
                      f"timed both as one batch and clip by clip, value = the faster ({mode})",
            "value_is": mode, "batched_samples_per_s": batched, "clip_by_clip_samples_per_s": clipwise,
            "cores_available": avail, "thread_probe_s": {str(k): round(v, 4) for k, v in probe.items()},
            "stages": {"encoder": nsmp / best_stage[0], "rvq": nsmp / best_stage[1], "decoder": nsmp / best_stage[2]},
            "stages_batched": {"encoder": nsmp / stage[0], "rvq": nsmp / stage[1], "decoder": nsmp / stage[2]},
            "one_thread": {"value": CLIP / sum(stage1), "sample": f"1 clip x {CLIP} samples, 1 thread, 1 timed after 1 warm-up"}}
    return base, parity, (sd, spec, z)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=BATCH_PER_GPU)
    ap.add_argument("--cpu-items", type=int, default=8, help="clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the launches eagerly instead of replaying a hipGraph")
    ap.add_argument("--arith", choices=("fp32", "mixed"), default="fp32",
                    help="arithmetic of the MEASURED configuration: fp32 = fp32-input MFMA everywhere (bitwise an fp32 FMA "
                         "chain); mixed = encoder + RVQ as fp32, decoder on the bf16x3 kernels (fp32-class accuracy)")
    ap.add_argument("--codebooks", choices=("survey", "latents"), default="latents",
                    help="synthetic codebook recipe of the measured run (see calibrate_codebooks); the other one is "
                         "measured as well and reported under other_codebooks")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurement of the other arithmetic")
    ap.add_argument("--no-train-step", action="store_true",
                    help="skip the config-5 training-step measurement (a child process, outside the timed region)")
    args = ap.parse_args()

    from audio_generation_amd import dist as agx_dist
    from audio_generation_amd import ops

    rank, local_rank, world = agx_dist.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 through `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU path)"
    # one GPU per rank.  AGX_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks (ranks then share devices and the barrier / MAX-reduce run on CPU tensors).
    backend = os.environ.get("AGX_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks but {n_dev} GPUs visible")
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    agx_dist.init(backend)
    red_dev = dev if backend == "nccl" else "cpu"

    bsz = args.batch_per_gpu
    model = build_model(dev)
    x_cpu = make_inputs(bsz, rank)
    x = x_cpu.to(dev)                       # inputs resident in HBM before the timed region
    sigma = calibrate_codebooks(model, x, args.codebooks, n_clips=8)
    if world > 1:
        # weights + codebooks replicated.  Every rank calibrated on its own shard (inputs are seeded per rank):
        # rank 0's codebooks win, so all ranks search the same codebooks (the RVQ's candidate count is data dependent).
        model.quantizer.sync_from_rank0()
        sigma = agx_dist.gather_floats(float(sigma), device=red_dev)[0]
    if args.arith == "mixed":
        model.set_conv_arithmetic(decoders="bf16x3")

    def eager_step():
        with torch.no_grad():
            return model(x)

    if args.no_graph:
        step = eager_step
    else:
        from audio_generation_amd.graph import GraphedForward
        graphed = GraphedForward(model, x)   # static input = x, already resident in HBM
        step = graphed.replay

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    agx_dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    agx_dist.barrier()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    elapsed = agx_dist.max_over_ranks(mine, device=red_dev)
    per_rank_ms = agx_dist.gather_floats(1e3 * mine / args.steps, device=red_dev)   # a straggler shows up here
    y, commit, index = out
    distinct0 = int(index[..., 0].unique().numel())

    ms_per_step = 1e3 * elapsed / args.steps
    total_samples = world * bsz * CLIP * args.steps
    result = {
        "metric": "24kHz samples/s encode->RVQ->decode", "value": total_samples / elapsed, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
        "config": {"workload": "Soundstream default: 8-codebook RVQ (1024 x 512), strides [2,4,5,8], 24 kHz, "
                               f"batch {bsz}/GPU x {CLIP} samples, fp32, eval forward (encode->RVQ->decode)",
                   "batch_per_gpu": bsz, "global_batch": bsz * world, "clip_samples": CLIP,
                   "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "launch": "eager" if args.no_graph else "hipGraph replay",
                   "codebooks": ("SURVEY 8(d): randn(Q,K,D) * sigma, seed 7" if args.codebooks == "survey" else
                                 "stage 0 = latent frames + 0.1 sigma noise, stage q = randn * sigma * 0.6^q, seed 7"),
                   "codebook_sigma": sigma, "distinct_stage0_codes": distinct0,
                   "arithmetic": ("fp32-input MFMA in every conv (bitwise an fp32 FMA chain), RVQ fp32 scores + exact fp64 re-check"
                                  if args.arith == "fp32" else
                                  "encoder + RVQ as in fp32 mode; decoder convs on the bf16x3 kernels (operands split into three "
                                  "bf16 pieces, six bf16 MFMAs per product block, fp32 accumulation: same error vs fp64 as fp32)")},
    }

    if rank == 0 and not args.no_roofline:
        # instrumented pass (outside the timed region): per-launch HIP-event durations
        timer = LaunchTimer()
        ops.set_observer(timer)
        prof_steps = max(2, min(5, args.steps))
        for _ in range(prof_steps):
            eager_step()
        ops.set_observer(None)
        per = timer.summary(prof_steps)
        dom_name, dom = max(per.items(), key=lambda kv: kv[1]["ms"])
        mfma_bound = dom_name.startswith(("conv_mfma", "rvq", "resblock", "2x:conv_mfma"))
        exec_flops = 2.0 * sum(r["macs"] for r in per.values()) / prof_steps
        ref_flops = 2.0 * sum(r["ref_macs"] for r in per.values()) / prof_steps
        enc_bytes = 4864.0  # layer-boundary bytes per input sample, encoder
        if mfma_bound and dom_name.endswith(":bf16x3"):
            roof = {"bound": "mfma", "achieved": 6 * dom["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6 * dom["tflops"] / PEAK_BF16_TFLOPS, "traffic": None,
                    "note": "bf16x3 kernel: executed bf16 MFMA flops = 6 x the fp32-equivalent count"}
        elif mfma_bound:
            roof = {"bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": dom["tflops"] / PEAK_FP32_TFLOPS, "traffic": None}
        else:
            roof = {"bound": "hbm", "achieved": dom["gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": dom["gbps"] / PEAK_HBM_GBPS, "traffic": None}
        # HBM bytes per launch of that kernel from the committed PMC passes (tools/pmc_summary.py)
        try:
            from audio_generation_amd.build import source_hash
            pmc_file = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pmc = pmc_file["kernels"]
            keys = pmc_entry(pmc, dom_name)
            if keys:
                roof["traffic"] = sum(pmc[k]["bytes_per_launch"] for k in keys) / len(keys)
                roof["traffic_kernel"] = keys
                roof["algorithmic_bytes_per_launch"] = dom["bytes"] / dom["launches"]
                roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
                here, there = source_hash(), pmc_file.get("kernel_sources_sha16")
                roof["traffic_source"] = ("profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, "
                                          "per launch); " +
                                          ("measured on THIS build (kernel sources sha16 " + here + ")" if here == there else
                                           f"measured on the build with kernel sources sha16 {there or 'unrecorded (round 3, commit eadd679)'}"
                                           f"; this run's sources are {here} -- the counters are from that earlier build of the "
                                           "same kernel, not from this run"))
        except (OSError, ValueError, KeyError):
            pass
        roof.update({"kernel": dom_name, "avg_launch_us": dom["avg_us"],
                     "launches_per_step": dom["launches_per_step"],
                     "share_of_step": dom["ms_per_step"] / ms_per_step,
                     "flops_counted": "executed (polyphase) MACs x 2 (fp32-equivalent for bf16x3 kernels); reference-counted total alongside",
                     "whole_forward_tflops_executed": 1e-9 * exec_flops / ms_per_step,
                     "whole_forward_tflops_reference_count": 1e-9 * ref_flops / ms_per_step,
                     "whole_forward_frac_of_fp32_peak": 1e-9 * exec_flops / ms_per_step / PEAK_FP32_TFLOPS,
                     "encoder": encoder_roofline(model, x, bsz, enc_bytes),
                     "kernels_note": "per-kernel times come from an EAGER instrumented pass (HIP events around every C-ABI call, "
                                     "outside the timed region); their sum exceeds ms_per_step of the hipGraph replay by the "
                                     "inter-launch gaps the graph removes",
                     "kernels_sum_ms_per_step": round(sum(v["ms_per_step"] for v in per.values()), 4),
                     "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "avg_us": round(v["avg_us"], 2),
                                     "launches_per_step": v["launches_per_step"], "tflops": round(v["tflops"], 2),
                                     "gbps": round(v["gbps"], 1)} for k, v in sorted(per.items())}})
        result["roofline"] = roof
        if args.arith == "mixed":       # the measured configuration mixes the two matrix pipes: its own ceilings
            result["roofline_mixed"] = arithmetic_roofline(per, ms_per_step)

    if rank == 0 and world == 1 and args.cpu_items > 0:
        with torch.no_grad():
            z_gpu = model._run_encoders(x)
        n_cpu = min(args.cpu_items, bsz)
        base, parity, (sd_cpu, spec_cpu, z_cpu) = cpu_baseline_and_parity(model, x_cpu, y, index, z_gpu, n_cpu)
        parity["codebooks"] = args.codebooks
        result["cpu_baseline"] = base
        result["parity"] = parity
        result["speedup_vs_cpu_baseline"] = result["value"] / base["value"]
        # the same parity block under the OTHER synthetic-codebook recipe (the encoder, hence z, does not depend on it)
        from oracle import codec as _codec, rvq as _rvq
        other = "latents" if args.codebooks == "survey" else "survey"
        keep = {k: v.clone() for k, v in model.quantizer.state_dict().items()}
        calibrate_codebooks(model, x, other, n_clips=8)
        with torch.no_grad():
            y_o, _, idx_o = model(x[:n_cpu])
            cbs_o = model.quantizer.codebooks.detach().cpu()
            zq_c, idx_c, _ = _rvq.residual_quantize(z_cpu, cbs_o)
            y_c = _codec.decode_latents(zq_c, sd_cpu, spec_cpu)
        result["parity_other_codebooks"] = dict(codebooks=other, **parity_block(model, sd_cpu, spec_cpu, y_o, idx_o,
                                                                                z_gpu[:n_cpu], (z_cpu, idx_c, y_c)))
        model.quantizer.load_state_dict(keep)

    if rank == 0 and world == 1 and not args.no_extra:
        # secondary measurements, outside the timed region and NOT the headline value: the other arithmetics
        def measure(dec, enc):
            """Same protocol as the headline (hipGraph replay unless --no-graph), plus the eager figure."""
            model.set_conv_arithmetic(decoders=dec, encoders=enc)
            n2 = max(5, min(args.steps, 20))
            with torch.no_grad():
                for _ in range(3):
                    y2, _, index2 = model(x)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n2):
                    y2, _, index2 = model(x)
                torch.cuda.synchronize()
                dt_eager = (time.perf_counter() - t1) / n2
            dt, launch = dt_eager, "eager"
            if not args.no_graph:
                from audio_generation_amd.graph import GraphedForward
                g2 = GraphedForward(model, x)
                for _ in range(3):
                    g2.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n2):
                    y2, _, index2 = g2.replay()
                torch.cuda.synchronize()
                dt, launch = (time.perf_counter() - t1) / n2, "hipGraph replay"
            t2 = LaunchTimer()            # instrumented pass under THIS arithmetic: the per-pipe roofline of the configuration
            ops.set_observer(t2)
            with torch.no_grad():
                for _ in range(2):
                    model(x)
            ops.set_observer(None)
            return {"value": bsz * CLIP / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt, "launch": launch,
                    "ms_per_step_eager": 1e3 * dt_eager, "roofline_mixed": arithmetic_roofline(t2.summary(2), 1e3 * dt),
                    "index_agreement_with_measured_run": float((index2 == index).float().mean()),
                    "waveform_rms_vs_measured_run": float((y2 - y).double().pow(2).mean().sqrt())}

        others = []
        if args.arith == "fp32":
            others.append(dict(arithmetic="mixed: encoder + RVQ fp32, decoder bf16x3 (fp32-class accuracy, not the bitwise "
                                          "fp32 chain; the supported opt-in)", **measure("bf16x3", "fp32")))
        else:
            others.append(dict(arithmetic="fp32: fp32-input MFMA everywhere", **measure("fp32", "fp32")))
        allbf = dict(arithmetic="all bf16x3 (encoder too): shown for the trade-off only -- near-tie indices can flip, so "
                                "it is NOT a supported configuration", **measure("bf16x3", "bf16x3"))
        with torch.no_grad():      # its flips against the fp32 run, put through the near-tie proof (whole batch)
            from oracle import neartie as _neartie
            z_bf, idx_bf = model._run_encoders(x), model(x)[2]
            model.set_conv_arithmetic(decoders="fp32", encoders="fp32")
            z_fp = model._run_encoders(x)
            rep = _neartie.explain_disagreements(z_bf.cpu().transpose(1, 2).reshape(-1, z_bf.shape[1]).numpy(),
                                                 z_fp.cpu().transpose(1, 2).reshape(-1, z_fp.shape[1]).numpy(),
                                                 idx_bf.cpu().reshape(-1, idx_bf.shape[-1]).numpy(),
                                                 index.cpu().reshape(-1, index.shape[-1]).numpy(),
                                                 model.quantizer.codebooks.detach().cpu().numpy())
            enc = encoder_roofline_bf16x3(model, x, bsz)
        allbf["flips_vs_fp32_run_proved_near_ties"] = rep["proved"]
        allbf["near_tie_report"] = {k: rep[k] for k in ("frames", "frames_with_a_disagreement", "codes_disagreeing",
                                                        "max_margin_over_bound", "max_relative_margin",
                                                        "max_latent_error_relative", "unexplained", "negative_margins")}
        allbf["encoder"] = enc
        others.append(allbf)
        model.set_conv_arithmetic(decoders="bf16x3" if args.arith == "mixed" else "fp32")
        result["other_arithmetic"] = others
        # the other synthetic-codebook recipe, same arithmetic as the measured run (eager)
        other = "latents" if args.codebooks == "survey" else "survey"
        keep = {k: v.clone() for k, v in model.quantizer.state_dict().items()}
        calibrate_codebooks(model, x, other, n_clips=8)
        r = measure("bf16x3" if args.arith == "mixed" else "fp32", "fp32")
        r.pop("index_agreement_with_measured_run"), r.pop("waveform_rms_vs_measured_run")
        with torch.no_grad():
            r["distinct_stage0_codes"] = int(model(x)[2][..., 0].unique().numel())
        result["other_codebooks"] = dict(recipe=other, **r)
        model.quantizer.load_state_dict(keep)

    if rank == 0 and world == 1 and not args.no_train_step:
        # BASELINE configs[4] at its per-GPU batch (32 clips): one whole training step -- generator forward + native backward,
        # six discriminators x three passes, low-pass / pre-emphasis / 7-window mel terms, seven Adam steps -- measured by
        # tools/train_step_bench.py in a CHILD process after everything above (never inside the timed region); its JSON line
        # carries ms/step, the executed FLOPs summed per launch, the fraction of the fp32 MFMA peak and the peak memory.
        import subprocess

        def child(extra_env):
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_step_bench.py"), str(bsz), "2"],
                                     env=dict(os.environ, AGX_GAN="1", **extra_env), capture_output=True, text=True, timeout=600)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                return json.loads(line[-1]) if line else {"error": (out.stderr or out.stdout)[-400:]}
            except Exception as exc:    # the headline line must not depend on the secondary measurement
                return {"error": repr(exc)[:400]}

        result["training_step"] = child({})
        # the same step with the opt-in bf16x3 arithmetic (fp32-class accuracy) on the decoder and on the
        # discriminators' Conv2d forward, backward-data AND weight-gradient contraction (knob dw2_bf = 1); its fraction is taken
        # against the blended bf16x3 / fp32 floor (tools/train_step_bench.py), never against the fp32 peak
        result["training_step_bf16x3"] = child({"AGX_BF16X3": "1"})

    if rank == 0:
        print(json.dumps(result), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
