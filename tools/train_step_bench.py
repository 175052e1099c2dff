#!/usr/bin/env python3
"""Secondary measurement (not the headline metric): one full training step of the codec on the
native kernels -- forward, backward (native_backward.py), gradient all-reduce and an Adam step --
at config S.  One process per GPU (torchrun sets RANK / WORLD_SIZE); the only collective is ONE
flattened all-reduce of the gradients per step (RCCL over xGMI; AGX_DIST_BACKEND=gloo rehearses the
same code on a box with fewer GPUs than ranks).
AGX_GAN=1 adds the reference's six training discriminators (training.py:570-576) and runs the step as
Trainer.mini_epoch does (training.py:363-385): discriminator_generator_loss per discriminator, the
discriminator loss backward, the generator loss backward, one Adam step each -- BASELINE config 5 without the
mel / pre-emphasis terms (torchaudio).  Discriminator forward AND backward run on the HIP kernels
(discriminator.py: _STFTDiscNative / _WaveBlockNative); the all-reduce then covers generator + discriminator grads.
usage: [torchrun --nproc-per-node N] train_step_bench.py [batch_per_gpu] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import dist as agx_dist  # noqa: E402
from audio_generation_amd import ops  # noqa: E402
from audio_generation_amd.vae import CausalVQAE  # noqa: E402


class FlopCounter:
    """ops observer that only counts: executed MACs of every compute launch of a step (no events, no sync)."""

    def __init__(self):
        self.total, self.total_bf, self.by_kind = 0, 0, {}

    def _add(self, kind, n):
        self.total += n
        if kind.endswith(":bf16x3"):      # a layer whose descriptor asks for the bf16x3 arithmetic: bf16 matrix pipe
            self.total_bf += n
        self.by_kind[kind] = self.by_kind.get(kind, 0) + n

    def begin(self, kind, info):
        import bench
        if kind == "rvq":
            b, t, d, k, q = info
            self._add("rvq", b * t * q * k * d)
        elif kind == "other":
            self._add(info[0], info[2] if len(info) > 2 else 0)
        elif kind == "resblock":
            e1, _, _ = bench.conv_work(info)
            e2, _, _ = bench.conv_work(ops.conv_desc(info.kind, info.batch, info.c_out, info.c_out, info.l_in, 1))
            self._add("conv_forward" + (":bf16x3" if info.impl == 3 else ""), e1 + e2)
        else:
            self._add("conv_forward" + (":bf16x3" if info.impl == 3 else ""), ops._conv_macs(info))
        return None

    def end(self, tok):
        pass

    def macs(self, kind, n):
        self._add(kind, n)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rank, local_rank, world = agx_dist.env_world()
    backend = os.environ.get("AGX_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    agx_dist.init(backend)
    torch.manual_seed(0)                      # identical initial weights on every rank
    model = CausalVQAE(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
                       codebook_dim=512, input_format="n c l", wavelet_decoders=False).to(dev).train()
    gen = torch.Generator().manual_seed(1234 + rank)          # every rank its own shard of the batch
    x = (0.1 * torch.randn(batch, 1, 72000, generator=gen)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        model.quantizer.init_from_latents(model._run_encoders(x[:4]))
    model.quantizer.sync_from_rank0()         # every rank fed its own shard above: rank 0's codebooks win
    update_cb = os.environ.get("AGX_UPDATE_CODEBOOK", "1") == "1"     # training.py:305-308, 326
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    bf_mode = os.environ.get("AGX_BF16X3", "0")       # 1: decoder + every discriminator Conv2d layer with a bf16x3 form; 2: decoder +
    bf = bf_mode in ("1", "2")                        #    only the discriminator layers that have the bf16x3 RING kernel (3 x 3, stride 1)
    if bf:
        model.set_conv_arithmetic(decoders="bf16x3")
    gan = os.environ.get("AGX_GAN", "0") == "1"
    discs, opt_d = [], []
    if gan:
        from audio_generation_amd.discriminator import (STFTDiscriminator, WaveFormDiscriminator,
                                                        discriminator_generator_loss)
        wins = [int(w) for w in os.environ.get("AGX_GAN_WINS", "2048,1024,512,256,128").split(",") if w]
        discs = [WaveFormDiscriminator(1)] + [STFTDiscriminator(win_length=w) for w in wins]
        discs = [d.to(dev).train() for d in discs]
        if bf:
            from audio_generation_amd.discriminator import set_arithmetic
            for d in discs:
                set_arithmetic(d, "bf16x3" if bf_mode == "1" else "bf16x3_ring")
        opt_d = [torch.optim.Adam(d.parameters(), lr=8e-4) for d in discs]
    # the reconstruction-side terms of Trainer.mini_epoch (training.py:313-359): low-pass of the input batch,
    # pre-emphasis before the MSE, the 7-window mel loss  (AGX_SIGNAL=0 switches them off)
    signal = gan and os.environ.get("AGX_SIGNAL", "1") == "1"
    if signal:
        from audio_generation_amd import signal_ops as sg
        windows = [2 ** i for i in range(5, 12)]
        specs = [sg.MelSpectrogram(24000, max(w, 512), w, w // 4, 64, True).to(dev) for w in windows]

    def all_params():
        return list(model.parameters()) + [p for d in discs for p in d.parameters()]

    # the exchange step: ONE flat gradient buffer, allocated here, every .grad a view into it
    bucket = agx_dist.GradBucket(all_params())

    two_calls = os.environ.get("AGX_TWO_CALLS", "0") == "1"   # A/B: the reference's literal backward order

    def step():
        bucket.zero_()
        if signal and not two_calls:          # step.training_backward: one traversal per discriminator, graphs freed one by one
            from audio_generation_amd.step import training_backward
            loss, _, _ = training_backward(model, x, discs, sample_rate=24000, frequency_filter=5000.0, pre_emphasis=0.97,
                                           spectrograms=specs, spec_windows=windows, spec_loss_weight=0.01,
                                           update_codebook=update_cb)
        else:
            if signal:
                xin = sg.lowpass_biquad(x, 24000, 5000.0)
                y, commit, _ = model(xin, update_codebook=update_cb)
                loss = ((sg.preemphasis(y, 0.97) - sg.preemphasis(xin, 0.97)) ** 2).mean() + commit
                loss = loss + sg.multispectral_reconstruction_loss(xin, y, specs, windows, spec_loss_weight=0.01)
            else:
                xin = x
                y, commit, _ = model(xin, update_codebook=update_cb)
                loss = ((y - xin) ** 2).mean() + commit
            if gan:                                 # training.py:363-376
                d_loss = 0
                for d in discs:
                    g_loss, d_loss_i = discriminator_generator_loss(xin, y, d)
                    loss = loss + g_loss
                    d_loss = d_loss + d_loss_i
                d_loss.backward(retain_graph=True)
            loss.backward()
        bucket.allreduce_mean_()               # in place on the flat buffer (RCCL; staged through the host on gloo)
        opt.step()
        for o in opt_d:
            o.step()
        return float(loss.detach())

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ab = {}
    knob = os.environ.get("AGX_AB_KNOB")          # in-process A/B of one tuning knob (devices differ by several per cent)
    if knob:
        from audio_generation_amd import _lib
        for value in (0, 1, 0, 1):
            _lib.load().agx_set_tuning(knob.encode(), value)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ab.setdefault(f"{knob}={value}", []).append(1e3 * (time.perf_counter() - t0) / steps)
    t0 = time.perf_counter()
    losses = [step() for _ in range(steps)]
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    # executed FLOPs of one step, summed per launch by the ops observer (forward convs / blocks / RVQ / Conv2d / STFT + every
    # backward-data and weight-gradient launch; elementwise, reduction and optimizer work is not counted)
    counter = FlopCounter()
    ops.set_observer(counter)
    step()
    ops.set_observer(None)
    torch.cuda.synchronize()
    step_flop = 2.0 * counter.total
    ms = agx_dist.max_over_ranks(ms, device=dev if backend == "nccl" else "cpu")
    # replicas must stay identical: min and max over ranks of a checksum of parameters AND buffers (EMA codebooks)
    lo, hi = zip(*[agx_dist.replica_checksums(m) for m in [model] + discs])
    same = all(a == b for a, b in zip(lo, hi))
    if rank == 0:
        # Rooflines.  fp32 launches: the fp32-input MFMA peak (157.3 TFLOP/s).  Launches of bf16x3 descriptors: the dense bf16
        # MFMA peak at six bf16 flops per fp32-equivalent flop = 2500 / 6 = 416.7 fp32-equivalent TFLOP/s (classified by the
        # layer's descriptor; the few layers of such a descriptor that fall back to an fp32 kernel are counted on the bf16
        # pipe too -- that only lowers the fraction).  A step that mixes both is held against the BLENDED floor
        #   t_floor = flop_bf16x3 / 416.7 T + flop_fp32 / 157.3 T.
        bf_flop = 2.0 * counter.total_bf
        fp_flop = step_flop - bf_flop
        floor_ms = 1e3 * (bf_flop / (2500e12 / 6) + fp_flop / 157.3e12)
        roof = {"executed_tflop_fp32_pipe": fp_flop * 1e-12, "executed_tflop_bf16_pipe_fp32_equivalent": bf_flop * 1e-12,
                "blended_floor_ms": floor_ms, "frac_of_blended_roofline": floor_ms / ms,
                "peaks_tflops": {"fp32_mfma": 157.3, "bf16x3_fp32_equivalent": 2500.0 / 6}}
        if not bf:
            roof["frac_of_fp32_mfma_peak"] = step_flop / ms * 1e-9 / 157.3
        print(json.dumps({"what": "train step (fwd + native bwd + grad all-reduce + Adam), config S" +
                          (f" + {len(discs)} discriminators (native forward + backward)" if gan else "") +
                          (" + low-pass, pre-emphasis, 7-window mel loss" if signal else "") +
                          (" [bf16x3: decoder forward; discriminator Conv2d forward, backward-data AND the weight-gradient "
                           "contraction (knob dw2_bf = 1)" +
                           (", 3 x 3 stride-1 ring layers only]" if bf_mode == "2" else "]") if bf else ""),
                          "backward_order": ("two calls (training.py:374, 380)" if (two_calls or not signal) else "step.training_backward"),
                          "n_gpus": world, "batch_per_gpu": batch, "ms_per_step": ms, "replicas_in_sync": same, "update_codebook": update_cb,
                          "samples_per_s": world * batch * 72000 / ms * 1e3, "losses": losses,
                          "executed_tflop_per_step_per_gpu": step_flop * 1e-12,
                          "executed_tflops": world * step_flop / ms * 1e-9, **roof,
                          "flop_groups_tflop": {k: round(2e-12 * v, 3) for k, v in sorted(counter.by_kind.items())},
                          "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, **({"ab_ms_per_step": ab} if ab else {})}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
