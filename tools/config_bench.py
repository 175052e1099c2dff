#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2] / configs[3] (the headline line is bench.py):

  C3  Soundstream default + attention bottleneck  Transformer(512, depth 1, 8 heads x 64, context 225)
      in place of the RVQ, batch 32 x 72 000 samples
  C4  wavelet decoder (reference wiring [F,T,F,F]), stereo, batch 8 x 144 000 samples (3 s @ 48 kHz)
  C4m the same + a multiresolution layer in every encoder and decoder block (build-defined placement)

Prints one JSON line per config: throughput (hipGraph replay), per-kernel table from HIP events,
and parity of one clip against the CPU oracle.   usage: config_bench.py [C3|C4|all]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import LaunchTimer  # noqa: E402
from audio_generation_amd import ops  # noqa: E402
from audio_generation_amd.graph import GraphedForward  # noqa: E402
from audio_generation_amd.transformers import Transformer, TransformerBottleneck  # noqa: E402
from audio_generation_amd.vae import CausalVQAE  # noqa: E402
from oracle import attention as oattn, codec  # noqa: E402

DEV = "cuda"


def run(name):
    torch.manual_seed(0)
    bf16 = name == "C3bf16"      # config 3 as BASELINE states it: attention contractions on the bf16 MFMA (fp32 accumulate / softmax)
    if bf16:
        name = "C3"
    if name == "C3":
        kw = dict(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
                  codebook_dim=512, input_format="n c l", wavelet_decoders=False)
        b, c, length = 32, 1, 72000
        model = CausalVQAE(**kw)
        tf = Transformer(512, depth=1, heads=8, head_dim=64, context_x=225)
        for att, _ in tf.layers:
            att.attention_dtype = "bf16" if bf16 else "fp32"
        model.replace_quantizer(TransformerBottleneck(tf))
        spec = codec.CodecSpec(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                               wavelet_decoders=False, input_format="n c l")
    else:
        wd = [False, True, False, False]
        # C4m: the same + the build-defined multires placement in every encoder and decoder block (CausalVQAE docstring)
        mr = dict(multires_encoders=True, multires_decoders=True, multires_kernel_size=2, multires_depth=3) if name == "C4m" else {}
        kw = dict(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
                  codebook_dim=512, input_format="n c l", wavelet_decoders=wd, **mr)
        b, c, length = 8, 2, 144000
        model = CausalVQAE(**kw)
        spec = codec.CodecSpec(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                               wavelet_decoders=wd, input_format="n c l", **mr)
    model = model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(1234)
    x_cpu = (0.1 * torch.randn(b, c, length, generator=gen)).clamp(-1, 1)
    model = model.to(DEV)
    x = x_cpu.to(DEV)
    if name in ("C4", "C4m"):
        with torch.no_grad():
            model.quantizer.init_from_latents(model._run_encoders(x[:4]))
        sd["quantizer.codebooks"] = model.quantizer.codebooks.detach().cpu().clone()

    g = GraphedForward(model, x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        out = g.replay()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps

    timer = LaunchTimer()
    ops.set_observer(timer)
    with torch.no_grad():
        for _ in range(3):
            model(x)
    ops.set_observer(None)
    per = timer.summary(3)

    # parity of clip 0 against the oracle
    with torch.no_grad():
        y1 = model(x[:1])[0].cpu()
    x1 = x_cpu[:1]
    z = codec.encode_latents(x1, sd, spec)
    if name == "C3":
        tsd = {k[len("quantizer.transformer."):]: v for k, v in sd.items() if k.startswith("quantizer.transformer.")}
        want = codec.decode_latents(oattn.transformer(z, tsd, 8), sd, spec)
    else:
        from oracle import rvq
        want = codec.decode_latents(rvq.residual_quantize(z, sd["quantizer.codebooks"])[0], sd, spec)
    rms = float((y1.double() - want.double()).pow(2).mean().sqrt())
    att = {k: v for k, v in per.items() if k.startswith("attention_alibi")}
    extra = {}
    if att:
        k, v = next(iter(att.items()))
        peak = 2500.0 if bf16 else 157.3     # dense bf16 MFMA / fp32 MFMA peak (MI355X_MICROARCH.md), TFLOP/s
        extra["attention"] = {"kernel": k, "avg_us": round(v["avg_us"], 1), "tflops": round(v["tflops"], 2),
                              "frac_of_mfma_peak": v["tflops"] / peak, "peak_tflops": peak,
                              "flops_counted": "QK^T + PV: 4 B H T^2 Dh", "bound_note":
                              "32 x 8 = 256 (batch, head) pairs x 2 query tiles of 225 frames: 0.83 GFLOP in all -- launch / latency bound"}
    print(json.dumps({
        "config": name + ("bf16" if bf16 else ""), "arithmetic": ("attention contractions bf16 MFMA, fp32 accumulate + softmax; convs fp32"
                                                              if bf16 else "fp32 everywhere"), **extra, "batch": b, "channels": c, "clip_samples": length, "ms_per_step": ms,
        "samples_per_s": b * length / ms * 1e3, "launch": "hipGraph replay", "dtype": "f32",
        "waveform_rms_vs_oracle_clip0": rms,
        "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "avg_us": round(v["avg_us"], 1),
                        "launches_per_step": v["launches_per_step"], "tflops": round(v["tflops"], 2),
                        "gbps": round(v["gbps"], 1)} for k, v in sorted(per.items())}}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for n in (("C3", "C3bf16", "C4", "C4m") if which == "all" else (which,)):
        run(n)
