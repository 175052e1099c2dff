#!/usr/bin/env python3
"""Forward time of the reference's training discriminators (training.py:570-576: WaveFormDiscriminator(1) +
STFTDiscriminator(win) for win in 2048..128) on one MI355X, batch 32 x 72000 samples, training mode
(one spectral-norm power iteration per layer per pass), HIP events.  Prints one JSON line per discriminator
and the per-launch breakdown of the heaviest one."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import discriminator as ad  # noqa: E402
from audio_generation_amd import ops  # noqa: E402


def conv2d_flops(d, b, length):
    """2 x MACs of one forward of an STFT discriminator."""
    n, hop = d.n_fft, d.hop_length
    t, f = 1 + length // hop, n
    total = 2.0 * b * t * (2 * n) * n            # the DFT itself
    for m in [d.first_conv] + [c for blk in d.blocks for c in (blk.layers[0], blk.layers[2])] + [d.final_conv]:
        kh, kw = m.kernel_size
        t = (t + 2 * m.padding[0] - kh) // m.stride[0] + 1
        f = (f + 2 * m.padding[1] - kw) // m.stride[1] + 1
        total += 2.0 * b * m.out_channels * m.in_channels * kh * kw * t * f
    return total


def main():
    b, length = int(os.environ.get("B", 32)), 72000
    dev = torch.device("cuda")
    torch.manual_seed(0)
    x = (0.1 * torch.randn(b, 1, length)).clamp(-1, 1).to(dev)
    discs = [ad.WaveFormDiscriminator(1)] + [ad.STFTDiscriminator(win_length=w) for w in (2048, 1024, 512, 256, 128)]
    tot_ms = 0.0
    for d in discs:
        d = d.to(dev).train()
        with torch.no_grad():
            for _ in range(2):
                d(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                d(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot_ms += ms
        row = {"discriminator": d.name, "batch": b, "samples": length, "ms_forward": round(ms, 3),
               "params": sum(p.numel() for p in d.parameters())}
        if isinstance(d, ad.STFTDiscriminator):
            fl = conv2d_flops(d, b, length)
            row["gflop"] = round(fl * 1e-9, 1)
            row["tflops"] = round(fl / ms * 1e-9, 1)
        print(json.dumps(row), flush=True)
        del d
        torch.cuda.empty_cache()
    print(json.dumps({"all_six_forward_ms": round(tot_ms, 2), "note": "x3 passes per discriminator_generator_loss"}))


if __name__ == "__main__":
    main()
