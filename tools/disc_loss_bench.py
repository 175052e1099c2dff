#!/usr/bin/env python3
"""Time of discriminator_generator_loss + both backward passes per discriminator (B x 72000, training mode)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import discriminator as ad  # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda")
    torch.manual_seed(0)
    x = (0.1 * torch.randn(b, 1, 72000)).clamp(-1, 1).to(dev)
    discs = [ad.WaveFormDiscriminator(1)] + [ad.STFTDiscriminator(win_length=w) for w in (2048, 1024, 512, 256, 128)]
    if os.environ.get("AGX_ONLY"):
        discs = [d for d in discs if os.environ["AGX_ONLY"] in d.name]
    tot = 0.0
    for d in discs:
        d = d.to(dev).train()
        y = (x + 0.01 * torch.randn_like(x)).requires_grad_(True)

        def step():
            for p in d.parameters():
                p.grad = None
            y.grad = None
            gl, dl = ad.discriminator_generator_loss(x, y, d)
            dl.backward(retain_graph=True)
            gl.backward()

        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        tot += ms
        print(json.dumps({"discriminator": d.name, "batch": b, "ms_loss_fwd_bwd": round(ms, 2),
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
        del d
        torch.cuda.empty_cache()
    print(json.dumps({"all_six_ms": round(tot, 1)}))


main()
