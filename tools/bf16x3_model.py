#!/usr/bin/env python3
"""Config S forward with the decoder (and optionally the encoder) on the bf16x3 kernels: step time, waveform
difference vs the all-fp32 run, index agreement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda")
    model = bench.build_model(dev)
    x = bench.make_inputs(32, 0).to(dev)
    bench.calibrate_codebooks(model, x[:8])
    res = {}
    for name, kw in (("fp32", {}), ("dec bf16x3", {"decoders": "bf16x3"}), ("enc+dec bf16x3", {"decoders": "bf16x3", "encoders": "bf16x3"})):
        model.set_conv_arithmetic(**kw)
        with torch.no_grad():
            for _ in range(3):
                y, _, idx = model(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y, _, idx = model(x)
            e1.record()
            torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 10, y, idx)
    y0, i0 = res["fp32"][1], res["fp32"][2]
    for name, (ms, y, idx) in res.items():
        print(f"{name:16s} {ms:7.3f} ms/step  {32 * 72000 / ms * 1e-3:7.1f} Msamples/s   waveform rms diff vs fp32 "
              f"{float((y - y0).pow(2).mean().sqrt()):.2e}   index agreement {float((idx == i0).float().mean()):.6f}")


main()
