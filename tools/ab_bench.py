#!/usr/bin/env python3
"""In-process A/B timing of libagx kernels on the config-S layer shapes.

Interleaves the variants round-robin (cdna guide rule 24: perf deltas come from
interleaved rounds in ONE process on ONE device) and prints median / min per
variant.  usage: ab_bench.py knob v0 v1 [...]   e.g.  ab_bench.py resblock_res_lds 0 1
"""
import ctypes
import statistics
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import _lib, ops  # noqa: E402
from audio_generation_amd.vae import CausalResidualBlock1d  # noqa: E402


def time_fn(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    knob, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
    lib = _lib.load()
    dev = "cuda"
    shapes = [(32, 72000), (64, 36000), (128, 9000), (256, 1800)]
    for c, length in shapes:
        for d in (1, 9):
            m = CausalResidualBlock1d(c, c, dilation=d).to(dev).eval()
            if __import__("os").environ.get("AGX_BF16X3") == "1":
                m.conv1.impl = m.conv2.impl = _lib.IMPL_MFMA_BF16X3
            x = torch.randn(32, c, length, device=dev)
            with torch.no_grad():
                m.run(x, 0.1)
            res = {v: [] for v in values}
            for _ in range(7):
                for v in values:
                    lib.agx_set_tuning(knob.encode(), v)
                    with torch.no_grad():
                        res[v].append(time_fn(lambda: m.run(x, 0.1)))
            flops = 2.0 * 32 * c * c * 8 * length
            print(f"C={c:3d} d={d} L={length}: " + "  ".join(
                f"{knob}={v}: med {statistics.median(r):7.1f} us min {min(r):7.1f} ({flops / statistics.median(r) * 1e-6:5.1f} TF)"
                for v, r in res.items()), flush=True)
    lib.agx_set_tuning(knob.encode(), values[-1])


if __name__ == "__main__":
    main()
