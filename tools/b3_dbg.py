#!/usr/bin/env python3
"""Where the time of resblock_b3 goes: the kernel timed with parts left out (knob b3_dbg: 1 input loads + split, 2 GEMM2,
4 residual / epilogue traffic, 8 weight DMA, 16 GEMM1 MFMAs; results are wrong in those runs, timing only).
usage: b3_dbg.py [C ...]"""
import statistics
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import _lib  # noqa: E402
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
    lib = _lib.load()
    want = [int(v) for v in sys.argv[1:]] or [32, 64, 128, 256]
    shapes = {32: 72000, 64: 36000, 128: 9000, 256: 1800}
    masks = [0, 1, 2, 4, 8, 16, 1 | 8, 2 | 4, 1 | 2 | 4 | 8, 31]
    for c in want:
        length = shapes[c]
        for d in (1, 9):
            m = CausalResidualBlock1d(c, c, dilation=d).to("cuda").eval()
            m.conv1.impl = _lib.IMPL_MFMA_BF16X3
            x = torch.randn(32, c, length, device="cuda")
            with torch.no_grad():
                m.run(x, 0.1)
            res = {v: [] for v in masks}
            for _ in range(5):
                for v in masks:
                    lib.agx_set_tuning(b"b3_dbg", v)
                    with torch.no_grad():
                        res[v].append(time_fn(lambda: m.run(x, 0.1)))
            lib.agx_set_tuning(b"b3_dbg", 0)
            flops = 2.0 * 32 * c * c * 8 * length
            base = statistics.median(res[0])
            print(f"C={c} d={d}: full {base:.1f} us ({flops / base * 1e-6:.1f} TF) | " + "  ".join(
                f"-{v}: {statistics.median(r):.1f}" for v, r in res.items() if v), flush=True)


if __name__ == "__main__":
    main()
