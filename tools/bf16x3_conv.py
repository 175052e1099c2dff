#!/usr/bin/env python3
"""fp32-MFMA vs bf16x3 on the non-residual convs of config S: time and error against an fp64 CPU conv."""
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import ops  # noqa: E402
from audio_generation_amd._lib import CONV_CAUSAL, CONV_TRANSPOSED, CONV_UPSAMPLE, IMPL_MFMA, IMPL_MFMA_BF16X3  # noqa: E402
from tools.ab_bench import time_fn  # noqa: E402

LAYERS = [(CONV_CAUSAL, 64, 128, 9, 4, 36000), (CONV_CAUSAL, 256, 512, 17, 8, 1800), (CONV_CAUSAL, 512, 512, 3, 1, 225),
          (CONV_TRANSPOSED, 512, 512, 7, 1, 225), (CONV_UPSAMPLE, 512, 256, 17, 8, 225), (CONV_UPSAMPLE, 256, 128, 11, 5, 1800),
          (CONV_UPSAMPLE, 128, 64, 9, 4, 9000), (CONV_UPSAMPLE, 64, 32, 5, 2, 36000)]


def main():
    tot = {IMPL_MFMA: 0.0, IMPL_MFMA_BF16X3: 0.0}
    for kind, ci, co, k, s, length in LAYERS:
        torch.manual_seed(ci + k)
        x = torch.randn(32, ci, length, device="cuda")
        wshape = (ci, co, k) if kind == CONV_TRANSPOSED else (co, ci, k)
        w = torch.randn(*wshape, device="cuda") / (ci * k) ** 0.5
        out = {}
        for impl in (IMPL_MFMA, IMPL_MFMA_BF16X3):
            d = ops.conv_desc(kind, 32, ci, co, length, k, s, 1, 1, 0.1, impl)
            pk = ops.conv_pack(d, w)
            y = ops.conv_forward(d, x, pk, None)
            t = statistics.median(time_fn(lambda: ops.conv_forward(d, x, pk, None)) for _ in range(5))
            out[impl] = (y, t)
            tot[impl] += t
        y32, t32 = out[IMPL_MFMA]
        ybf, tbf = out[IMPL_MFMA_BF16X3]
        err = float((y32 - ybf).abs().max()) / float(y32.abs().max())
        print(f"kind {kind} {ci:3d}->{co:3d} K={k:2d} s={s} L={length:5d}: fp32 {t32:7.1f} us   bf16x3 {tbf:7.1f} us  "
              f"({t32 / tbf:4.2f}x)   max |diff| / max |y| = {err:.2e}", flush=True)
    print(f"total: fp32 {tot[IMPL_MFMA]:.1f} us   bf16x3 {tot[IMPL_MFMA_BF16X3]:.1f} us")


if __name__ == "__main__":
    main()
