#!/usr/bin/env python3
"""The decoder's bf16x3 resampling convs (csrc/conv_b3.hip) fed with fp32 activations (split per tile in registers) against
the same layers fed with activation planes (split once, staged by LDS-DMA), config-S shapes at batch 32; also the standalone
split pass.  HIP events, 20 launches each.   usage: planes_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import _lib, ops  # noqa: E402

DEV = "cuda"
LAYERS = [("k7", _lib.CONV_TRANSPOSED, 512, 512, 7, 1, 225), ("up8", _lib.CONV_UPSAMPLE, 512, 256, 17, 8, 225),
          ("up5", _lib.CONV_UPSAMPLE, 256, 128, 11, 5, 1800), ("up4", _lib.CONV_UPSAMPLE, 128, 64, 9, 4, 9000),
          ("up2", _lib.CONV_UPSAMPLE, 64, 32, 5, 2, 36000)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    b = 32
    for name, kind, cin, cout, k, s, length in LAYERS:
        wshape = (cin, cout, k) if kind == _lib.CONV_TRANSPOSED else (cout, cin, k)
        v = (torch.randn(wshape) / (cin * k) ** 0.5).to(DEV)
        bias = torch.randn(cout).to(DEV)
        x = torch.randn(b, cin, length, device=DEV)
        desc = ops.conv_desc(kind, b, cin, cout, length, k, s, 1, _lib.EPI_LEAKY_PRE, 0.1, _lib.IMPL_MFMA_BF16X3)
        packed = ops.conv_pack(desc, v)
        planes = ops.planes_split(x)
        macs = ops._conv_macs(desc)
        t_f = timed(lambda: ops.conv_forward(desc, x, packed, bias))
        t_p = timed(lambda: ops.conv_forward_planes(desc, planes, packed, bias))
        t_s = timed(lambda: ops.planes_split(x))
        same = torch.equal(ops.conv_forward(desc, x, packed, bias), ops.conv_forward_planes(desc, planes, packed, bias))
        print(f"{name:4s} {cin:3d}->{cout:3d} L={length:5d}: fp32 input {t_f:7.1f} us = {2e-6 * macs / t_f:6.1f} TF   planes input "
              f"{t_p:7.1f} us = {2e-6 * macs / t_p:6.1f} TF   (standalone split {t_s:6.1f} us = "
              f"{1e-3 * 10 * x.numel() / t_s:6.0f} GB/s)   bit-identical: {same}", flush=True)


if __name__ == "__main__":
    main()
