#!/usr/bin/env python3
"""Randomised cross-check of the bf16x3 Conv2d ring kernel (conv_b3.hip: conv2d_b3_kernel) and of the weight-gradient paths added
in round 3 against the fp32 kernels: forward (bias + LeakyReLU), backward-data (plain / masked + gradient add) and weight gradient
on random layer shapes -- 3 x 3 stride 1, (3,4)/(1,2), (4,4)/(2,2), feature maps from 4 to 700 columns, 1 to 60 rows.  Every case
is computed twice on the GPU (bf16x3 descriptor vs fp32 descriptor); the bf16x3 arithmetic is fp32-class, so the two must agree to
3e-5 of the result's largest magnitude.  Prints the worst relative difference per kernel name; exits non-zero above the bound.
usage: c2b3_stress.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import _lib, ops  # noqa: E402
from audio_generation_amd._lib import EPI_LEAKY_PRE, IMPL_MFMA_BF16X3  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = "cuda"
    worst, count = {}, {}

    def note(name, a, r):
        err = float((a - r).abs().max()) / max(1e-6, float(r.abs().max()))
        if not (err == err):
            err = float("inf")
        worst[name] = max(worst.get(name, 0.0), err)
        count[name] = count.get(name, 0) + 1

    for case in range(n):
        kh, kw, sh, sw = rnd.choice([(3, 3, 1, 1), (3, 3, 1, 1), (3, 4, 1, 2), (4, 4, 2, 2)])
        cin = rnd.choice([32, 64, 128, 256])
        cout = rnd.choice([32, 64, 128, 256, 512])
        h = rnd.choice([rnd.randint(1, 12), rnd.randint(12, 60)])
        w = rnd.choice([rnd.choice([4, 8, 16, 32, 64, 128]), rnd.randint(4, 80), rnd.randint(80, 700)])
        if sw == 2:
            w += w % 2
        if sh == 2:
            h += h % 2
        b = rnd.randint(1, 3)
        pad = ((kh - 1) // 2, (kw - 1) // 2)
        if h + 2 * pad[0] < kh or w + 2 * pad[1] < kw:
            continue
        torch.manual_seed(case)
        x = torch.randn(b, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, kh, kw, device=dev) / (cin * kh * kw) ** 0.5
        bias = torch.randn(cout, device=dev)
        res = {}
        for impl in (IMPL_MFMA_BF16X3, 0):
            d = ops.conv2d_desc(b, cin, cout, h, w, kh, kw, (sh, sw), pad, EPI_LEAKY_PRE, 0.2, impl)
            d0 = ops.conv2d_desc(b, cin, cout, h, w, kh, kw, (sh, sw), pad, 0, 0.2, impl)
            y = ops.conv2d_forward(d, x, ops.conv2d_pack(d, wt), bias)
            if impl:
                dy, extra = torch.randn_like(y), torch.randn_like(x)
                names = (ops.conv2d_kernel_name(d), ops.conv2d_bwd_data_kernel_name(d0))
            pk = ops.conv2d_pack_bwd(d0, wt)
            g0 = ops.conv2d_bwd_data(d0, dy, pk)
            g1 = ops.conv2d_bwd_data(d0, dy, pk, x, 0.2, add=extra)
            dw, db = ops.conv2d_bwd_weight(d0, x, dy)
            res[impl] = (y, g0, g1, dw, db)
        a, r = res[IMPL_MFMA_BF16X3], res[0]
        note("forward: " + names[0], a[0], r[0])
        note("backward-data: " + names[1], a[1], r[1])
        note("backward-data, add + mask: " + names[1], a[2], r[2])
        note("weight gradient (bf16x3 descriptor)", a[3], r[3])
        note("bias gradient (bf16x3 descriptor)", a[4], r[4])
    torch.cuda.synchronize()
    bad = 0
    for k in sorted(worst):
        flag = "" if worst[k] <= 3e-5 else "   <-- ABOVE 3e-5"
        bad += worst[k] > 3e-5
        print(f"{count[k]:4d} cases  worst {worst[k]:.2e}  {k}{flag}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
