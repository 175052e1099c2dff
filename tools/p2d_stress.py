#!/usr/bin/env python3
"""Randomised cross-check of the Conv2d ring kernel (conv_p.hip, D2 geometries) against the patch-tile kernel it replaces:
forward and backward-data (plain, masked, with a gradient add) on random layer shapes, both computed on the GPU and
switched with the knob conv_impl.  Prints the worst relative difference per geometry; exits non-zero above 3e-5.
usage: p2d_stress.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import _lib, ops  # noqa: E402
from audio_generation_amd._lib import EPI_LEAKY_PRE  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _lib.load()
    dev = "cuda"
    worst, count, skipped = {}, {}, 0
    for case in range(n):
        kw, sw = rnd.choice([(3, 1), (4, 2)])
        kh = rnd.choice([1, 3, 3, 4, 5]) if kw == 3 else rnd.choice([3, 4])
        sh = rnd.choice([1, 1, 2]) if kh >= 3 else 1
        cin = rnd.choice([32, 64, 128, 256])
        cout = rnd.choice([32, 64, 128, 256, 512])
        h, w, b = rnd.randint(1, 12), rnd.choice([rnd.randint(4, 80), rnd.randint(80, 700)]), rnd.randint(1, 3)
        ph = (kh - 1) // 2
        if h + 2 * ph < kh or w + 2 < kw:
            continue
        torch.manual_seed(case)
        x = torch.randn(b, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, kh, kw, device=dev) / (cin * kh * kw) ** 0.5
        bias = torch.randn(cout, device=dev)
        d = ops.conv2d_desc(b, cin, cout, h, w, kh, kw, (sh, sw), (ph, 1), EPI_LEAKY_PRE, 0.2)
        d0 = ops.conv2d_desc(b, cin, cout, h, w, kh, kw, (sh, sw), (ph, 1))
        lib.agx_set_tuning(b"conv_impl", 1)
        names = (ops.conv2d_kernel_name(d), ops.conv2d_bwd_data_kernel_name(d0))
        if not any(nm.startswith("conv_p2d") for nm in names):
            skipped += 1
            continue
        outs = []
        for impl in (1, 0):
            lib.agx_set_tuning(b"conv_impl", impl)
            y = ops.conv2d_forward(d, x, ops.conv2d_pack(d, wt), bias)
            dy = torch.randn_like(y) if impl == 1 else dy
            pk = ops.conv2d_pack_bwd(d0, wt)
            extra = torch.randn_like(x) if impl == 1 else extra
            g0 = ops.conv2d_bwd_data(d0, dy, pk)
            g1 = ops.conv2d_bwd_data(d0, dy, pk, x, 0.2, add=extra)
            outs.append((y, g0, g1))
        lib.agx_set_tuning(b"conv_impl", 1)
        for nm, i in ((names[0], 0), (names[1], 1), (names[1], 2)):
            a, r = outs[0][i], outs[1][i]
            err = float((a - r).abs().max()) / max(1.0, float(r.abs().max()))
            if not (err == err):
                err = float("inf")
            worst[nm] = max(worst.get(nm, 0.0), err)
            count[nm] = count.get(nm, 0) + 1
            if err > 3e-5:
                print("MISMATCH", nm, (b, cin, cout, kh, kw, sh, sw, h, w), err)
    for nm in sorted(worst):
        print(f"{nm:36s} checks {count[nm]:4d}  worst relative difference {worst[nm]:.2e}")
    print(f"{skipped} of {n} cases had no ring geometry")
    sys.exit(1 if any(v > 3e-5 for v in worst.values()) else 0)


if __name__ == "__main__":
    main()
