#!/usr/bin/env python3
"""Random-shape stress of the round-4 kernels against the kernels they replace (no oracle needed: every comparison is HIP vs HIP):
  * conv_p one-phase geometries (k1 / same11 / same3 incl. GELU / residual / post-activation epilogues) vs conv_mfma (knob conv_impl = 0);
  * conv_b3 strided down-convs + causal k3 (bf16x3) vs the fp32 ring;
  * conv_b3 fed with activation planes vs the same kernel fed with fp32 (must be bit-identical), plane output of the k7 layer.
Random batch 1..5, random lengths 1..3000 (ragged tails, clips shorter than a tile or a DMA cell), random activation flags.
usage: r4_stress.py [cases per family] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import _lib, ops  # noqa: E402

DEV = "cuda"
L = _lib


def knob(name, v):
    assert _lib.load().agx_set_tuning(name.encode(), v) == 0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))       # noqa: E731
    worst = {}

    def note(fam, err, scale, ctx):
        rel = err / max(scale, 1e-30)
        if rel > worst.get(fam, (0, None))[0]:
            worst[fam] = (rel, ctx)

    # ---- conv_p one-phase geometries vs conv_mfma ----
    one_phase = [(L.CONV_CAUSAL, 512, 512, 1), (L.CONV_CAUSAL, 512, 1536, 1), (L.CONV_CAUSAL, 128, 128, 1), (L.CONV_CAUSAL, 256, 256, 1),
                 (L.CONV_SAME, 256, 512, 11), (L.CONV_SAME, 512, 128, 3), (L.CONV_CAUSAL, 512, 512, 3)]
    epis = (0, L.EPI_LEAKY_PRE, L.EPI_GELU_PRE, L.EPI_RESIDUAL, L.EPI_RESIDUAL | L.EPI_LEAKY_POST, L.EPI_LEAKY_PRE | L.EPI_RESIDUAL,
            L.EPI_GELU_PRE | L.EPI_RESIDUAL | L.EPI_LEAKY_POST)
    for i in range(n):
        kind, cin, cout, k = one_phase[i % len(one_phase)]
        b, length, epi = ri(1, 5), ri(4, 3000), epis[ri(0, len(epis) - 1)]
        v = (torch.randn(cout, cin, k, generator=gen) / (cin * k) ** 0.5).to(DEV)
        bias = torch.randn(cout, generator=gen).to(DEV)
        x = torch.randn(b, cin, length, generator=gen).to(DEV)
        res = torch.randn(b, cout, length, generator=gen).to(DEV) if epi & L.EPI_RESIDUAL else None
        d = ops.conv_desc(kind, b, cin, cout, length, k, 1, 1, epi, 0.1, L.IMPL_AUTO)
        packed = ops.conv_pack(d, v)
        knob("conv_impl", 1)
        assert ops.conv_kernel_name(d).startswith("conv_p<"), ops.conv_kernel_name(d)
        y = ops.conv_forward(d, x, packed, bias, res=res)
        knob("conv_impl", 0)
        y0 = ops.conv_forward(d, x, packed, bias, res=res)
        knob("conv_impl", 1)
        note("conv_p one-phase vs conv_mfma", float((y - y0).abs().max()), float(y0.abs().max()), (cin, cout, k, b, length, epi))
    # ---- conv_b3 strided / k3 vs the fp32 ring ----
    strided = [(32, 64, 5, 2), (64, 128, 9, 4), (128, 256, 11, 5), (256, 512, 17, 8), (512, 512, 3, 1), (256, 512, 9, 4)]
    for i in range(n):
        cin, cout, k, s = strided[i % len(strided)]
        b, length, act = ri(1, 5), ri(max(k, 4), 3000), ri(0, 1)
        v = (torch.randn(cout, cin, k, generator=gen) / (cin * k) ** 0.5).to(DEV)
        bias = torch.randn(cout, generator=gen).to(DEV)
        x = torch.randn(b, cin, length, generator=gen).to(DEV)
        ys = []
        for impl in (L.IMPL_MFMA_BF16X3, L.IMPL_AUTO):
            d = ops.conv_desc(L.CONV_CAUSAL, b, cin, cout, length, k, s, 1, L.EPI_LEAKY_PRE if act else 0, 0.1, impl)
            ys.append(ops.conv_forward(d, x, ops.conv_pack(d, v), bias))
        note("conv_b3 strided / k3 vs the fp32 ring", float((ys[0] - ys[1]).abs().max()), float(ys[1].abs().max()), (cin, cout, k, s, b, length, act))
    # ---- planes in / out vs fp32 in (bit-identical) ----
    planes = [(L.CONV_TRANSPOSED, 512, 512, 7, 1), (L.CONV_UPSAMPLE, 512, 256, 17, 8), (L.CONV_UPSAMPLE, 256, 128, 11, 5),
              (L.CONV_UPSAMPLE, 128, 64, 9, 4), (L.CONV_UPSAMPLE, 64, 32, 5, 2)]
    mism = 0
    for i in range(n):
        kind, cin, cout, k, s = planes[i % len(planes)]
        b, length, act = ri(1, 5), ri(1, 2500), ri(0, 1)
        wshape = (cin, cout, k) if kind == L.CONV_TRANSPOSED else (cout, cin, k)
        v = (torch.randn(wshape, generator=gen) / (cin * k) ** 0.5).to(DEV)
        bias = torch.randn(cout, generator=gen).to(DEV)
        x = torch.randn(b, cin, length, generator=gen).to(DEV)
        d = ops.conv_desc(kind, b, cin, cout, length, k, s, 1, L.EPI_LEAKY_PRE if act else 0, 0.1, L.IMPL_MFMA_BF16X3)
        packed = ops.conv_pack(d, v)
        y = ops.conv_forward(d, x, packed, bias)
        yp = ops.conv_forward_planes(d, ops.planes_split(x), packed, bias)
        ok = torch.equal(y, yp)
        if kind == L.CONV_TRANSPOSED:
            ok = ok and torch.equal(ops.planes_join(ops.conv_forward_planes(d, ops.planes_split(x), packed, bias, out_planes=True)), y)
        mism += 0 if ok else 1
    torch.cuda.synchronize()
    for fam, (rel, ctx) in worst.items():
        print(f"{fam:42s} {n} cases: worst |difference| / max|y| = {rel:.2e}   at {ctx}")
    print(f"{'conv_b3 planes in / out vs fp32 input':42s} {n} cases: {mism} not bit-identical")
    bad = mism > 0 or any(rel > 2e-5 for rel, _ in worst.values())
    print("STRESS", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
