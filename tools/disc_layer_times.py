#!/usr/bin/env python3
"""Per-layer forward / backward-data / weight-gradient times of one STFT discriminator (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import discriminator as ad  # noqa: E402
from audio_generation_amd import ops  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    if os.environ.get("AGX_LIB"):      # A/B against a variant build of the library (measurement only)
        from audio_generation_amd import _lib as _l
        _l.LIB_PATH = os.path.abspath(os.environ["AGX_LIB"])
    win = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = "cuda"
    for kv in os.environ.get("AGX_KNOBS", "").split(","):     # e.g. AGX_KNOBS=patch_tie=0
        if kv:
            from audio_generation_amd import _lib
            _lib.load().agx_set_tuning(kv.split("=")[0].encode(), int(kv.split("=")[1]))
    d = ad.STFTDiscriminator(win_length=win).to(dev).train()
    if len(sys.argv) > 3:          # arithmetic: bf16x3 | bf16x3_ring
        ad.set_arithmetic(d, sys.argv[3])
    x = 0.1 * torch.randn(b, 1, 72000, device=dev)
    with torch.no_grad():
        h = ops.stft(x.squeeze(1), win, True)
        print(f"stft fwd {timeit(lambda: ops.stft(x.squeeze(1), win, True)):8.3f} ms   bwd {timeit(lambda: ops.stft_backward(h, 72000, win, True)):8.3f} ms")
        convs = [d.first_conv] + [c for blk in d.blocks for c in (blk.layers[0], blk.layers[2])] + [d.final_conv]
        tot = [0.0, 0.0, 0.0]
        for c in convs:
            y = c.run2d(h, None)
            tape = c._tape
            desc = c.desc2d(h, bwd=True)
            names = (ops.conv2d_kernel_name(c.desc2d(h, 0.2)).split("(")[0], ops.conv2d_bwd_data_kernel_name(desc).split("(")[0])
            w = c.raw_weight.detach()
            pk = ops.conv2d_pack_bwd(desc, w, tape[0])
            dy = torch.randn_like(y)
            hh = h
            tf = timeit(lambda: c.run2d(hh, None))
            am = (torch.randn_like(hh), torch.randn_like(hh))      # the op as the training step runs it: LeakyReLU mask + gradient add
            tx = timeit(lambda: ops.conv2d_bwd_data(desc, dy, pk))
            txm = timeit(lambda: ops.conv2d_bwd_data(desc, dy, pk, am[0], 0.2, add=am[1]))
            if c is d.first_conv:    # what _SNConv.bwd2d runs for the 2-channel layer
                tx2 = timeit(lambda: ops.conv2d_bwd_data_fewchannels(desc, dy, w, tape[0]))
                print(f"   first conv dx: direct {tx:.3f} ms, column-split {tx2:.3f} ms")
                tx = tx2
            tw = timeit(lambda: ops.conv2d_bwd_weight(desc, hh, dy, w, *tape))
            fl = 2.0 * y.numel() * c.in_channels * c.kernel_size[0] * c.kernel_size[1]
            print(f"{c.in_channels:4d}->{c.out_channels:4d} k{tuple(c.kernel_size)} s{tuple(c.stride)} in {tuple(h.shape[2:])}: "
                  f"fwd {tf:7.3f} ms ({fl / tf * 1e-9:5.1f} TF)  dx {tx:7.3f} ms ({fl / tx * 1e-9:5.1f} TF; +mask+add {txm:6.3f})  "
                  f"dW {tw:7.3f} ms ({fl / tw * 1e-9:5.1f} TF)   [{names[0]} | {names[1]}]")
            tot[0] += tf; tot[1] += tx; tot[2] += tw
            h = y
        print(f"total fwd {tot[0]:.2f}  dx {tot[1]:.2f}  dW {tot[2]:.2f} ms")


main()
