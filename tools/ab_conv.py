#!/usr/bin/env python3
"""In-process A/B timing of the resampling / bottleneck convs of config S (everything that is
not a fused residual block).  usage: ab_conv.py knob v0 v1 [...]   e.g.  ab_conv.py conv_short 0 1"""
import statistics
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import _lib  # noqa: E402
from audio_generation_amd.vae import CausalConv1d, CausalConvT1d, CausalUpsampleConv1d  # noqa: E402
from tools.ab_bench import time_fn  # noqa: E402

LAYERS = [  # (class, c_in, c_out, kernel, stride, l_in)
    (CausalConv1d, 32, 64, 5, 2, 72000), (CausalConv1d, 64, 128, 9, 4, 36000), (CausalConv1d, 128, 256, 11, 5, 9000),
    (CausalConv1d, 256, 512, 17, 8, 1800), (CausalConv1d, 512, 512, 3, 1, 225), (CausalConvT1d, 512, 512, 7, 1, 225),
    (CausalUpsampleConv1d, 512, 256, 17, 8, 225), (CausalUpsampleConv1d, 256, 128, 11, 5, 1800),
    (CausalUpsampleConv1d, 128, 64, 9, 4, 9000), (CausalUpsampleConv1d, 64, 32, 5, 2, 36000)]


def main():
    knob, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
    lib = _lib.load()
    tot = {v: 0.0 for v in values}
    for cls, ci, co, k, s, length in LAYERS:
        m = cls(ci, co, k, stride=s).to("cuda").eval()
        x = torch.randn(32, ci, length, device="cuda")
        res = {v: [] for v in values}
        for _ in range(7):
            for v in values:
                lib.agx_set_tuning(knob.encode(), v)
                with torch.no_grad():
                    m.run(x, 1, 0.1)
                    res[v].append(time_fn(lambda: m.run(x, 1, 0.1)))
        for v in values:
            tot[v] += statistics.median(res[v])
        print(f"{cls.__name__:22s} {ci:3d}->{co:3d} K={k:2d} s={s} L={length:5d}: " + "  ".join(
            f"{knob}={v}: med {statistics.median(r):7.1f} us min {min(r):7.1f}" for v, r in res.items()), flush=True)
    print("total: " + "  ".join(f"{knob}={v}: {t:8.1f} us" for v, t in tot.items()))
    lib.agx_set_tuning(knob.encode(), values[-1])


if __name__ == "__main__":
    main()
