#!/usr/bin/env python3
"""Diagnostic: resblock_p with a dedicated DMA wave (knob rb_lw, one workgroup per CU) against the shipped form
(DMA instructions spread over the MFMA waves' phases), same results required; d = 1 blocks of config S."""
import statistics
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import _lib  # noqa: E402
from audio_generation_amd.vae import CausalResidualBlock1d  # noqa: E402
from tools.ab_bench import time_fn  # noqa: E402

lib = _lib.load()


def setk(**kw):
    for k, v in kw.items():
        assert lib.agx_set_tuning(k.encode(), v) == 0


for c, length in [(32, 72000), (64, 36000), (128, 9000), (256, 1800)]:
    m = CausalResidualBlock1d(c, c, dilation=1).to("cuda").eval()
    x = torch.randn(32, c, length, device="cuda")
    res, out = {}, {}
    for name, kw in (("2wg", dict(rb_lw=0, rb_wgs=0)), ("1wg", dict(rb_lw=0, rb_wgs=1)), ("1wg+loader", dict(rb_lw=1, rb_wgs=0))):
        setk(**kw)
        with torch.no_grad():
            out[name] = m.run(x, 0.1)
            res[name] = [time_fn(lambda: m.run(x, 0.1)) for _ in range(7)]
    setk(rb_lw=0, rb_wgs=0)
    assert torch.equal(out["2wg"], out["1wg"]), "1wg differs"
    d = float((out["2wg"] - out["1wg+loader"]).abs().max())
    flops = 2.0 * 32 * c * c * 8 * length
    print(f"C={c:3d}: " + "  ".join(f"{k}: {statistics.median(v):7.1f} us ({flops / statistics.median(v) * 1e-6:5.1f} TF)" for k, v in res.items())
          + f"  max|diff| loader vs shipped {d:.1e}", flush=True)
    assert d == 0.0
