#!/usr/bin/env python3
"""A few launches of the bf16x3 ring block on the config-S layer shapes (for rocprofv3 passes).  usage: b3_run.py [reps]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import _lib  # noqa: E402
from audio_generation_amd.vae import CausalResidualBlock1d  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for c, length in ((32, 72000), (64, 36000), (128, 9000), (256, 1800)):
    m = CausalResidualBlock1d(c, c, dilation=9).to("cuda").eval()
    m.conv1.impl = _lib.IMPL_MFMA_BF16X3
    x = torch.randn(32, c, length, device="cuda")
    with torch.no_grad():
        for _ in range(reps):
            m.run(x, 0.1)
torch.cuda.synchronize()
print("done")
