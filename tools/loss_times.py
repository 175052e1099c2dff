#!/usr/bin/env python3
"""Bandwidth of the loss reductions on one large feature-map pair (32 x 32 x 282 x 1024 floats = 1.18 GB per operand, the
first feature map of the window-1024 STFT discriminator at batch 32): agx_reduce_mean, agx_feature_means and their gradients."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import ops
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randn(32, 32, 282, 1024, device="cuda"); y = torch.randn_like(x)
gb = x.numel() * 4 / 1e9
g1, g2 = torch.tensor([0.5], device="cuda"), torch.tensor([0.5, 0.25], device="cuda")
for name, fn, passes in (
        ("reduce_mean L1 (2 reads)", lambda: ops.reduce_mean(x, ops.REDUCE_L1, y), 2),
        ("reduce_mean |x + eps| (1 read)", lambda: ops.reduce_mean(x, ops.REDUCE_ABS_EPS), 1),
        ("feature_means (2 reads)", lambda: ops.feature_means(x, y), 2),
        ("reduce_mean_backward L1 (2 reads, 2 writes)", lambda: ops.reduce_mean_backward(x, ops.REDUCE_L1, g1, y, True), 4),
        ("feature_means_backward (2 reads, 2 writes)", lambda: ops.feature_means_backward(x, y, g2), 4)):
    t = timeit(fn)
    print(f"{name:46s} {t:7.3f} ms  {passes * gb / t:6.2f} TB/s")
