#!/usr/bin/env python3
"""Times of the 1-D weight-gradient kernel on the generator's layer shapes (config S, batch 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import ops, _lib
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CAUSAL, TRANSPOSED = 0, 1
tot = 0.0
shapes = [(CAUSAL, 32, 32, 72000, 7, 1, 9, 6), (CAUSAL, 32, 32, 72000, 1, 1, 1, 6), (CAUSAL, 32, 64, 72000, 5, 2, 1, 1),
          (CAUSAL, 64, 64, 36000, 7, 1, 3, 6), (CAUSAL, 64, 64, 36000, 1, 1, 1, 6), (CAUSAL, 64, 128, 36000, 9, 4, 1, 1),
          (CAUSAL, 128, 128, 9000, 7, 1, 1, 6), (CAUSAL, 128, 128, 9000, 1, 1, 1, 6), (CAUSAL, 128, 256, 9000, 11, 5, 1, 1),
          (CAUSAL, 256, 256, 1800, 7, 1, 9, 6), (CAUSAL, 256, 256, 1800, 1, 1, 1, 6), (CAUSAL, 256, 512, 1800, 17, 8, 1, 1),
          (CAUSAL, 512, 512, 225, 7, 1, 1, 1), (TRANSPOSED, 512, 256, 225, 17, 8, 1, 1), (TRANSPOSED, 128, 64, 9000, 9, 4, 1, 1)]
for (kind, cin, cout, l, k, s, d, count) in shapes:
    desc = ops.conv_desc(kind, B, cin, cout, l, k, s, d)
    lo = ops.conv_out_len(desc)
    x = torch.randn(B, cin, l, device="cuda"); dy = torch.randn(B, cout, lo, device="cuda")
    v = torch.randn(cout, cin, k, device="cuda") if kind == CAUSAL else torch.randn(cin, cout, k, device="cuda")
    g = torch.ones(v.shape[0], 1, 1, device="cuda")
    t = timeit(lambda: ops.conv_bwd_weight(desc, x, dy, v, g))
    fl = 2.0 * B * cin * cout * k * (lo if kind == CAUSAL else l)
    tot += t * count
    print(f"{'conv ' if kind == CAUSAL else 'convT'} {cin:4d}->{cout:4d} k{k:2d} s{s} d{d} L={l:6d}: {t:7.3f} ms ({fl / t * 1e-9:5.1f} TF)  x{count}")
print(f"weighted total {tot:.2f} ms")
