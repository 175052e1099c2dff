#!/usr/bin/env python3
"""A/B of the conv2d weight-gradient kernel over the dw_wgs knob (workgroups aimed for)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_generation_amd import ops, _lib
lib = _lib.load()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
knob = (sys.argv[2] if len(sys.argv) > 2 else "dw_wgs").encode()
vals = [int(v) for v in sys.argv[3:]] or [768, 512, 1024, 1536, 2048]
_narrow = [(128, 128, 3, 3, 1, 1, 1125, 16), (128, 256, 4, 4, 2, 2, 1125, 16), (256, 256, 3, 3, 1, 1, 562, 8), (256, 256, 3, 4, 1, 2, 562, 8), (256, 256, 3, 3, 1, 1, 562, 4)]
for (cin, cout, kh, kw, sh, sw, h, w) in _narrow if __import__('os').environ.get('AGX_DW_SHAPES') == 'narrow' else [(32, 32, 3, 3, 1, 1, 282, 1024), (64, 64, 3, 3, 1, 1, 282, 512), (128, 128, 3, 3, 1, 1, 141, 256), (64, 128, 4, 4, 2, 2, 282, 512), (256, 256, 3, 3, 1, 1, 70, 64)]:
    x = torch.randn(B, cin, h, w, device="cuda")
    pad = ((kh - 1) // 2, (kw - 1) // 2)
    d = ops.conv2d_desc(B, cin, cout, h, w, kh, kw, (sh, sw), pad, impl=int(__import__('os').environ.get('AGX_DW_IMPL', '0')))
    ho = (h + 2 * pad[0] - kh) // sh + 1; wo = (w + 2 * pad[1] - kw) // sw + 1
    dy = torch.randn(B, cout, ho, wo, device="cuda")
    fl = 2.0 * dy.numel() * cin * kh * kw
    out = []
    for a in vals:
        lib.agx_set_tuning(knob, a)
        t = timeit(lambda: ops.conv2d_bwd_weight(d, x, dy))
        out.append(f"{a}: {t:.3f} ms ({fl/t*1e-9:.1f} TF)")
    lib.agx_set_tuning(knob, vals[0])
    print(f"{cin}->{cout} k{kh}x{kw} s{sh}: " + "  ".join(out))
