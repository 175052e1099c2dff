#!/usr/bin/env python3
"""Diagnostic: where the conv2d weight-gradient kernel (conv2d_bwd_weight_shared_kernel) spends its item loop.
Builds a probe copy of the library (conv_bwd_weight.hip with -DAGX_STAMPS, the other objects as they are) into
audio_generation_amd/lib/libagx_stamps.so when called with "build" (no GPU needed), otherwise loads it, runs one weight
gradient per layer shape and prints, per wave and item, the mean cycles of the four segments of an item:
   issue (LDS reads + next item's DMA), MFMAs issued, wait for the own DMA, barrier.
usage: dw_stamps.py build | dw_stamps.py [batch]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "audio_generation_amd", "lib")
PROBE = os.path.join(LIB, "libagx_stamps.so")


def build():
    obj = os.path.join(LIB, "obj")
    probe_o = os.path.join(LIB, "conv_bwd_weight_stamps.o")
    src = os.path.join(ROOT, "audio_generation_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DAGX_STAMPS", "-I",
                           os.path.join(ROOT, "include"), "-I", src, "-Wno-unused-function", "-c",
                           os.path.join(src, "conv_bwd_weight.hip"), "-o", probe_o])
    objs = [os.path.join(obj, f) for f in sorted(os.listdir(obj)) if f.endswith(".o") and f != "conv_bwd_weight.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE, probe_o] + objs)
    print(PROBE)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return build()
    import numpy as np
    import torch
    from audio_generation_amd import _lib
    _lib.LIB_PATH = PROBE
    from audio_generation_amd import ops
    lib = _lib.load()
    lib.agx_debug_read_stamps.restype = ctypes.c_int
    lib.agx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    buf = np.zeros(1 << 16, dtype=np.uint64)
    for (cin, cout, kh, kw, sh, sw, h, w) in [(128, 128, 3, 3, 1, 1, 141, 256), (256, 256, 3, 3, 1, 1, 70, 64),
                                              (128, 128, 3, 4, 1, 2, 141, 256), (128, 256, 4, 4, 2, 2, 141, 128)]:
        x = torch.randn(B, cin, h, w, device="cuda")
        pad = ((kh - 1) // 2, (kw - 1) // 2)
        d = ops.conv2d_desc(B, cin, cout, h, w, kh, kw, (sh, sw), pad)
        ho = (h + 2 * pad[0] - kh) // sh + 1
        wo = (w + 2 * pad[1] - kw) // sw + 1
        dy = torch.randn(B, cout, ho, wo, device="cuda")
        for _ in range(2):
            ops.conv2d_bwd_weight(d, x, dy)
        torch.cuda.synchronize()
        lib.agx_debug_read_stamps(buf.ctypes.data, 1 << 16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d_bwd_weight(d, x, dy)
        e1.record()
        torch.cuda.synchronize()
        lib.agx_debug_read_stamps(buf.ctypes.data, 1 << 16)
        t = buf.reshape(-1, 8).astype(np.float64)
        t = t[t[:, 4] > 0]
        per = t[:, :4] / t[:, 4:5]
        fl = 2.0 * dy.numel() * cin * kh * kw
        ms = e0.elapsed_time(e1)
        print(f"{cin}->{cout} k{kh}x{kw} s({sh},{sw}) in ({h},{w}): whole op {ms:.3f} ms = {fl / ms * 1e-9:.1f} TFLOP/s (stamped build); "
              f"{len(t)} waves, {t[:, 4].mean():.0f} items each; ticks per item and wave, mean [p10 .. p90]:")
        for k, name in enumerate(("issue: LDS reads + next DMA", "MFMAs (64, issue-limited)", "wait for own DMA", "barrier")):
            print(f"    {name:28s} {per[:, k].mean():8.0f}  [{np.percentile(per[:, k], 10):8.0f} .. {np.percentile(per[:, k], 90):8.0f}]")
        print(f"    {'sum':28s} {per.sum(axis=1).mean():8.0f}")


if __name__ == "__main__":
    main()
