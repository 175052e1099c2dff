#!/usr/bin/env python3
"""Diagnostic: where conv2d_b3_kernel (the bf16x3 Conv2d ring) spends its time.  Builds a probe copy of the library (conv_b3.hip
with -DC2B3_STAMPS, the other objects as they are) into audio_generation_amd/lib/libagx_c2b3_stamps.so when called with "build"
(no GPU needed); otherwise loads it, runs the forward of a few discriminator layers (window 1024, batch 32) and prints, per wave
(mean over the waves of the launch), the cycles (s_memtime ticks) by segment:
   weight DMA issue | DMA wait (vmcnt) + barriers | MFMA groups | LDS operand reads issue | chunk end: split + LDS writes | epilogue | rest | chunk end: input loads issue
(second line per layer: the same launch without a bias vector)
usage: c2b3_stamps.py build | c2b3_stamps.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "audio_generation_amd", "lib")
PROBE = os.path.join(LIB, "libagx_c2b3_stamps.so")


def build():
    obj = os.path.join(LIB, "obj")
    probe_o = os.path.join(LIB, "conv_b3_stamps.o")
    src = os.path.join(ROOT, "audio_generation_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DC2B3_STAMPS", *(["-DC2B3_NOSTORE"] if os.environ.get("C2B3_NOSTORE") else []), "-I",
                           os.path.join(ROOT, "include"), "-I", src, "-Wno-unused-function", "-c",
                           os.path.join(src, "conv_b3.hip"), "-o", probe_o])
    objs = [os.path.join(obj, f) for f in sorted(os.listdir(obj)) if f.endswith(".o") and f != "conv_b3.o"]
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
    from audio_generation_amd._lib import EPI_LEAKY_PRE, IMPL_MFMA_BF16X3
    lib = _lib.load()
    lib.agx_debug_read_c2b3_stamps.restype = ctypes.c_int
    lib.agx_debug_read_c2b3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = 1 << 16
    buf = np.zeros(n, dtype=np.uint64)
    names = ["weight DMA issue", "DMA wait + barriers", "MFMA groups", "LDS reads", "split + LDS writes", "epilogue", "rest (tile start, step tails)", "input loads issue"]
    for cin, cout, h, w, kh, kw, sh, sw in [(32, 32, 282, 1024, 3, 3, 1, 1), (64, 64, 282, 512, 3, 3, 1, 1),
                                            (128, 128, 141, 256, 3, 3, 1, 1), (256, 256, 70, 64, 3, 3, 1, 1),
                                            (64, 128, 282, 512, 4, 4, 2, 2)]:
        b = 32
        x = torch.randn(b, cin, h, w, device="cuda")
        wt = torch.randn(cout, cin, kh, kw, device="cuda") / (cin * kh * kw) ** 0.5
        bias = torch.randn(cout, device="cuda")
        pad = ((kh - 1) // 2, (kw - 1) // 2)
        d = ops.conv2d_desc(b, cin, cout, h, w, kh, kw, (sh, sw), pad, EPI_LEAKY_PRE, 0.2, IMPL_MFMA_BF16X3)
        pk = ops.conv2d_pack(d, wt)
        ops.conv2d_forward(d, x, pk, bias)
        torch.cuda.synchronize()
        lib.agx_debug_read_c2b3_stamps(buf.ctypes.data, n)        # clear
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d_forward(d, x, pk, bias)
        e1.record()
        torch.cuda.synchronize()
        lib.agx_debug_read_c2b3_stamps(buf.ctypes.data, n)
        st = buf.reshape(-1, 8).astype(np.float64)
        st = st[st.sum(axis=1) > 0]
        tot = st[:, :8].sum(axis=1)
        print(f"{cin:4d}->{cout:4d} k({kh},{kw}) s({sh},{sw}) on ({h},{w}): {ops.conv2d_kernel_name(d)}  {e0.elapsed_time(e1):.3f} ms, "
              f"{len(st)} waves, mean ticks per wave {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
        print("     " + "  ".join(f"{nm} {100 * st[:, k].mean() / tot.mean():.1f}%" for k, nm in enumerate(names)))
        d0 = ops.conv2d_desc(b, cin, cout, h, w, kh, kw, (sh, sw), pad, EPI_LEAKY_PRE, 0.2, IMPL_MFMA_BF16X3)
        ops.conv2d_forward(d0, x, pk, None)
        torch.cuda.synchronize()
        lib.agx_debug_read_c2b3_stamps(buf.ctypes.data, n)
        st = buf.reshape(-1, 8).astype(np.float64)
        st = st[st.sum(axis=1) > 0]
        t2 = st[:, :8].sum(axis=1)
        print(f"     no bias: mean ticks per wave {t2.mean():.0f}; " + "  ".join(f"{nm} {100 * st[:, k].mean() / t2.mean():.1f}%" for k, nm in enumerate(names)))


main()
