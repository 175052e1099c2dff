#!/usr/bin/env python3
"""Diagnostic: where one residual stage of rvq_forward spends its cycles (agx_rvq_debug_stamps), on the bench's workload
(config S, batch 32 x 72 000, `latents` codebooks).  Thread 0 of every workgroup stamps s_memtime at the phase boundaries of
ONE stage; printed: median [p10 .. p90] over the workgroups, per phase, for every stage in turn.
The stamps (and the b3_dbg = 7 / 8 / 9 knobs that widen / zero / repurpose the score bound) exist in a PROBE copy of the library
only: `rvq_stamps.py build` compiles csrc/rvq.hip and csrc/core.hip with -DAGX_RVQ_PROBE (no GPU needed) and links them with the
product objects into audio_generation_amd/lib/libagx_rvq_probe.so, which the measuring run loads instead of libagx.so.
usage: rvq_stamps.py build | rvq_stamps.py [batch | C4] [stage ...]     (C4: BASELINE config 4 -- stereo, 8 x 144 000 samples, wavelet decoder)"""
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audio_generation_amd import _lib, ops  # noqa: E402

PHASES = [("score GEMM, pass 0", 2, 3), ("bounds + min exchange, pass 0", 3, 4), ("candidate lists, pass 0", 4, 5),
          ("score GEMM, pass 1", 5, 6), ("bounds + min exchange, pass 1", 6, 7), ("candidate lists, pass 1", 7, 8),
          ("decide (32 lanes)", 8, 9), ("binary64 distances of the listed pairs", 9, 10),
          ("pick + next stage's tables", 10, 11), ("overflow search", 11, 12), ("update r / out / norms", 12, 13),
          ("whole stage", 2, 13)]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "audio_generation_amd", "lib")
PROBE = os.path.join(LIBDIR, "libagx_rvq_probe.so")


def build():
    obj, src = os.path.join(LIBDIR, "obj"), os.path.join(ROOT, "audio_generation_amd", "csrc")
    probe_objs = []
    for name in ("rvq", "core"):
        o = os.path.join(LIBDIR, f"{name}_rvq_probe.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DAGX_RVQ_PROBE", "-I",
                               os.path.join(ROOT, "include"), "-I", src, "-Wno-unused-function", "-c",
                               os.path.join(src, name + ".hip"), "-o", o])
        probe_objs.append(o)
    objs = [os.path.join(obj, f) for f in sorted(os.listdir(obj)) if f.endswith(".o") and f not in ("rvq.o", "core.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE] + probe_objs + objs)
    print(PROBE)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return build()
    _lib.LIB_PATH = PROBE
    c4 = len(sys.argv) > 1 and sys.argv[1] == "C4"
    batch = 8 if c4 else (int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    stages = [int(a) for a in sys.argv[2:]] or list(range(8))
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(1234)
    if c4:
        from audio_generation_amd.vae import CausalVQAE
        torch.manual_seed(0)
        model = CausalVQAE(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024, codebook_dim=512,
                           input_format="n c l", wavelet_decoders=[False, True, False, False]).eval().to(dev)
        x = (0.1 * torch.randn(batch, 2, 144000, generator=gen)).clamp(-1, 1).to(dev)
        with torch.no_grad():
            model.quantizer.init_from_latents(model._run_encoders(x[:4]))
    else:
        model = bench.build_model(dev)
        x = (0.1 * torch.randn(batch, 1, 72000, generator=gen)).clamp(-1, 1).to(dev)
        bench.calibrate_codebooks(model, x[:4], "latents")
    lib = _lib.load()
    with torch.no_grad():
        z = model._run_encoders(x)
        n_wg = (z.shape[0] * z.shape[2] + 31) // 32
        buf = torch.zeros(n_wg * 16, dtype=torch.int64, device=dev)

        for _ in range(3):
            model.quantizer.quantize_bcl(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            model.quantizer.quantize_bcl(z)
        e1.record()
        torch.cuda.synchronize()
        print(f"quantiser call (rvq_forward + its reductions), stamps off: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us; {n_wg} workgroups")
        for q in stages:
            buf.zero_()
            lib.agx_rvq_debug_stamps(buf.data_ptr(), q)
            model.quantizer.quantize_bcl(z)
            torch.cuda.synchronize()
            lib.agx_rvq_debug_stamps(None, 0)
            t = buf.cpu().numpy().reshape(n_wg, 16)
            print(f"== stage {q}: whole kernel {np.median(t[:, 15] - t[:, 0]):.0f} ticks, stage loop {np.median(t[:, 1] - t[:, 0]):.0f}; "
                  f"pairs sent to the binary64 distance: median {np.median(t[:, 14]):.0f}, p90 {np.percentile(t[:, 14], 90):.0f}, max {t[:, 14].max()}")
            tot = (t[:, 15] - t[:, 0]).astype(np.float64)
            ovf = (t[:, 12] - t[:, 11]).astype(np.float64)
            print(f"   slowest workgroup: whole kernel {tot.max():.0f} ticks (p99 {np.percentile(tot, 99):.0f}); workgroups whose stage-{q} "
                  f"overflow search ran: {int((ovf > 2000).sum())} (longest {ovf.max():.0f} ticks)")
            for name, a, b in PHASES:
                d = (t[:, b] - t[:, a]).astype(np.float64)
                d = d[(t[:, a] > 0) & (t[:, b] > 0)]
                if len(d):
                    print(f"   {name:42s} {np.median(d):8.0f}  [{np.percentile(d, 10):8.0f} .. {np.percentile(d, 90):8.0f}]")


if __name__ == "__main__":
    main()
