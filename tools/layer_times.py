#!/usr/bin/env python3
"""Per-launch times of one config-S forward (HIP events), in execution order."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audio_generation_amd import ops

def main():
    if os.environ.get("AGX_LIB"):      # A/B against a variant build of the library (measurement only)
        from audio_generation_amd import _lib as _l
        _l.LIB_PATH = os.path.abspath(os.environ["AGX_LIB"])
    dev = torch.device("cuda")
    model = bench.build_model(dev)
    x = bench.make_inputs(32, 0).to(dev)
    bench.calibrate_codebooks(model, x[:8], "latents")
    arith = sys.argv[1] if len(sys.argv) > 1 else "fp32"          # fp32 | mixed (decoder bf16x3) | bf16x3 (everything)
    model.set_conv_arithmetic(decoders="bf16x3" if arith != "fp32" else "fp32", encoders="bf16x3" if arith == "bf16x3" else "fp32")
    with torch.no_grad():
        for _ in range(2): model(x)
    agg = {}
    for rep in range(5):
        t = bench.LaunchTimer(); t.descs = []
        orig = t.begin
        def begin(kind, info, orig=orig, t=t):
            tok = orig(kind, info); t.descs.append((kind, info)); return tok
        t.begin = begin
        ops.set_observer(t)
        with torch.no_grad(): model(x)
        ops.set_observer(None)
        torch.cuda.synchronize()
        for i, ((name, (ex, ref, nb), e0, e1), (kind, info)) in enumerate(zip(t.records, t.descs)):
            agg.setdefault(i, []).append((name, ex, nb, e0.elapsed_time(e1), kind, info))
    tot = 0
    for i, v in agg.items():
        name, ex, nb, _, kind, info = v[0]
        ms = sorted(r[3] for r in v)[len(v)//2]
        tot += ms
        shape = (f"Cin={info.c_in} Cout={info.c_out} L={info.l_in} K={info.kernel} s={info.stride} d={info.dilation}"
                 if kind != "rvq" else str(info))
        print(f"{i:2d} {name:26s} {ms*1e3:8.1f} us  {2e-9*ex/ms:6.1f} TF  {1e-6*nb/ms:7.1f} GB/s  {shape}")
    print("sum", tot, "ms")
main()
