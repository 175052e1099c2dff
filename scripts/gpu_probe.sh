#!/bin/bash
# usage: scripts/gpu_probe.sh <tag>  -- in-kernel stamp breakdown of the fused residual block + the bare MFMA ceilings
tag=${1:-x}
mkdir -p gpurun_out
{
for c in "64 36000 1" "64 36000 9" "32 72000 1" "128 9000 1" "256 1800 1"; do
  timeout -k 5 60 ./tools/rb_probe_bin $c || exit 1
done
timeout -k 5 120 ./tools/mfma_peak_bin
} > gpurun_out/probe_$tag.txt 2>&1
cat gpurun_out/probe_$tag.txt
