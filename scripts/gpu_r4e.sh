#!/bin/bash
mkdir -p gpurun_out
tag=${1:-r4e}
python tools/layer_times.py mixed > gpurun_out/layer_times_mixed_$tag.txt 2>&1 || { tail -5 gpurun_out/layer_times_mixed_$tag.txt; exit 1; }
grep -E "b3|sum" gpurun_out/layer_times_mixed_$tag.txt
python tools/planes_times.py 2>/dev/null
for c in "32 72000 9" "64 36000 9" "128 9000 9"; do timeout -k 5 60 ./tools/b3_probe_bin $c || exit 1; done > gpurun_out/b3_probe_$tag.txt 2>&1
grep -E "kernel|whole tile|chunk 1 |GEMM2|epilogue" gpurun_out/b3_probe_$tag.txt
