#!/bin/bash
# usage: scripts/gpu_r4_extras.sh <tag>  -- round-4 secondary evidence (kept under profiles/)
tag=${1:-r04}
o=gpurun_out
mkdir -p $o
timeout -k 10 200 python tools/layer_times.py > $o/layer_times_$tag.txt 2>/dev/null || exit 1
timeout -k 10 200 python tools/layer_times.py mixed > $o/layer_times_mixed_$tag.txt 2>/dev/null || exit 1
timeout -k 10 400 python tools/config_bench.py > $o/config3_config4_bench_$tag.jsonl 2>/dev/null || exit 1
timeout -k 10 300 python tools/rvq_verify_run.py > $o/rvq_verify_counts_$tag.txt 2>/dev/null; echo "rvq_verify rc=$?"
timeout -k 10 200 python tools/planes_times.py > $o/planes_times_$tag.txt 2>/dev/null || exit 1
for c in "32 72000 9" "64 36000 9" "64 36000 1" "128 9000 9" "256 1800 9"; do timeout -k 5 60 ./tools/b3_probe_bin $c || exit 1; done > $o/b3_probe_$tag.txt 2>&1
timeout -k 10 120 ./tools/mfma_bf16_err_bin > $o/mfma_bf16_err_$tag.txt 2>&1
tail -3 $o/layer_times_$tag.txt; tail -2 $o/layer_times_mixed_$tag.txt; cat $o/rvq_verify_counts_$tag.txt; grep -E "clock|kernel" $o/b3_probe_$tag.txt; cat $o/mfma_bf16_err_$tag.txt | cut -c1-200
