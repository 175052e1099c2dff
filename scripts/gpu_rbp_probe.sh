#!/bin/bash
tag=${1:-x}
mkdir -p gpurun_out
{
for c in "64 36000 1" "32 72000 1" "128 9000 1" "256 1800 1"; do timeout -k 5 60 ./tools/rbp_probe_bin $c || exit 1; done
} > gpurun_out/rbp_probe_$tag.txt 2>&1
cat gpurun_out/rbp_probe_$tag.txt
