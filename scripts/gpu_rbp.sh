#!/bin/bash
# usage: scripts/gpu_rbp.sh <tag> -- persistent residual block: parity tests, then A/B against the first kernel
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_resblock_p.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_rbp_$tag.log 2>&1
rc=$?
tail -12 gpurun_out/pytest_rbp_$tag.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 300 python tools/ab_bench.py rb_impl 0 1 > gpurun_out/ab_rbp_$tag.txt 2>&1
cat gpurun_out/ab_rbp_$tag.txt
