#!/bin/bash
# usage: scripts_gpu_quick.sh <tag> [pytest -k expr]   -- tests + bench only
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q ${2:+-k "$2"} > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -25 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rc=$?
tail -5 gpurun_out/bench_$tag.err
cat gpurun_out/bench_$tag.json
exit $rc
