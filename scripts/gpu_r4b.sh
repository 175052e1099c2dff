#!/bin/bash
# usage: scripts/gpu_r4b.sh <tag> "<pytest -k expr or empty>" [extra command]   -- selected GPU tests, then an extra measurement
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x ${2:+-k "$2"} > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
shift; shift
if [ $# -gt 0 ]; then
  timeout -k 10 600 "$@" > gpurun_out/extra_$tag.log 2>&1
  rc=$?
  tail -12 gpurun_out/extra_$tag.log | cut -c1-1500
fi
exit $rc
