#!/bin/bash
# usage: scripts_gpu_run.sh <tag>   -- tests, bench, rocprof kernel trace (GPU box)
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -25 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rc=$?
tail -5 gpurun_out/bench_$tag.err
cat gpurun_out/bench_$tag.json
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-items 0 --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
rc=$?
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*stats*" | head
exit $rc
