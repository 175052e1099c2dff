#!/bin/bash
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trainprof_$tag -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 16 3 > $GRAFT_REPO_ROOT/gpurun_out/trainprof_$tag.log 2>&1
rc=$?
tail -2 $GRAFT_REPO_ROOT/gpurun_out/trainprof_$tag.log | cut -c1-300
exit $rc
