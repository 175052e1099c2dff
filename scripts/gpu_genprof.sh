#!/bin/bash
# usage: scripts/gpu_genprof.sh <tag> -- rocprofv3 kernel stats of the generator-only training step (batch 32)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/genprof_$tag -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 32 3 > $GRAFT_REPO_ROOT/gpurun_out/genprof_$tag.log 2>&1
rc=$?
tail -1 $GRAFT_REPO_ROOT/gpurun_out/genprof_$tag.log | cut -c1-300
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/genprof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6/5:.1f} ms/step (2 warm-up + 3 timed)")
for r in rows[:32]:
    print(f'{float(r["TotalDurationNs"])/1e6/5:9.2f} ms/step {100*float(r["TotalDurationNs"])/tot:5.1f}%  x{int(r["Calls"])//5:5d}  {r["Name"][:120]}')
PY
exit $rc
