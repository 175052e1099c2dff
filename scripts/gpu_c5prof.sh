#!/bin/bash
# usage: scripts/gpu_c5prof.sh <tag> [batch] -- rocprofv3 kernel stats of the config-5 training step (generator + 6 discriminators)
tag=${1:-x}
batch=${2:-32}
cd /tmp && export TMPDIR=/tmp
export AGX_GAN=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5prof_$tag -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py $batch 2 > $GRAFT_REPO_ROOT/gpurun_out/c5prof_$tag.log 2>&1
rc=$?
tail -2 $GRAFT_REPO_ROOT/gpurun_out/c5prof_$tag.log | cut -c1-300
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/c5prof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 5 steps (2 warm-up + 2 timed + the FLOP-count step)")
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6/5:9.2f} ms/step {100*float(r["TotalDurationNs"])/tot:5.1f}%  x{int(r["Calls"])//5:5d}  {r["Name"][:110]}')
PY
exit $rc
