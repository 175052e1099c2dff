#!/bin/bash
# usage: scripts/gpu_prof_one.sh <tag> <script.py> [args]  -- rocprofv3 kernel-trace stats of one python script
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
cd $GRAFT_REPO_ROOT && python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)
if not f:
    print(open("gpurun_out/prof_$tag.log").read()[-2000:])
else:
    for r in list(csv.DictReader(open(f[0])))[:14]:
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
