#!/bin/bash
# usage: scripts_gpu_run.sh <tag>   -- full evidence run on the GPU box:
#   pytest -m gpu, bench (hipGraph + eager), rocprofv3 kernel-trace stats, PMC passes
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -8 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rc=$?
tail -3 gpurun_out/bench_$tag.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-graph --cpu-items 0 --no-roofline --no-extra --no-train-step > gpurun_out/bench_eager_$tag.json 2>/dev/null
python -c "
import json
for f in ('bench_$tag.json','bench_eager_$tag.json'):
    r=json.load(open('gpurun_out/'+f)); print(f, r['ms_per_step'], r['value']/1e6, r['config'].get('launch'))
"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-items 0 --no-roofline --no-graph --no-extra --no-train-step > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
cd $GRAFT_REPO_ROOT && ./scripts/gpu_pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/pmc_$tag.log
python tools/pmc_summary.py gpurun_out/pmc_$tag 100 gpurun_out/pmc_traffic_$tag.json > gpurun_out/pmc_summary_$tag.txt 2>&1
exit $rc
