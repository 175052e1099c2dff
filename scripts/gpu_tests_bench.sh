#!/bin/bash
# usage: scripts/gpu_tests_bench.sh <tag>  -- pytest -m gpu then the default bench line
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rc=$?
tail -3 gpurun_out/bench_$tag.err
python -c "
import json
r=json.load(open('gpurun_out/bench_$tag.json'))
print(r['ms_per_step'], r['value']/1e6, r['roofline']['kernel'], r['roofline']['frac'], r['roofline'].get('traffic_kernel'), r['roofline']['encoder'])
print(r['parity']); print(r['cpu_baseline']); print(r.get('other_codebooks')); print(r['config']['distinct_stage0_codes'])
"
exit $rc
