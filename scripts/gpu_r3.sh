#!/bin/bash
# usage: scripts/gpu_r3.sh <tag> [pytest -k expr]  -- GPU tests (stop at the first failure), default bench line, configs 3 / 4
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x ${2:+-k "$2"} > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -12 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 500 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rc=$?
tail -3 gpurun_out/bench_$tag.err
if [ $rc -ne 0 ]; then echo "bench rc=$rc"; exit $rc; fi
python - <<PY
import json
r=json.load(open('gpurun_out/bench_$tag.json'))
print(r['ms_per_step'], r['value']/1e6, r['roofline']['kernel'], round(r['roofline']['frac'],4))
print({k:v['ms_per_step'] for k,v in r['roofline']['kernels'].items()})
print('parity', r['parity']); print('parity other', r.get('parity_other_codebooks'))
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
for o in r.get('other_arithmetic', []): print({k:v for k,v in o.items() if k!='arithmetic'})
PY
timeout -k 10 300 python tools/config_bench.py all > gpurun_out/config_$tag.jsonl 2> gpurun_out/config_$tag.err
rc=$?
tail -2 gpurun_out/config_$tag.err
python - <<PY
import json
for l in open('gpurun_out/config_$tag.jsonl'):
    r=json.loads(l); print(r['config'], r['ms_per_step'], r['waveform_rms_vs_oracle_clip0'], r.get('attention'), {k:v['avg_us'] for k,v in r['kernels'].items() if 'layernorm' in k or 'attention' in k})
PY
exit $rc
