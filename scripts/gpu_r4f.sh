#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python tools/config_bench.py ${1:-all} > gpurun_out/config_bench_r4f.jsonl 2> gpurun_out/config_bench_r4f.err || { tail -5 gpurun_out/config_bench_r4f.err; exit 1; }
python - <<'PY'
import json
for ln in open('gpurun_out/config_bench_r4f.jsonl'):
    r=json.loads(ln)
    print(r['config'], round(r['ms_per_step'],3), 'rms', r['waveform_rms_vs_oracle_clip0'])
    for k,v in r['kernels'].items():
        if 'k1' in k or 'same' in k or 'conv_mfma' in k or 'layernorm' in k or 'attention' in k: print('   ',k,v)
PY
