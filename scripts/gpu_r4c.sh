#!/bin/bash
# config bench (C3 / C4) + the bf16x3 block's stamp probe
mkdir -p gpurun_out
timeout -k 10 500 python tools/config_bench.py all > gpurun_out/config_bench_r4c.jsonl 2> gpurun_out/config_bench_r4c.err || { tail -5 gpurun_out/config_bench_r4c.err; exit 1; }
python - <<'PY'
import json
for ln in open('gpurun_out/config_bench_r4c.jsonl'):
    r=json.loads(ln)
    print(r['config'], round(r['ms_per_step'],3), 'rms', r['waveform_rms_vs_oracle_clip0'])
    for k,v in r['kernels'].items():
        if 'k1' in k or 'same' in k or 'conv_mfma' in k or 'layernorm' in k or 'attention' in k: print('   ',k,v)
PY
for c in "32 72000 9" "64 36000 9" "128 9000 9" "256 1800 9" "64 36000 1"; do timeout -k 5 60 ./tools/b3_probe_bin $c || exit 1; done > gpurun_out/b3_probe_r4c.txt 2>&1
cat gpurun_out/b3_probe_r4c.txt
