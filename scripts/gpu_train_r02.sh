#!/bin/bash
# usage: scripts/gpu_train_r02.sh <tag> -- training-step measurements: generator step, config-5 step (batch 32), 2-rank rehearsal
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 300 python tools/train_step_bench.py 32 3 > gpurun_out/train_gen_$tag.json 2> gpurun_out/train_gen_$tag.err || { tail -5 gpurun_out/train_gen_$tag.err; exit 1; }
cut -c1-400 gpurun_out/train_gen_$tag.json
AGX_GAN=1 timeout -k 10 500 python tools/train_step_bench.py 32 2 > gpurun_out/train_c5_$tag.json 2> gpurun_out/train_c5_$tag.err || { tail -5 gpurun_out/train_c5_$tag.err; exit 1; }
cut -c1-500 gpurun_out/train_c5_$tag.json
./scripts/gpu_2rank_train.sh > gpurun_out/train_2rank_$tag.txt 2>&1
cat gpurun_out/train_2rank_$tag.txt
