#!/bin/bash
# usage: scripts/gpu_extra.sh <tag>  -- the secondary measurements kept under profiles/ (not the headline metric)
tag=${1:-x}
o=gpurun_out
mkdir -p $o
timeout -k 10 200 python tools/layer_times.py > $o/layer_times_$tag.txt 2>/dev/null || exit 1
timeout -k 10 300 python tools/disc_bench.py 32 > $o/disc_forward_b32_$tag.jsonl 2>/dev/null || exit 1
timeout -k 10 400 python tools/disc_loss_bench.py 32 > $o/disc_loss_fwd_bwd_b32_$tag.jsonl 2>/dev/null || exit 1
timeout -k 10 300 python tools/train_step_bench.py 32 5 2>/dev/null | tail -1 > $o/train_step_generator_b32_$tag.json || exit 1
AGX_GAN=1 timeout -k 10 500 python tools/train_step_bench.py 32 3 2>/dev/null | tail -1 > $o/train_step_config5_b32_$tag.json || exit 1
AGX_GAN=1 AGX_BF16X3=1 timeout -k 10 500 python tools/train_step_bench.py 32 3 2>/dev/null | tail -1 > $o/train_step_config5_b32_bf16x3_$tag.json || exit 1
timeout -k 10 300 python tools/config_bench.py > $o/config3_config4_bench_$tag.jsonl 2>/dev/null || exit 1
timeout -k 10 100 python tools/signal_times.py > $o/signal_times_$tag.txt 2>/dev/null || exit 1
tail -2 $o/disc_loss_fwd_bwd_b32_$tag.jsonl; cut -c1-200 $o/train_step_config5_b32_$tag.json
