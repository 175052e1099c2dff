#!/bin/bash
# rehearsal of the N>1 TRAINING path on a 1-GPU box: 2 ranks share the GPU, gloo carries the one flattened
# gradient all-reduce (generator + discriminators); replicas must stay bit-identical (replicas_in_sync)
export AGX_DIST_BACKEND=gloo AGX_GAN=1 AGX_GAN_WINS=1024,128
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/train_step_bench.py 2 2 2>&1 | tail -2 | cut -c1-600
