#!/bin/bash
# rehearsal of the N>1 launch path on a 1-GPU box: 2 ranks share the GPU, gloo for the barrier/reduce
export AGX_DIST_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --cpu-items 0 2>&1 | tail -3 | cut -c1-700
