#!/bin/bash
export TMPDIR=/tmp
DFX_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --rows 2e8 --prewarm-steps 3 2>&1 | tail -3 | cut -c1-900
