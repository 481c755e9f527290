#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
DFX_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rows 2e8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/bench_shared2.log 2>&1; echo rc=$?; tail -n 3 gpurun_out/bench_shared2.log | cut -c1-1200
