#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "emulated" 2>&1 | tail -3
DFX_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rows 2e8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/bench_shared2.log 2>&1; echo rc=$?; tail -n 1 gpurun_out/bench_shared2.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra'].get('verified_sum_of_group_sums_equals_ungrouped_sum')); print({k:v for k,v in d['extra']['kernels'].items()})"
