#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu.log
bash tools/gpu_profile_bench.sh 2>&1 | tail -9 | cut -c1-250
