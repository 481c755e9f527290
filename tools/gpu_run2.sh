#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 30 gpurun_out/pytest_gpu.log
echo "== bench =="; timeout 900 python bench.py --rows ${ROWS:-1e9} --steps 3 --warmup 1 --cpu-sample-rows 2e7 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -n 3 gpurun_out/bench.log
