#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms']); print(d['extra'])"
