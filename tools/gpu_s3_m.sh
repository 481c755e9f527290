#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 14 gpurun_out/pytest_gpu.log
