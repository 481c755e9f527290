#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_csv.py -q -x --timeout 600 -s 2>&1 | tail -n 30
