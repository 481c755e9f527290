#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "avg_matches or more_than_eight or replayed" 2>&1 | tail -n 4
KPROBE_QUERIES=16 timeout 300 python tools/kprobe.py 1e9 1e6 1 2>&1 | grep -v amdgpu.ids | grep -v "^rows="
bash tools/gpu_profile_r2.sh
