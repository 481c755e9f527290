#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "partition or skew" 2>&1 | tail -n 2
for wl in headline cfg3; do timeout 300 python tools/prof_query.py $wl 1000000000 3 2>&1 | tail -2; done
