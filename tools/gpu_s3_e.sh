#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "grouped or growth or resident or skew" 2>&1 | tail -n 2
for i in 1 2; do
timeout 300 python tools/prof_query.py headline 1000000000 5 2>&1 | tail -2
NOPROF=1 timeout 300 python tools/prof_query.py headline 1000000000 5 2>&1 | tail -1
done
