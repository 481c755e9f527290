#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "fewgroup" 2>&1 | tail -n 2
for i in 1 2; do for fg in 1 0; do NOPROF=$((i-1)) timeout 300 python tools/prof_query.py q1 268435456 5 agg.fewgroup=$fg 2>&1 | tail -2; done; done
