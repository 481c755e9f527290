#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu.log
for i in 1 2; do for wl in headline cfg3; do timeout 300 python tools/prof_query.py $wl 1000000000 3 2>&1 | tail -2; done; done
timeout 300 python tools/prof_query.py headline 268435456 3 agg.strategy=1 2>&1 | tail -2
