#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "strateg or partition or skew or keys or growth or sentinel or property" > gpurun_out/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_part.log
for m in 2 1; do timeout 120 python tools/kprobe.py 268435456 1e6 1 agg.strategy=3 agg.partition_mode=$m 2>&1 | grep -E "partition|groups_out"; done
timeout 120 python tools/kprobe.py 268435456 1e6 0 agg.strategy=3 2>&1 | grep -E "partition|groups_out"
