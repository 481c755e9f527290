#!/bin/bash
# Round 2, GPU call 3: routing-scratch layout / region size / producer count experiments, hot keys under Zipf, regression suite.
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python tools/kprobe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^rows=" ; }
run 1e9 1e6 1 agg.partition_layout=0 agg.partition_defer=1
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=1
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=4
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=1 agg.partition_cap_rows=512
run 1e9 1e6 1 agg.partition_layout=0 agg.partition_defer=1 agg.partition_cap_rows=512
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=1 agg.partition_producers=192
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=1 agg.partition_producers=128
run 1e9 1e6 0 agg.partition_layout=1 agg.partition_defer=1
run 1e9 1e6 0 agg.partition_layout=1 agg.partition_defer=4
run 1e9 1e6 0 agg.partition_layout=0 agg.partition_defer=1
run 1e9 1e6 1 zipf agg.replay_in_place=1
run 1e9 1e6 1 zipf agg.replay_in_place=1 agg.hot_keys=0
run 1e9 1e6 0 zipf agg.replay_in_place=1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 --deselect tests/test_gpu_scale.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 800 -k zipf > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 4 gpurun_out/pytest_scale.log
