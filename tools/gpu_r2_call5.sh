#!/bin/bash
# Round 2, GPU call 5: pass 2 with explicitly pipelined row loads; where do the sporadic slow queries come from; library exchange (RCCL, world 1).
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python tools/kprobe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^rows=" ; }
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 --deselect tests/test_gpu_scale.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu.log
export KPROBE_QUERIES=12
run 1e9 1e6 1 agg.partition_defer=1
KPROBE_NOGC=1 run 1e9 1e6 1 agg.partition_defer=1
run 1e9 1e6 1 agg.partition_defer=1 agg.narrow_keys=0
run 1e9 1e6 1 agg.partition_defer=4
run 1e9 1e6 1 agg.partition_defer=2
run 1e9 1e6 0 agg.partition_defer=1
run 1e9 1e6 0 agg.partition_defer=1 agg.narrow_keys=0
run 1e9 1e6 0 agg.partition_defer=2
run 1e9 1e6 1 zipf agg.replay_in_place=1
run 1e9 1e6 0 zipf agg.replay_in_place=1
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 800 > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 4 gpurun_out/pytest_scale.log
