#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/debug_chunks.py 2>&1 | grep -v amdgpu.ids | tail -25
run() { echo "== $*"; timeout 300 python tools/kprobe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^rows=" ; }
export KPROBE_QUERIES=6
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=2
run 1e9 1e6 1 agg.partition_layout=1 agg.partition_defer=3
run 1e9 1e6 0 agg.partition_layout=1 agg.partition_defer=2
run 1e9 1e6 1 zipf agg.replay_in_place=1
run 1e9 1e6 0 zipf agg.replay_in_place=1
