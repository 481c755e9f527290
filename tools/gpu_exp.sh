#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for m in 2 130 34 130; do timeout 120 python $R/tools/kprobe.py 268435456 1e6 1 agg.strategy=3 agg.partition_mode=$m 2>&1 | grep -E "partition |groups_out"; done
timeout 120 python $R/tools/kprobe.py 268435456 1e6 1 agg.strategy=3 lo=1e9 hi=2e9 2>&1 | grep -E "partition |groups_out"
