#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "partition or skew or resident or growth or grouped_aggregates or exchange or large_properties or fused" 2>&1 | tail -n 3
for i in 1 2; do
for m in 2 258; do echo "-- partition_mode=$m"; for wl in headline cfg3; do timeout 300 python tools/prof_query.py $wl 1000000000 3 agg.partition_mode=$m 2>&1 | tail -2; done; done
done
