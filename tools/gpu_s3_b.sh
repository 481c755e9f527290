#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=268435456
echo "== q1 =="; for fg in 1 0; do timeout 300 python tools/prof_query.py q1 $R 3 agg.fewgroup=$fg 2>&1 | tail -2; done
echo "== headline cap_rows sweep =="
for a in "batch=67108864" "batch=67108864 agg.partition_cap_rows=1024" "batch=67108864 agg.partition_cap_rows=512" "batch=134217728 agg.partition_cap_rows=1024" "batch=268435456 agg.partition_cap_rows=2048" "batch=33554432" ; do echo "-- $a"; timeout 300 python tools/prof_query.py headline 1000000000 3 $a 2>&1 | tail -2; done
echo "== cfg3 / cfg2 =="; for wl in cfg3 cfg2; do timeout 300 python tools/prof_query.py $wl 1000000000 3 2>&1 | tail -2; done
