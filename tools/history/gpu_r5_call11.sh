#!/bin/bash
# round 5, call 11: dense split with the hash in the scanners; 2^27-row launches for dense scans
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c11; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_NO_TORCH=1
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "four_scanners or clustered or narrow_rows_fall_back or wide or sentinel or skew" > $OUT/pytest_ws.log 2>&1; echo "ws tests rc=$?"; tail -n 5 $OUT/pytest_ws.log | cut -c1-400
for opt in "agg.pass1_ws_dense_scanners=4" "agg.pass1_ws_dense_scanners=8" "agg.pass1_ws_dense_scanners=4" "agg.pass1_ws_dense_scanners=4 agg.partition_split_rows=134217728"; do
  echo "== cfg3 dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_dense.txt
for opt in "agg.pass1_ws_dense_scanners=8" "agg.pass1_ws_dense_scanners=4"; do
  echo "== selectivity hi=1024.0 $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 lo=204.8 hi=1024.0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_sel.txt
