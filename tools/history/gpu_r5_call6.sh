#!/bin/bash
# round 5, call 6: the wave-specialised pass 1 with 4 scanner + 12 router waves on dense scans (selectivity 0.5 / 0.8 / 1.0)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c6; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_NO_TORCH=1
timeout 900 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 600 -x -k "four_scanners or clustered" > $OUT/pytest_ws.log 2>&1; echo "ws tests rc=$?"; tail -n 5 $OUT/pytest_ws.log | cut -c1-400
for opt in "agg.pass1_ws_dense=-1" "agg.pass1_ws_dense=1 agg.pass1_ws_dense_scanners=8" "agg.pass1_ws_dense=1 agg.pass1_ws_dense_scanners=4" "agg.pass1_ws_dense=1 agg.pass1_ws_dense_scanners=4"; do
  echo "== cfg3 dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_dense.txt
for hi in 716.8 1024.0; do for opt in "agg.pass1_ws_dense=-1" "agg.pass1_ws_dense_scanners=8" "agg.pass1_ws_dense_scanners=4"; do
  echo "== selectivity hi=$hi $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 lo=204.8 hi=$hi $opt 2>&1 | tail -n 3 | cut -c1-400
done; done | tee $OUT/kprobe_sel.txt
for opt in "agg.pass1_ws=8" "agg.pass1_ws=4"; do
echo "== headline $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_headline.txt
