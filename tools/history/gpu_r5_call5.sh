#!/bin/bash
# round 5, call 5: dense pass 1 -- full row groups routed straight from registers (ring kernel), the wave-specialised kernel on dense scans
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c5; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_NO_TORCH=1
for opt in "agg.pass1_ws_dense=0" "agg.pass1_ws_dense=1" "agg.pass1_ws_dense=0" "agg.pass1_ws_dense=1"; do
  echo "== cfg3 dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_dense.txt
for hi in 716.8 1024.0; do for opt in agg.pass1_ws_dense=0 agg.pass1_ws_dense=1; do
  echo "== selectivity hi=$hi $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 lo=204.8 hi=$hi $opt 2>&1 | tail -n 3 | cut -c1-400
done; done | tee $OUT/kprobe_sel.txt
echo "== headline"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 2>&1 | tail -n 3 | cut -c1-400 | tee $OUT/kprobe_headline.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "partition or narrow or wide or skew or shared or resident or grouped or tile or drain_split" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 6 $OUT/pytest_sel.log | cut -c1-300
