#!/bin/bash
# round 5, call 1: Infinity Cache / LDS micro-benchmarks, the tile-sorted pass 1 (parity tests, per-kernel times against the ring kernel)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c1; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 120 tools/ubench3 > $OUT/ubench3.jsonl 2>&1; echo "ubench3 rc=$?"; cat $OUT/ubench3.jsonl | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "tile_sorted or after_the_drain_split" > $OUT/pytest_tile.log 2>&1; echo "tile tests rc=$?"; tail -n 12 $OUT/pytest_tile.log | cut -c1-400
export DFX_NO_TORCH=1
for opt in agg.pass1_tile=0 agg.pass1_tile=1; do
  echo "== cfg3 dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 4 | cut -c1-400
done | tee $OUT/kprobe_dense.txt
for opt in agg.pass1_tile=0 agg.pass1_tile=1; do
  echo "== wide keys dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 wide $opt 2>&1 | tail -n 4 | cut -c1-400
done | tee $OUT/kprobe_wide.txt
echo "== headline (selective)"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 2>&1 | tail -n 4 | cut -c1-400 | tee $OUT/kprobe_headline.txt
echo "== wide keys with the headline's filter"; timeout 300 python tools/kprobe.py 1073741824 1e6 1 wide 2>&1 | tail -n 4 | cut -c1-400 | tee -a $OUT/kprobe_wide.txt
echo "== dense, 2^24-row launches (routed rows of a window fit the Infinity Cache)"
for opt in "agg.partition_split_rows=16777216" "agg.partition_split_rows=33554432"; do
  timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_windows.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "partition or narrow or wide or skew or shared or resident or grouped" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 6 $OUT/pytest_sel.log | cut -c1-300
