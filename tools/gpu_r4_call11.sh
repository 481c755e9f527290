#!/bin/bash
# round 4, call 11: per-aggregate scans route narrow rows through the wave-specialised kernel (tests, per-kernel times)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c11; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 > $OUT/pytest_a.log 2>&1; echo "plan+fuzz rc=$?"; tail -n 6 $OUT/pytest_a.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "accumulators or avg or shared or operand or narrow or wide or skew or partition" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 6 $OUT/pytest_sel.log | cut -c1-300
DFX_NO_TORCH=1 timeout 300 python tools/qprobe.py 1073741824 headline,sum_min_w > $OUT/qprobe.txt 2>&1; tail -n 6 $OUT/qprobe.txt | cut -c1-220
