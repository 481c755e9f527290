#!/bin/bash
# round 4, call 7: the tests call 6 failed (per-aggregate scans next to the un-fused filter path)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c7; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_plan.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 > $OUT/pytest_a.log 2>&1; echo "plan+fuzz rc=$?"; tail -n 6 $OUT/pytest_a.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "accumulators or avg or shared or one_operand or aggregate_over_filter or computed" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 4 $OUT/pytest_sel.log | cut -c1-300
