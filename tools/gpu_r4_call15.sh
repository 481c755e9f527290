#!/bin/bash
# round 4, call 15: eight row groups per trip for the 4-byte-key flavour (fewer bytes per row group: more of them in flight)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c15; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
for i in 1 2 3; do timeout 300 python tools/qprobe.py 1073741824 headline,int32key 2>&1 | tail -n 2 | cut -c1-200; done | tee $OUT/qprobe.txt
timeout 600 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 600 -x -k "int32" > $OUT/pytest_a.log 2>&1; echo "int32 rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-300
