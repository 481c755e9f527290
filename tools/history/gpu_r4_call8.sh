#!/bin/bash
# round 4, call 8: the 4-byte-key flavour of the plan kernels (tests + per-kernel times), then the round's profile run
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c8; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 600 > $OUT/pytest_plan.log 2>&1; echo "plan rc=$?"; tail -n 6 $OUT/pytest_plan.log | cut -c1-300
DFX_NO_TORCH=1 timeout 300 python tools/qprobe.py 1073741824 headline,int32key,min,nullv > $OUT/qprobe.txt 2>&1; tail -n 40 $OUT/qprobe.txt | cut -c1-200
timeout 1500 bash tools/gpu_profile_r4.sh > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -n 60 $OUT/profile.log | cut -c1-260
