#!/bin/bash
# round 3: soak of the lock-free device protocols (tools/soak.py), default options and the ring-kernel / two-pass alternatives
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c13; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "failed_allocation" 2>&1 | tail -2; timeout 400 python tools/soak.py 300 1 > $OUT/soak1.log 2>&1; echo "soak rc=$?"; tail -4 $OUT/soak1.log | cut -c1-300
