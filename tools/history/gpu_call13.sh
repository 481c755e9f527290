#!/bin/bash
# round 3: soak of the lock-free device protocols and differential runs between kernel families (tools/soak.py)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c13; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
timeout 400 python tools/soak.py ${SOAK_SECONDS:-240} ${SOAK_SEED:-2} > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -4 $OUT/soak.log | cut -c1-600
