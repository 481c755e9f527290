#!/bin/bash
# round 4: does the headline's step time depend on how long the process has been running? (two fresh processes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4warm; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
for i in 1 2; do timeout 200 python tools/warm_probe.py 15 2>&1 | tail -n 2; done | tee $OUT/warm_probe.txt
