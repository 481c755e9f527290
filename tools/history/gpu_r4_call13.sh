#!/bin/bash
# round 4, call 13: the two flavours of the single-pass filter across selectivities
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c13; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
for sel in 0.02 0.1 0.2 0.3 0.5 0.8; do for d in 0 1; do
  timeout 120 python tools/filter_probe.py 1073741824 sel=$sel filter.dense=$d 2>&1 | grep "filter as written" | cut -c1-260
done; done | tee $OUT/filter_flavours.txt
