#!/bin/bash
# round 3: per-kernel breakdown of the queries beside the headline (where the off-signature time goes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c9; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
for wl in headline neighbour oneterm product diffop threecol cfg3; do
  timeout 120 python tools/prof_query.py $wl 1e9 3 batch=134217728 2>&1 | tail -2
done
echo "== interpreter"; timeout 120 python tools/prof_query.py headline 1e9 3 batch=134217728 scan.fast=0 2>&1 | tail -2
echo "== headline, ring kernel"; timeout 120 python tools/prof_query.py headline 1e9 3 batch=134217728 agg.pass1_ws=0 2>&1 | tail -2
