#!/bin/bash
# round 4: ten more minutes of tools/soak.py on the final tree (three seeds; differential alternatives include the round's new options)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4soak; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
for seed in 11 12 13; do timeout 260 python tools/soak.py 200 $seed > $OUT/soak_$seed.log 2>&1; echo "soak seed $seed rc=$?"; tail -n 1 $OUT/soak_$seed.log | cut -c1-200; done
