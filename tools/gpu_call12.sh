#!/bin/bash
# round 3: range-form terms in FastPolicy + scalar-base loads for the generic policies: parity (parity + fuzz files), then timings
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c12; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 800 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest.log
export DFX_NO_TORCH=1
for wl in threeterm diffop threecol neighbour headline; do timeout 120 python tools/prof_query.py $wl 1e9 3 batch=134217728 2>&1 | tail -2; done
echo "== interpreter"; timeout 120 python tools/prof_query.py headline 1e9 3 batch=134217728 scan.fast=0 2>&1 | tail -2
