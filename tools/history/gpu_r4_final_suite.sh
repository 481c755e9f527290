#!/bin/bash
# round 4, closing run 1: the whole GPU suite on the final tree (what the driver runs), then smoke()
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4final; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 8 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 $OUT/smoke.log | cut -c1-300
