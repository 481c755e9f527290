#!/bin/bash
# round 3: one-sided predicates / scaled arguments on the static pass 1 (new parity test), then the default bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c10; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "one_sided" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest.log
SECONDS=0; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; echo "bench wall ${SECONDS}s"; cut -c1-900 $OUT/bench.json
