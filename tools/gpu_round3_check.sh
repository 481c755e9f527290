#!/bin/bash
# round 3, last check of the committed tree: smoke, the whole GPU suite, the bench line with the driver's flags
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3check; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E 'passed|failed|error' $OUT/pytest.log | tail -3
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall ${SECONDS}s"; cut -c1-300 $OUT/bench.json
