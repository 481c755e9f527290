#!/bin/bash
# round 3, second GPU call: full GPU suite over the wave-specialised pass 1 + two-level look-back filter, then A/B timings
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c2; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 800 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 8 $OUT/pytest.log
export DFX_NO_TORCH=1 KPROBE_BATCH_LOG2=27
for ws in 0 12 8 14 0 12; do echo "== headline pass1_ws=$ws"; timeout 120 python tools/kprobe.py 1e9 1e6 1 agg.pass1_ws=$ws 2>&1 | tail -3; done
echo "== filter probes"
timeout 120 python tools/filter_probe.py 1073741824 2>&1 | tail -2
timeout 120 python tools/filter_probe.py 1073741824 filter.single_pass=0 2>&1 | tail -2
timeout 120 python tools/filter_probe.py 1073741824 scan.fast=0 2>&1 | tail -2
