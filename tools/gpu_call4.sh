#!/bin/bash
# round 3, fourth GPU call: exchange tests (all cases), filter/partition tests over the specialised loops, probes, pin probe
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c4; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_exchange_world2.py -m gpu -q --timeout 400 > $OUT/pytest_exchange.log 2>&1; echo "pytest exchange rc=$?"; tail -n 12 $OUT/pytest_exchange.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x --timeout 800 -k "filter or golden or config2 or headline or exchange or emulated or partition or narrow or aggregates_of_one or fuzz" > $OUT/pytest_filter.log 2>&1; echo "pytest filter rc=$?"; tail -n 6 $OUT/pytest_filter.log
export DFX_NO_TORCH=1 KPROBE_BATCH_LOG2=27
echo "== filter probes"
timeout 120 python tools/filter_probe.py 1073741824 2>&1 | tail -2
timeout 120 python tools/filter_probe.py 1073741824 2>&1 | tail -2
for ws in 0 8 6 10 0 8 6; do echo "== headline pass1_ws=$ws"; timeout 120 python tools/kprobe.py 1e9 1e6 1 agg.pass1_ws=$ws 2>&1 | tail -3 | cut -c1-400; done
echo "== pin probe"; timeout 120 python tools/pin_probe.py 2>&1 | tail -12
