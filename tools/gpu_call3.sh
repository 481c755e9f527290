#!/bin/bash
# round 3, third GPU call: multi-process exchange tests, fused filter v3 (tests + probe), splits of the wave-specialised pass 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c3; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_exchange_world2.py -m gpu -q -x --timeout 400 > $OUT/pytest_exchange.log 2>&1; echo "pytest exchange rc=$?"; tail -n 25 $OUT/pytest_exchange.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x --timeout 800 -k "filter or golden or config2 or exchange or emulated or partial or partition or narrow" > $OUT/pytest_filter.log 2>&1; echo "pytest filter rc=$?"; tail -n 6 $OUT/pytest_filter.log
export DFX_NO_TORCH=1 KPROBE_BATCH_LOG2=27
echo "== filter probes"
timeout 120 python tools/filter_probe.py 1073741824 2>&1 | tail -2
timeout 120 python tools/filter_probe.py 1073741824 filter.single_pass=0 2>&1 | tail -2
for ws in 0 8 108 106 10 110 12 0 8 108; do echo "== headline pass1_ws=$ws"; timeout 120 python tools/kprobe.py 1e9 1e6 1 agg.pass1_ws=$ws 2>&1 | tail -3 | cut -c1-400; done
