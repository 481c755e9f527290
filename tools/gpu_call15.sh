#!/bin/bash
# round 3: pass 2 without the key-plane write-back when no slot was claimed: partitioned-strategy tests, timings, a short soak
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c15; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x --timeout 600 -k "partition or narrow or shared or skew or growth or resident or grouped or scale or headline or config" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E 'passed|failed' $OUT/pytest.log | tail -2
export DFX_NO_TORCH=1 KPROBE_BATCH_LOG2=27
for i in 1 2 3; do timeout 120 python tools/kprobe.py 1e9 1e6 1 2>&1 | tail -1; done
timeout 120 python tools/kprobe.py 1e9 1e6 1 lo=2000 hi=3000 2>&1 | tail -1
timeout 200 python tools/soak.py 60 5 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -1 $OUT/soak.log | cut -c1-300
