#!/bin/bash
# round 3, sixth GPU call: the full GPU suite, two-batch routing (config 3 dense + the routers), host streaming with / without pinning
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c6; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 800 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest.log
export DFX_NO_TORCH=1 KPROBE_BATCH_LOG2=27
for o in "agg.pass1_ws=8" "agg.pass1_ws=0" "agg.pass1_ws=8"; do echo "== headline $o"; timeout 120 python tools/kprobe.py 1e9 1e6 1 $o 2>&1 | tail -3 | cut -c1-400; done
for i in 1 2; do echo "== config 3 (no filter)"; timeout 120 python tools/kprobe.py 1e9 1e6 0 2>&1 | tail -3 | cut -c1-400; done
echo "== host streaming"; timeout 120 python tools/host_stream_probe.py 2>&1 | tail -3
DFX_HOST_PIN=1 timeout 120 python tools/host_stream_probe.py 2>&1 | tail -3
