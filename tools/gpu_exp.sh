#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "strateg or partition or skew or keys or growth or sentinel or property" 2>&1 | tail -3
for f in 1 0; do timeout 120 python tools/kprobe.py 268435456 1e6 $f agg.strategy=3 2>&1 | grep -E "partition|groups_out"; done
