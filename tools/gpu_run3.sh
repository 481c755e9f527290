#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_gpu.log
for wl in cfg2 headline cfg3 q1; do python tools/prof_query.py $wl 268435456 3 2>&1 | tail -1; done
python tools/prof_query.py headline 268435456 3 agg.strategy=1 2>&1 | tail -1; python tools/prof_query.py cfg3 268435456 3 agg.strategy=1 2>&1 | tail -1
python tools/prof_query.py cfg2 268435456 3 scan.fast=0 2>&1 | tail -1
