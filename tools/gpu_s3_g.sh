#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err
bash tools/gpu_profile_bench.sh 2>&1 | tail -14
