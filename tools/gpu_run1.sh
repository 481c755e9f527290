#!/bin/bash
# First GPU session: micro-benchmarks that decide the aggregate strategy + full GPU test-suite + bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E 'Marketing Name|gfx' | head -4 > gpurun_out/rocminfo.txt
echo "== ubench ==" ; timeout 600 ./tools/ubench > gpurun_out/ubench.jsonl 2> gpurun_out/ubench.err; echo "ubench rc=$?"; tail -n 70 gpurun_out/ubench.jsonl
echo "== smoke =="; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 5 gpurun_out/smoke.log
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -x --deselect tests/test_gpu_parity.py::test_large_properties_filter_groupby_sum > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 40 gpurun_out/pytest_gpu.log
echo "== bench small =="; timeout 900 python bench.py --rows 2.7e8 --steps 3 --warmup 1 --cpu-sample-rows 5e6 > gpurun_out/bench_small.log 2>&1; echo "bench rc=$?"; tail -n 5 gpurun_out/bench_small.log
