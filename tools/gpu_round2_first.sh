#!/bin/bash
# First GPU call of round 2 (one gpurun, ~6 min): is everything still green on a fresh box, and which pass-2 lookup variant wins?
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_round2_first.sh'
# Then, if the queue variant wins in the micro-benchmark: rebuild HERE with DFX_EXTRA_CXXFLAGS=-DDFX_PA_QUEUE (build.py reads it),
# run the GPU suite and bench.py again, and compare partition_agg in extra.kernels.
mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_pass2.hip -o /tmp/ubench_pass2 \
  && timeout 120 /tmp/ubench_pass2 256 > gpurun_out/ubench_pass2.jsonl 2> gpurun_out/ubench_pass2.err; echo "ubench_pass2 rc=$?"; cat gpurun_out/ubench_pass2.jsonl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_overlap.hip -o /tmp/ubench_overlap \
  && timeout 120 /tmp/ubench_overlap > gpurun_out/ubench_overlap.jsonl 2> gpurun_out/ubench_overlap.err; echo "ubench_overlap rc=$?"; cat gpurun_out/ubench_overlap.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu.log
DFX_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "replayed_in_place" 2>&1 | tail -n 5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_default.json
