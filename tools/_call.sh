mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 800 --deselect tests/test_gpu_scale.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log
run() { echo "== $*"; timeout 300 python tools/kprobe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^rows=" ; }
export KPROBE_QUERIES=60
run 1e9 1e6 1 export.kernel_copy=1
run 1e9 1e6 1 export.kernel_copy=0
HSA_ENABLE_SDMA=0 run 1e9 1e6 1 export.kernel_copy=0
