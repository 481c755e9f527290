export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 500 -k "one_operand or shared_operand or three_word or narrow" > gpurun_out/pytest_shared.log 2>&1; echo "pytest rc=$?"; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/pytest_shared.log | tail -30
for o in "" "agg.shared_operand=0"; do echo "== neighbour $o"; timeout 200 python tools/prof_query.py neighbour 1e9 4 $o 2>&1 | grep -v amdgpu.ids | tail -3; done
