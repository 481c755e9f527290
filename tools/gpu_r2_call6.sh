#!/bin/bash
# Round 2, GPU call 6: lean pass 2 (reserved-register prefetch, branch-free fast path), regression + timings.
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python tools/kprobe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^rows=" ; }
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_scale.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_gpu.log
export KPROBE_QUERIES=10
run 1e9 1e6 1
run 1e9 1e6 1 agg.narrow_keys=0
run 1e9 1e6 1 agg.pass2_stream=0
run 1e9 1e6 0
run 1e9 1e6 0 agg.narrow_keys=0
run 1e9 1e6 1 zipf agg.replay_in_place=1
run 1e9 1e6 0 zipf agg.replay_in_place=1
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 800 > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 4 gpurun_out/pytest_scale.log
timeout 900 python bench.py > gpurun_out/bench_call6.json 2> gpurun_out/bench_call6.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_call6.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_call6.json"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    print(d["extra"]["kernels"])
    print({k: v for k, v in d["extra"].items() if k.startswith("cfg")})
except Exception as e:
    print("no bench line:", e)
PY
