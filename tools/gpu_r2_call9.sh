#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python tools/kprobe.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^rows=" ; }
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_scale.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_gpu.log
export KPROBE_QUERIES=8
run 1e9 1e6 1
run 1e9 1e6 0
run 1e9 1e6 1 zipf
run 1e9 1e6 0 zipf
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 800 > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 4 gpurun_out/pytest_scale.log
timeout 900 python bench.py > gpurun_out/bench_call9.json 2> gpurun_out/bench_call9.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_call9.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_call9.json"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "cold", d["extra"].get("cold_first_step_ms"))
    print(d["extra"]["kernels"])
    for k, v in d["extra"].items():
        if isinstance(v, dict) and "rows_per_s" in v: print(k, f"{v['rows_per_s']/1e9:.1f} Grows/s", v["roofline"]["frac"])
        elif isinstance(v, dict) and "error" in v: print(k, v)
except Exception as e:
    print("no bench line:", e)
PY
