#!/bin/bash
# Round 2, GPU call 2: streaming pass 2 + deferred pass 2 + emit/calibration changes: regression suite, A/B timings, bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 --deselect tests/test_gpu_scale.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -s --timeout 800 > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 8 gpurun_out/pytest_scale.log
for filt in 1 0; do
for opts in "agg.pass2_stream=0 agg.partition_defer=1 agg.emit_async=0 agg.calibration_memo=0" "agg.pass2_stream=1 agg.partition_defer=1" "agg.pass2_stream=0 agg.partition_defer=4" "agg.pass2_stream=1 agg.partition_defer=4" "agg.pass2_stream=1 agg.partition_defer=8 agg.partition_defer_batches=16"; do
  echo "== filt=$filt $opts"; timeout 300 python tools/kprobe.py 1e9 1e6 $filt $opts 2>&1 | grep -v amdgpu.ids | grep "un-instr\|partition\|hash_agg\|compact\|emit"
done; done
timeout 900 python bench.py > gpurun_out/bench_call2.json 2> gpurun_out/bench_call2.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_call2.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_call2.json"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    print(d["extra"]["kernels"])
    print({k: v for k, v in d["extra"].items() if k.startswith("cfg")})
except Exception as e:
    print("no bench line:", e)
PY
