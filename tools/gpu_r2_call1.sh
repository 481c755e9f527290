#!/bin/bash
# Round 2, GPU call 1: parity at the benchmark's size, the round-1 leftovers (in-place replay), the CU-mask probe, a
# Zipf timing and the baseline bench line of this box.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r2_call1.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/cumask_probe.hip -o /tmp/cumask_probe \
  && for k in 160 192 128; do timeout 120 /tmp/cumask_probe $k; done > gpurun_out/cumask_probe.jsonl 2> gpurun_out/cumask_probe.err; echo "cumask rc=$?"; cat gpurun_out/cumask_probe.jsonl; tail -3 gpurun_out/cumask_probe.err
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -s --timeout 800 > gpurun_out/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 15 gpurun_out/pytest_scale.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_csv.py -m gpu -q --timeout 300 -k "boolean_column_error or arrows_order or errors_mirror or pushdown" 2>&1 | tail -n 5
DFX_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "replayed_in_place" 2>&1 | tail -n 5
for opts in "" "agg.replay_in_place=1"; do timeout 300 python tools/kprobe.py 1e9 1e6 1 zipf $opts 2>&1 | tail -n 12; done
timeout 900 python bench.py > gpurun_out/bench_call1.json 2> gpurun_out/bench_call1.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench_call1.json; tail -2 gpurun_out/bench_call1.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_call1.json"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    print({k: v for k, v in d["extra"].items() if k.startswith("cfg")})
except Exception as e:
    print("no bench line:", e)
PY
