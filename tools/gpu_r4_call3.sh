#!/bin/bash
# round 4, call 2: plan tests again (validity addressing fixed, library rebuilt), the tests call 1 failed, fuzz, bench, routing-window probes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c3; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 300 > $OUT/pytest_plan.log 2>&1; echo "plan tests rc=$?"; tail -n 12 $OUT/pytest_plan.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 -k "utf8_keys or int32 or multi_column or fewgroup or fuzz or nulls or null or avg or ungrouped" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 8 $OUT/pytest_sel.log | cut -c1-300
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c3/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["end_to_end_frac"], d["roofline"]["avg_launch_ms"])
for k,v in d["extra"].items():
    if isinstance(v,dict) and "roofline" in v and isinstance(v["roofline"],dict):
        vv=v.get("verified_vs_oracle") or {}
        print(f"{k:40s} {v.get('ms',0):9.2f} ms  frac {v['roofline']['frac']:.3f}  ok={vv.get('ok')} {str(vv.get('error',''))[:200]}")
    elif isinstance(v,dict) and "error" in v:
        print(k, "ERROR", v["error"])
PY
timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_soak.py -m gpu -q --timeout 600 > $OUT/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 5 $OUT/pytest_scale.log | cut -c1-300
