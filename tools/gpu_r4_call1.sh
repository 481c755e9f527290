#!/bin/bash
# round 4, call 1: the scan-plan kernels on the device for the first time -- smoke, the new plan tests, the whole GPU suite, one bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c1; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 300 > $OUT/pytest_plan.log 2>&1; echo "plan tests rc=$?"; tail -n 25 $OUT/pytest_plan.log | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_plan.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest.log | cut -c1-400
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall ${SECONDS}s"; tail -c 600 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c1/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["end_to_end_frac"], d["roofline"]["avg_launch_ms"])
for k,v in d["extra"].items():
    if isinstance(v,dict) and "roofline" in v:
        print(f"{k:40s} {v['ms']:9.2f} ms  frac {v['roofline']['frac']:.3f}  ok={(v.get('verified_vs_oracle') or {}).get('ok')} {(v.get('verified_vs_oracle') or {}).get('error','')[:200]}")
    elif isinstance(v,dict) and "error" in v:
        print(k, "ERROR", v["error"])
print({k:v for k,v in d["extra"].get("rows_1e10",{}).items() if k in ("ms","roofline")})
PY
