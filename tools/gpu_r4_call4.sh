#!/bin/bash
# round 4, call 4: the whole GPU suite (staged host stream, exchange fail-together, filter output sizing, tight SUM tolerance) + bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c4; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20; grep -E "ULP" $OUT/pytest.log | head -12
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall ${SECONDS}s"; tail -c 300 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c4/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["end_to_end_frac"], d["roofline"]["avg_launch_ms"])
for k,v in d["extra"].items():
    if isinstance(v,dict) and "roofline" in v and isinstance(v["roofline"],dict):
        vv=v.get("verified_vs_oracle") or {}
        print(f"{k:40s} {v.get('ms',0):9.2f} ms  frac {v['roofline']['frac']:.3f}  ok={vv.get('ok')} {str(vv.get('error',''))[:200]}")
    elif isinstance(v,dict) and "error" in v:
        print(k, "ERROR", v["error"])
PY
