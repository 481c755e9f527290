#!/bin/bash
# round 3, seventh GPU call: full GPU suite over the trimmed scan loops (reduce, ring, fewgroup), the bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c7; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --timeout 800 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 8 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json, os
try:
    d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3c7/bench.json"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "e2e", d["roofline"].get("end_to_end_frac"), "traffic", d["roofline"].get("traffic_over_algorithmic"), "cold", d["extra"].get("cold_first_step_ms"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
    print(d["extra"]["kernels"])
    for k, v in d["extra"].items():
        if isinstance(v, dict) and "rows_per_s" in v: print("  ", k, f"{v['rows_per_s']/1e9:.1f} Grows/s", v["roofline"]["frac"], (v.get("verified_vs_oracle") or {}).get("ok") if isinstance(v.get("verified_vs_oracle"), dict) else v.get("verified_vs_oracle"))
        elif isinstance(v, dict) and "error" in v: print("  ", k, v)
except Exception as e:
    print("no bench line:", e)
PY
