#!/bin/bash
# round 4, call 12: the dense flavour of the single-pass filter (tile in registers): tests, then the bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c12; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "filter" > $OUT/pytest_filter.log 2>&1; echo "filter rc=$?"; tail -n 8 $OUT/pytest_filter.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q --timeout 800 -k "config2" > $OUT/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -n 4 $OUT/pytest_scale.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c12/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "e2e", d["roofline"].get("end_to_end_frac"), "kernel", d["roofline"]["frac"])
for k, v in d["extra"].items():
    if isinstance(v, dict):
        r = v.get("roofline")
        f = r.get("frac") if isinstance(r, dict) else v.get("end_to_end_frac")
        print(" ", k, f, v.get("ms"))
PY
