#!/bin/bash
# round 4, call 9: three-column plan kernels (tests, per-kernel times), then the whole bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c9; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 600 > $OUT/pytest_plan.log 2>&1; echo "plan rc=$?"; tail -n 6 $OUT/pytest_plan.log | cut -c1-300
DFX_NO_TORCH=1 timeout 300 python tools/qprobe.py 1073741824 headline,sum_min_w,int32key,three > $OUT/qprobe.txt 2>&1; tail -n 12 $OUT/qprobe.txt | cut -c1-220
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c9/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "e2e", d["roofline"].get("end_to_end_frac"), "kernel", d["roofline"]["frac"])
for k, v in d["extra"].items():
    if isinstance(v, dict):
        r = v.get("roofline")
        f = r.get("frac") if isinstance(r, dict) else v.get("end_to_end_frac")
        print(" ", k, f, v.get("ms"))
PY
