#!/bin/bash
# round 4: the driver's bench command once more on whatever box comes up (the spread between boxes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4again; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4again/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "e2e", d["roofline"].get("end_to_end_frac"), "kernel", d["roofline"]["frac"])
for k, v in d["extra"].items():
    if isinstance(v, dict):
        r = v.get("roofline")
        f = r.get("frac") if isinstance(r, dict) else v.get("end_to_end_frac")
        print(" ", k, f)
PY
