#!/bin/bash
# round 4, closing run 2: the bench line of the final tree (the driver's flags), the same command under rocprofv3, the counters of
# pass 1 (tools/gpu_profile_r4.sh), the routing-window probe of config 3, a short soak with the round's new alternatives
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4final; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4final/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "e2e", d["roofline"].get("end_to_end_frac"), "kernel", d["roofline"]["frac"])
for k, v in d["extra"].items():
    if isinstance(v, dict):
        r = v.get("roofline")
        f = r.get("frac") if isinstance(r, dict) else v.get("end_to_end_frac")
        ok = (v.get("verified_vs_oracle") or {}).get("ok") if isinstance(v.get("verified_vs_oracle"), dict) else None
        print(" ", k, f, v.get("ms"), "verified", ok)
PY
timeout 1500 bash tools/gpu_profile_r4.sh > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -n 14 $OUT/profile.log | cut -c1-200
export DFX_NO_TORCH=1
( for w in 16777216 33554432 67108864; do echo "== agg.partition_split_rows=$w"; timeout 200 python tools/qprobe.py 1073741824 dense,headline agg.partition_split_rows=$w 2>&1 | tail -n 2 | cut -c1-200; done ) | tee $OUT/cfg3_window_probe.txt
timeout 420 python tools/soak.py 300 4 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -n 4 $OUT/soak.log | cut -c1-250
