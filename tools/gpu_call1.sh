#!/bin/bash
# round 3, first GPU call: the new single-pass filter + parity-at-scale tests, the bench line, instruction-cache counters of pass 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c1; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x --timeout 800 -k "filter or golden or scale or config or headline or uniform or zipf" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 8 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json, os
try:
    d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3c1/bench.json"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "e2e", d["roofline"].get("end_to_end_frac"), "cold", d["extra"].get("cold_first_step_ms"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
    print(d["extra"]["kernels"])
    for k, v in d["extra"].items():
        if isinstance(v, dict) and "rows_per_s" in v: print("  ", k, f"{v['rows_per_s']/1e9:.1f} Grows/s", v["roofline"]["frac"], v.get("verified_vs_oracle"))
        elif isinstance(v, dict) and "error" in v: print("  ", k, v)
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*" | sort -u > $OUT/avail_counters.txt; wc -l $OUT/avail_counters.txt; head -60 $OUT/avail_counters.txt | tr '\n' ' '
Q="python $R/tools/prof_query.py"
export DFX_NO_TORCH=1
pmc() { name=$1; shift; timeout 120 rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $Q headline 268435456 1 batch=134217728 > $OUT/$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
pmc ic2 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for d in ("ic1", "ic2"):
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "partition" not in k: continue
            k = ("pass1 " if "ring" in k else "pass2 ") + r["Counter_Name"]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in sorted(agg): print(d, k, agg[k] / cnt[k], "per dispatch,", cnt[k], "dispatches")
PY
find . -name "*counter_collection*.csv" -size +500k -delete 2>/dev/null
du -sh $OUT
