#!/bin/bash
# round 5, call 12: the interpreter specialised for null-free batches -- fuzz suite, the headline through it, SQ counters
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c12; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_NO_TORCH=1
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout 600 -x > $OUT/pytest_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -n 4 $OUT/pytest_fuzz.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "filter or project or ungrouped or grouped_aggregates" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 4 $OUT/pytest_sel.log | cut -c1-300
for i in 1 2; do timeout 300 python tools/qprobe.py 1073741824 interp,headline 2>&1 | tail -n 4 | cut -c1-250; done | tee $OUT/qprobe_interp.txt
cd /tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq_interp -o out -- python $R/tools/prof_query.py headline 268435456 1 batch=134217728 scan.fast=0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("/root/repo/gpurun_out/r5c12/sq_interp/**/*counter_collection*.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:110]
        if "partition" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print("  ", k, {c: round(x / cnt[(k, c)] / (134217728 / 64.0), 1) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
find $OUT -name "*counter_collection*.csv" -size +2000k -delete
