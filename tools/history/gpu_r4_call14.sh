#!/bin/bash
# round 4, call 14: scanner loop with scalar lane masks (tests, instruction counts per row group, per-kernel times)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c14; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 > $OUT/pytest_a.log 2>&1; echo "plan+fuzz rc=$?"; tail -n 4 $OUT/pytest_a.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "partition or narrow or wide or skew or shared or resident" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 4 $OUT/pytest_sel.log | cut -c1-300
export DFX_NO_TORCH=1
for i in 1 2; do timeout 300 python tools/qprobe.py 1073741824 headline,plan,three,int32key,nullv,reject 2>&1 | tail -n 6 | cut -c1-200; done | tee $OUT/qprobe.txt
cd /tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/sq_headline -o out -- python $R/tools/prof_query.py headline 268435456 1 batch=134217728 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("/root/repo/gpurun_out/r4c14/sq_headline/**/*counter_collection*.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:100]
        if "partition_ws" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k, {c: round(x / cnt[(k, c)] / (134217728 / 64.0), 2) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
find $OUT -name "*counter_collection*.csv" -size +2000k -delete
