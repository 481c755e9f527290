#!/bin/bash
# round 4, call 10: which kernels the two scans of SUM(v), MIN(w) run, how long each takes, and what they execute per row group
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c10; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp DFX_NO_TORCH=1
Q="python $R/tools/prof_query.py"
for wl in diffop threecol; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_$wl -o out -- $Q $wl 1073741824 2 batch=134217728 > /dev/null 2>&1
  echo "== $wl"; head -6 $OUT/stats_$wl/out_kernel_stats.csv | cut -c1-200
  rm -f $OUT/stats_$wl/out_kernel_trace.csv
done
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $OUT/sq_threecol -o out -- $Q threecol 268435456 1 batch=134217728 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("/root/repo/gpurun_out/r4c10/sq_threecol/**/*counter_collection*.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:100]
        if "partition" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k, {c: round(x / cnt[(k, c)] / (134217728 / 64.0), 2) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
find $OUT -name "*counter_collection*.csv" -size +2000k -delete
