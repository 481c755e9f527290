#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
rm -rf $R/gpurun_out/pmc; rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/pmc -o out -- python $R/tools/prof_query.py cfg3 268435456 1  > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/pmc/**/*counter_collection*.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "partition_agg" in k or "partition_ring" in k: print(k, {c: (round(x / cnt[(k, c)]), cnt[(k, c)]) for c, x in v.items()})
PY
done
rm -rf $R/gpurun_out/pmc
