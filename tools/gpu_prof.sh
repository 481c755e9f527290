#!/bin/bash
# rocprofv3 kernel stats + PMC passes for one workload.  usage: gpu_prof.sh <workload> <rows> [opts...]
WL=${1:-headline}; ROWS=${2:-268435456}; shift; shift
mkdir -p gpurun_out/prof_$WL
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/prof_query.py $WL $ROWS 3 "$@" > $R/gpurun_out/prof_$WL/plain.txt 2>&1; cat $R/gpurun_out/prof_$WL/plain.txt | tail -2
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/prof_$WL/stats -o out -- python $R/tools/prof_query.py $WL $ROWS 3 "$@" > /dev/null 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/prof_$WL/pmc1 -o out -- python $R/tools/prof_query.py $WL $ROWS 1 "$@" > /dev/null 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_$WL/pmc2 -o out -- python $R/tools/prof_query.py $WL $ROWS 1 "$@" > /dev/null 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/prof_$WL/pmc3 -o out -- python $R/tools/prof_query.py $WL $ROWS 1 "$@" > /dev/null 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_$WL/pmc4 -o out -- python $R/tools/prof_query.py $WL $ROWS 1 "$@" > /dev/null 2>&1
cd $R/gpurun_out/prof_$WL
find . -name "*kernel_stats*.csv" | head -1 | xargs -r head -12
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3","pmc4"):
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            if "hash_agg" in k or "reduce" in k or "predicate" in k:
                print(d, k, {c: (round(x,1), cnt[(k,c)]) for c,x in v.items()})
PY
