#!/bin/bash
# round 3: where the run-time decoded pass 1 (FastPolicy) and the interpreter lose their time: row groups per trip (code size
# of the loop) and instruction-cache counters
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c11; mkdir -p $OUT; export TMPDIR=/tmp DFX_NO_TORCH=1
cd $R
V=$R/datafusion_archive_amd/lib/variants
for lib in "" $V/libdfx_u2.so $V/libdfx_u1.so; do
  echo "== lib ${lib:-default}"
  DFX_LIB=$lib timeout 120 python tools/prof_query.py threeterm 1e9 3 batch=134217728 2>&1 | tail -2
done
cd /tmp
Q="python $R/tools/prof_query.py"
pmc() { name=$1; wl=$2; shift; shift; timeout 180 rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $Q $wl 268435456 1 batch=134217728 $EXTRA > $OUT/$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc ic_fast threeterm SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
pmc sq_fast threeterm SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM
EXTRA="scan.fast=0"
pmc ic_interp headline SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
pmc sq_interp headline SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM
EXTRA=""
pmc ic_static headline SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for d in ("ic_fast", "sq_fast", "ic_interp", "sq_interp", "ic_static"):
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_partition_ring" not in k and "k_partition_ws" not in k: continue
            k = r["Counter_Name"]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
        print(d, {k: round(agg[k] / cnt[k]) for k in sorted(agg)}, "dispatches", max(cnt.values()) if cnt else 0)
PY
find . -name "*counter_collection*.csv" -size +300k -delete 2>/dev/null; find . -name "*.db" -delete 2>/dev/null
du -sh $OUT
