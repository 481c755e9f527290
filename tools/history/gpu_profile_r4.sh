#!/bin/bash
# Round-4 evidence (one gpurun call): rocprofv3 kernel stats of the bench command; HBM traffic counters (FETCH_SIZE / WRITE_SIZE
# in separate passes, as the guide prescribes) and SQ instruction counters of pass 1 for the headline through its compile-time
# signature and through the scan plan (scan.plan = 2), and for the three-term query through the scan plan and through round 3's
# run-time decoded shape (scan.plan = 0) and the interpreter: scalar / vector instructions per 64-row group, before and after.
# Summaries land in gpurun_out/prof_r4/ -- copied to profiles/r04_*.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o out -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/stats/out_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
export DFX_NO_TORCH=1
Q="python $R/tools/prof_query.py"
pmc() { name=$1; wl=$2; opts=$3; shift; shift; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $Q $wl 268435456 1 batch=134217728 $opts > /dev/null 2>&1; }
pmc fetch_headline headline "" FETCH_SIZE
pmc write_headline headline "" WRITE_SIZE
pmc fetch_headlineplan headline "scan.plan=2" FETCH_SIZE
pmc write_headlineplan headline "scan.plan=2" WRITE_SIZE
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"
pmc sq_headline headline "" $SQ2
pmc sq_headlineplan headline "scan.plan=2" $SQ2
pmc sq_threeterm threeterm "" $SQ2
pmc sq_threetermfast threeterm "scan.plan=0" $SQ2
pmc sq_headlineinterp headline "scan.fast=0" $SQ2
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
ROWS = 2.0 * 268435456  # rows every run pushes through the kernels (one warm-up + one timed pass)
for d in sorted(glob.glob("fetch_*") + glob.glob("write_*") + glob.glob("sq_*")):
    wl = d.split("_", 1)[1]
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "partition" not in k:
                continue
            k = ("pass1 " if ("_ws" in k or "ring" in k or "k_partition<" in k or "sorted" in k) else "pass2 ") + k[:90]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            for c, x in v.items():
                e = res[wl + " | " + k]
                e[c + "_per_dispatch"] = x / cnt[(k, c)]
                e["dispatches"] = cnt[(k, c)]
                e["rows_per_dispatch"] = ROWS / cnt[(k, c)]
                if c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
                    e[c + "_per_64_row_group"] = x / cnt[(k, c)] / (ROWS / cnt[(k, c)] / 64.0)
json.dump(res, open("partition_counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()): print(k, {a: round(b, 1) for a, b in v.items()})
PY
head -12 bench_kernel_stats.csv | cut -c1-220
tail -c 1500 bench_under_rocprof.json | head -c 600; echo
rm -rf stats*/out_kernel_trace.csv */*/*.csv.gz 2>/dev/null
find . -name "*counter_collection*.csv" -size +2000k -delete 2>/dev/null
find . -name "*kernel_trace*.csv" -size +2000k -delete 2>/dev/null
du -sh $OUT
