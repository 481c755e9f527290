#!/bin/bash
# Round-3 evidence (one gpurun call, ~6 min): rocprofv3 kernel stats of the bench command and of the Q1 / config-2 workloads,
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes) and SQ counters of the partition
# kernels (pass 1 wave-specialised, pass 2) and of the single-pass filter, and the launch-size decomposition of pass 1
# (2^25 / 2^26 / 2^27 rows per launch, headline predicate and a predicate that rejects every row, both pass-1 flavours).
# Summaries land in gpurun_out/prof_r3/ -- copy them to profiles/r03_*.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o out -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/stats/out_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
export DFX_NO_TORCH=1
Q="python $R/tools/prof_query.py"
FP="python $R/tools/filter_probe.py 536870912"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_q1 -o out -- $Q q1 268435456 3 batch=134217728 > $OUT/q1.log 2>&1
cp $OUT/stats_q1/out_kernel_stats.csv $OUT/q1_kernel_stats.csv 2>/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_f -o out -- $FP > $OUT/filter.log 2>&1
cp $OUT/stats_f/out_kernel_stats.csv $OUT/filter_kernel_stats.csv 2>/dev/null
pmc() { name=$1; wl=$2; shift; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $Q $wl 268435456 1 batch=134217728 > /dev/null 2>&1; }
pmcf() { name=$1; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $FP > /dev/null 2>&1; }
pmc fetch_headline headline FETCH_SIZE
pmc write_headline headline WRITE_SIZE
pmc fetch_cfg3 cfg3 FETCH_SIZE
pmc write_cfg3 cfg3 WRITE_SIZE
pmcf fetch_filter FETCH_SIZE
pmcf write_filter WRITE_SIZE
pmc sq1_headline headline SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc sq2_headline headline SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pmcf sq1_filter SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmcf sq2_filter SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
ROWS = {"headline": 2.0 * 268435456, "cfg3": 2.0 * 268435456, "filter": 9.0 * 536870912}  # rows every run pushes through the kernels (warm-up + timed passes)
for d in sorted(glob.glob("*_headline") + glob.glob("*_cfg3") + glob.glob("*_filter")):
    wl = d.split("_", 1)[1]
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "partition" in k:
                k = ("pass1 " if ("_ws" in k or "ring" in k or "k_partition<" in k or "sorted" in k) else "pass2 ") + k[:80]
            elif "filter_fused" in k:
                k = "filter " + k[:80]
            else:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            for c, x in v.items():
                res[wl + " | " + k][c + "_per_dispatch"] = x / cnt[(k, c)]
                res[wl + " | " + k]["dispatches"] = cnt[(k, c)]
                res[wl + " | " + k]["rows_per_dispatch"] = ROWS[wl] / cnt[(k, c)]
json.dump(res, open("partition_counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()): print(k, {a: round(b, 1) for a, b in v.items()})
PY
head -14 bench_kernel_stats.csv | cut -c1-220
head -6 q1_kernel_stats.csv | cut -c1-220
head -5 filter_kernel_stats.csv | cut -c1-220
tail -c 1500 bench_under_rocprof.json | head -c 400; echo
# launch-size decomposition of pass 1: fixed cost + cost per row, with and without routed rows, both flavours
for ws in 8 0; do for lg in 25 26 27; do
  echo "== decomposition ws=$ws rows_per_launch=2^$lg headline"; KPROBE_BATCH_LOG2=$lg timeout 100 python $R/tools/kprobe.py 1e9 1e6 1 agg.pass1_ws=$ws agg.partition_defer=1 agg.merge_scan_batches=0 agg.strategy=3 agg.narrow_keys=1 2>&1 | tail -1 | cut -c1-300
  echo "== decomposition ws=$ws rows_per_launch=2^$lg all rows rejected"; KPROBE_BATCH_LOG2=$lg timeout 100 python $R/tools/kprobe.py 1e9 1e6 1 lo=2000 hi=3000 agg.pass1_ws=$ws agg.partition_defer=1 agg.merge_scan_batches=0 agg.strategy=3 agg.narrow_keys=1 2>&1 | tail -1 | cut -c1-300
done; done > $OUT/kprobe_decomposition.txt 2>&1
cat $OUT/kprobe_decomposition.txt
rm -rf stats*/out_kernel_trace.csv */*/*.csv.gz 2>/dev/null
find . -name "*counter_collection*.csv" -size +2000k -delete 2>/dev/null
find . -name "*kernel_trace*.csv" -size +2000k -delete 2>/dev/null
du -sh $OUT
