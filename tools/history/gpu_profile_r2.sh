#!/bin/bash
# Round-2 evidence (one gpurun call, ~5 min): rocprofv3 kernel stats of the bench command, the HBM traffic counters of the
# two partition kernels (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes), SQ counters of pass 1 /
# pass 2 (instruction mix, waits, LDS conflicts).  Summaries land in gpurun_out/prof_r2/ -- copy them to profiles/r02_*.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --prewarm-steps 30"
$BENCH > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o out -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/stats/out_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
Q="python $R/tools/prof_query.py"
pmc() { name=$1; wl=$2; shift; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $Q $wl 268435456 1 batch=134217728 > /dev/null 2>&1; }
pmc fetch_headline headline FETCH_SIZE
pmc write_headline headline WRITE_SIZE
pmc fetch_cfg3 cfg3 FETCH_SIZE
pmc write_cfg3 cfg3 WRITE_SIZE
pmc sq1_headline headline SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc sq2_headline headline SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pmc sq3_headline headline SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM
pmc sq1_cfg3 cfg3 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc sq2_cfg3 cfg3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json, os
res = collections.defaultdict(dict)
for d in sorted(glob.glob("*_headline") + glob.glob("*_cfg3")):
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "partition" not in k: continue
            k = ("pass1 " if "ring" in k or "k_partition<" in k or "sorted" in k else "pass2 ") + k[:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            for c, x in v.items():
                res[d.split("_", 1)[1] + " | " + k][c + "_per_dispatch"] = x / cnt[(k, c)]
                res[d.split("_", 1)[1] + " | " + k]["dispatches"] = cnt[(k, c)]
                # prof_query runs the query twice (warm-up + 1 iteration) over 2^28 rows: rows one dispatch scanned / aggregated
                res[d.split("_", 1)[1] + " | " + k]["rows_per_dispatch"] = 2.0 * 268435456 / cnt[(k, c)]
json.dump(res, open("partition_counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()): print(k, {a: round(b, 1) for a, b in v.items()})
PY
head -12 bench_kernel_stats.csv
tail -c 1500 bench_under_rocprof.json | head -c 600; echo
rm -rf stats/out_kernel_trace.csv */*/*.csv.gz 2>/dev/null
find . -name "*counter_collection*.csv" -size +2000k -delete 2>/dev/null
du -sh $OUT
