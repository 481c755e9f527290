#!/bin/bash
# rocprofv3 evidence for bench.py: kernel stats (same command as the bench) + HBM traffic counters in their own passes.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bench_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
$CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o out -- $CMD > $OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o out -- $CMD > /dev/null 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o out -- $CMD > /dev/null 2>&1
cd $OUT
cp stats/out_kernel_stats.csv kernel_stats.csv 2>/dev/null
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(f"{d}/*counter_collection*.csv"):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in agg:
            res.setdefault(k, {})[name + "_KB_total"] = agg[k]; res[k]["dispatches"] = cnt[k]
            res[k][name + "_KB_per_dispatch"] = agg[k] / cnt[k]
json.dump(res, open("hbm_counters.json", "w"), indent=1)
for k, v in res.items():
    if "partition" in k or "hash_agg" in k: print(k, v)
PY
head -8 kernel_stats.csv
rm -rf stats/out_kernel_trace.csv pmc_fetch pmc_write stats
