#!/bin/bash
# The round's evidence on ONE gpurun call each (tools/gpu.sh <timeout> tools/gpu_round.sh <stage> [round tag]):
#   suite    the whole GPU test suite (what the driver runs at round end)
#   bench    bench.py with the driver's flags -> gpurun_out/<tag>/bench.json + a one-screen summary
#   profile  rocprofv3 --kernel-trace --stats of the bench command; FETCH_SIZE / WRITE_SIZE (separate passes) and SQ counters of
#            pass 1 / pass 2 for the headline and for config 3 -> partition_counters.json; kernel stats of config 3
#   csv      the CSV source: its GPU tests, tools/csv_bench.py (1 GB of numeric text) and the rocprofv3 kernel summary of that run
#   csvpmc   SQ counters of the CSV kernels (256 MB of text; per 64-record tile for k_csv_parse) -> csv_counters.txt
#   pairpmc  FETCH_SIZE / WRITE_SIZE of pass 1 / pass 2 of the pair scan and of the planes (tools/prof_query.py diffop / avgmax / neighbour)
#   dry8     bench.py --gpus 8 as eight processes on this ONE GPU over the host-staged RCCL stand-in (plumbing only), and the
#            same with DFX_RCCL_LIB pointing at a missing file (must exit non-zero)
# Summaries land in gpurun_out/<tag>/ -- what is cited is copied to profiles/ by hand.
STAGES=${1:-bench}; TAG=${2:-r06}   # (several stages: comma-separated, run in order)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for STAGE in ${STAGES//,/ }; do
cd $R
echo "==== stage $STAGE"
case $STAGE in
csv)
  timeout 900 python -m pytest tests/test_gpu_csv.py -m gpu -q --timeout 600 > $OUT/pytest_csv.log 2>&1; echo "csv tests rc=$?"; tail -n 12 $OUT/pytest_csv.log | cut -c1-400
  timeout 600 python tools/csv_bench.py 1024 > $OUT/csv_bench.txt 2>&1; echo "csv bench rc=$?"; tail -n 4 $OUT/csv_bench.txt | cut -c1-400
  (cd /tmp && DFX_NO_TORCH=1 timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_csvstr -o out -- python $R/tools/csv_bench.py 1024 4194304 str > $OUT/csv_bench_str.txt 2>&1); grep "MB," $OUT/csv_bench_str.txt | tail -n 2 | cut -c1-400
  cp $OUT/stats_csvstr/out_kernel_stats.csv $OUT/csv_str_kernel_stats.csv 2>/dev/null; grep -m3 "csv" $OUT/csv_str_kernel_stats.csv | cut -c1-200
  (cd /tmp && DFX_NO_TORCH=1 timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_csv -o out -- python $R/tools/csv_bench.py 1024 > /dev/null 2>&1)
  cp $OUT/stats_csv/out_kernel_stats.csv $OUT/csv_kernel_stats.csv 2>/dev/null; head -n 12 $OUT/csv_kernel_stats.csv | cut -c1-200
  ;;
csvpmc)
  # SQ counters of the CSV kernels on 256 MB of text (per 64-record tile for k_csv_parse)
  cd /tmp; export DFX_NO_TORCH=1
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
    n=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout 600 rocprofv3 --output-format csv --pmc $set -d $OUT/csvpmc_$n -o out -- python $R/tools/csv_bench.py 256 > /dev/null 2>&1
  done
  cd $OUT
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("csvpmc_*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "csv" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
tiles = 3 * (4645000 + 63) // 64  # three passes of csv_bench over 4.645e6 records
with open("csv_counters.txt", "w") as out:
    for k, v in sorted(agg.items()):
        line = k[:60] + "  " + "  ".join(f"{c}={x:.3g}" + (f" ({x / tiles:.1f}/tile)" if "k_csv_parse" in k else "") for c, x in sorted(v.items()))
        print(line[:1500]); out.write(line + "\n")
PY
  find . -name "*counter_collection*.csv" -size +2000k -delete 2>/dev/null
  ;;
suite)
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -n 8 $OUT/pytest_gpu.log | cut -c1-300
  ;;
bench)
  SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_full.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench took $SECONDS s"; echo "bench rc=$?  last stdout line: $(tail -n 1 $OUT/bench.json | wc -c) bytes"; tail -n 3 $OUT/bench.err | cut -c1-300
  python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step", round(d["ms_per_step"], 4), "end to end", r.get("end_to_end_frac"), "dominant kernel", r["frac"], "cold first step ms", r.get("cold_first_step_ms"))
print("cpu_baseline", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), "verified", (d["extra"].get("verified_vs_oracle") or {}).get("ok"))
for k, v in d["extra"].items():
    if isinstance(v, dict) and ("roofline" in v or "error" in v):
        rf = v.get("roofline") or {}
        vo = v.get("verified_vs_oracle")
        print(f"  {k:38s} frac {rf.get('frac')}  ms {v.get('ms') and round(v['ms'], 2)}  verified {vo.get('ok') if isinstance(vo, dict) else vo}  {v.get('error', '')}")
PY
  ;;
profile)
  cd /tmp
  BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --full-out $OUT/bench_under_rocprof_full.json"
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o out -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null
  cp $OUT/stats/out_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
  export DFX_NO_TORCH=1
  Q="python $R/tools/prof_query.py"
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_cfg3 -o out -- $Q cfg3 1073741824 3 > /dev/null 2>&1
  cp $OUT/stats_cfg3/out_kernel_stats.csv $OUT/cfg3_kernel_stats.csv 2>/dev/null
  pmc() { name=$1; wl=$2; opts=$3; shift; shift; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/$name -o out -- $Q $wl 268435456 1 batch=134217728 $opts > /dev/null 2>&1; }
  SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
  for wl in headline cfg3; do
    pmc fetch_$wl $wl "" FETCH_SIZE
    pmc write_$wl $wl "" WRITE_SIZE
    pmc sq_$wl $wl "" $SQ2
  done
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
            k = ("pass2 " if "partition_agg" in k else "pass1 ") + k[:90]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            for c, x in v.items():
                e = res[wl + " | " + k]
                e[c + "_per_dispatch"] = x / cnt[(k, c)]
                e["dispatches"] = cnt[(k, c)]
                e["rows_per_dispatch"] = ROWS / cnt[(k, c)]
                if c.startswith("SQ_"):
                    e[c + "_per_64_row_group"] = x / cnt[(k, c)] / (ROWS / cnt[(k, c)] / 64.0)
json.dump(res, open("partition_counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()): print(k[:70], {a: round(b, 1) for a, b in v.items() if "per_64" in a or a in ("FETCH_SIZE_per_dispatch", "WRITE_SIZE_per_dispatch", "dispatches")})
PY
  head -12 bench_kernel_stats.csv | cut -c1-220
  head -8 cfg3_kernel_stats.csv | cut -c1-220
  rm -rf stats*/out_kernel_trace.csv */*/*.csv.gz 2>/dev/null
  find . -name "*counter_collection*.csv" -size +2000k -delete 2>/dev/null
  find . -name "*kernel_trace*.csv" -size +2000k -delete 2>/dev/null
  ;;
pairpmc)
  # FETCH_SIZE / WRITE_SIZE (separate passes) of the pair scan (two aggregates of different columns), of three accumulators over two
  # columns and of the planes of a shared operand: 2^28 rows, one warm-up + one timed pass -> pair_counters.json
  cd /tmp; export DFX_NO_TORCH=1
  Q="python $R/tools/prof_query.py"
  for wl in diffop avgmax neighbour; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --output-format csv --pmc $c -d $OUT/pmc_${wl}_$c -o out -- $Q $wl 268435456 1 batch=134217728 > /dev/null 2>&1
    done
  done
  cd $OUT
  python3 - <<'PY'
import csv, glob, collections, json
res = {}
for d in sorted(glob.glob("pmc_*")):
    _, wl, c = d.split("_", 2)
    for f in glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "partition" not in k or r["Counter_Name"] != c:
                continue
            k = ("pass2 " if "partition_agg" in k else "pass1 ") + k[:100]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k, x in agg.items():
            e = res.setdefault(wl + " | " + k, {})
            e[c + "_KB_per_dispatch"] = x / cnt[k]
            e["dispatches"] = cnt[k]
json.dump(res, open("pair_counters.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()): print(k[:120], {a: round(b, 1) for a, b in v.items()})
PY
  find . -name "*counter_collection*.csv" -size +2000k -delete 2>/dev/null
  ;;
dry8)
  export DFX_BENCH_SHARED_GPU=1 DFX_RCCL_LIB=$R/tests/native/librccl_stub.so
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --rows 5e7 --steps 2 --warmup 1 --full-out $OUT/bench_8rank_full.json > $OUT/bench_8rank.json 2> $OUT/bench_8rank.err; echo "8-rank dry run rc=$?"; tail -2 $OUT/bench_8rank.err | cut -c1-300
  python - $OUT/bench_8rank_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "rccl_ranks", d["config"]["rccl_ranks"], "exchange", d["config"]["exchange"][:60])
print("phases_ms", json.dumps(d["extra"].get("phases_ms")))
for k in ("cfg4_as_written", "cfg5_q1_shape"):
    v = d["extra"].get(k, {})
    print(k, "frac/GPU", (v.get("roofline") or {}).get("frac"), "phases", json.dumps(v.get("phases_ms")), "groups ok", v.get("every_group_emitted_once"))
print("sum-of-sums check", d["extra"]["verified_sum_of_group_sums_equals_ungrouped_sum"])
PY
  export DFX_RCCL_LIB=$R/tests/native/no_such_librccl.so
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 8 --rows 2e7 --steps 1 --warmup 0 > $OUT/bench_8rank_norccl.json 2> $OUT/bench_8rank_norccl.err; echo "8-rank run without a loadable RCCL rc=$? (must be non-zero)"; grep -m1 "refusing" $OUT/bench_8rank_norccl.err | cut -c1-300
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 8 --rows 2e7 --steps 1 --warmup 0 --allow-host-exchange > $OUT/bench_8rank_hostexchange.json 2> $OUT/bench_8rank_hostexchange.err; echo "... with --allow-host-exchange rc=$?"; cut -c1-400 $OUT/bench_8rank_hostexchange.json | head -1
  ;;
esac
done
