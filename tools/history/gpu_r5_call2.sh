#!/bin/bash
# round 5, call 2: the tile-BUCKETED pass 1 (parity, per-kernel times, SQ counters), small routing windows (Infinity Cache)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c2; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "tile_sorted" > $OUT/pytest_tile.log 2>&1; echo "tile tests rc=$?"; tail -n 12 $OUT/pytest_tile.log | cut -c1-400
export DFX_NO_TORCH=1
for opt in agg.pass1_tile=0 agg.pass1_tile=1 "agg.pass1_tile=1 agg.partition_defer=2"; do
  echo "== cfg3 dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_dense.txt
for opt in agg.pass1_tile=0 agg.pass1_tile=1; do
  echo "== wide keys dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 wide $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_wide.txt
echo "== dense, small launches (routed rows of a window fit the Infinity Cache)"
for opt in "agg.partition_split_rows=4194304" "agg.partition_split_rows=8388608" "agg.partition_split_rows=16777216"; do
  timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_windows.txt
echo "== selectivity 0.5 / 0.8 of the headline (tile on = default when dense)"
for hi in 716.8 1024.0; do for opt in agg.pass1_tile=0 agg.pass1_tile=1; do
  timeout 300 python tools/kprobe.py 1073741824 1e6 1 lo=204.8 hi=$hi $opt 2>&1 | tail -n 3 | cut -c1-400
done; done | tee $OUT/kprobe_sel.txt
cd /tmp
for opt in agg.pass1_tile=0 agg.pass1_tile=1; do
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/sq_$opt -o out -- python $R/tools/prof_query.py cfg3 268435456 1 $opt > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("/root/repo/gpurun_out/r5c2/sq_*/**/*counter_collection*.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:110]
        if "partition" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print(f.split("/")[-4] if "sq_" in f else f)
    for k, v in agg.items():
        print("  ", k, {c: round(x / cnt[(k, c)] / (67108864 / 64.0), 2) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
find $OUT -name "*counter_collection*.csv" -size +2000k -delete
