#!/bin/bash
# round 5, call 4: tile-bucketed pass 1 with batched copy-out reads
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c4; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_NO_TORCH=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "tile_sorted" > $OUT/pytest_tile.log 2>&1; echo "tile tests rc=$?"; tail -n 5 $OUT/pytest_tile.log | cut -c1-400
for opt in agg.pass1_tile=0 "agg.pass1_tile=1" "agg.pass1_tile=0" "agg.pass1_tile=1"; do
  echo "== cfg3 dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_dense.txt
for opt in agg.pass1_tile=0 agg.pass1_tile=1; do
  echo "== wide keys dense $opt"; timeout 300 python tools/kprobe.py 1073741824 1e6 0 wide $opt 2>&1 | tail -n 3 | cut -c1-400
done | tee $OUT/kprobe_wide.txt
cd /tmp
i=0
for opt in "agg.pass1_tile=1"; do
i=$((i+1))
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/sq_$i -o out -- python $R/tools/prof_query.py cfg3 268435456 1 $opt > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("/root/repo/gpurun_out/r5c4/*/**/*counter_collection*.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:110]
        if "partition" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print(f.split("r5c4/")[1].split("/")[0])
    for k, v in agg.items():
        print("  ", k, {c: round(x / cnt[(k, c)] / (67108864 / 64.0), 3) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
find $OUT -name "*counter_collection*.csv" -size +2000k -delete
