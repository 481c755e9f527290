#!/bin/bash
# round 4, call 6: per-aggregate scans, 5..8 keys, host stream forms, exchange; probes; bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c6; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_plan.py tests/test_gpu_exchange_world2.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 > $OUT/pytest_a.log 2>&1; echo "plan+exchange+fuzz rc=$?"; tail -n 6 $OUT/pytest_a.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "five_to_eight or filter or host_batches or multi_column or accumulators or avg or shared or one_operand" > $OUT/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -n 6 $OUT/pytest_sel.log | cut -c1-300
python tools/qprobe.py 1e9 sum_min_w 2>&1 | tee $OUT/qprobe.txt | cut -c1-250
python tools/qprobe.py 1e9 sum_min_w agg.split_aggregates=0 2>&1 | tee -a $OUT/qprobe.txt | cut -c1-250
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c6/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["end_to_end_frac"], d["roofline"]["avg_launch_ms"])
for k,v in d["extra"].items():
    if isinstance(v,dict) and "roofline" in v and isinstance(v["roofline"],dict):
        vv=v.get("verified_vs_oracle") or {}
        print(f"{k:40s} {v.get('ms',0):9.2f} ms  frac {v['roofline']['frac']:.3f}  ok={vv.get('ok')} {str(vv.get('error',''))[:200]}")
    elif isinstance(v,dict) and "error" in v:
        print(k, "ERROR", v["error"])
PY
