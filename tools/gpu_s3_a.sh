#!/bin/bash
# session-3 GPU call A: new few-group kernel (parity + speed), batch-size sweep of the headline, full GPU suite, default bench
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new tests =="; timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "fewgroup or q1_shape or grouped_aggregates" 2>&1 | tail -n 8
echo "== q1 =="; for fg in 1 0; do timeout 300 python tools/prof_query.py q1 268435456 3 agg.fewgroup=$fg 2>&1 | tail -1; done
echo "== headline batch sweep =="
for br in 67108864 134217728 268435456; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --batch-rows $br > gpurun_out/bench_br$br.json 2> gpurun_out/bench_br$br.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_br$br.json")); print($br, "value %.4g rows/s  ms/step %.3f  frac %.3f  verified %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["extra"]["verified_sum_of_group_sums_equals_ungrouped_sum"]), {k:(v["launches"], v["total_ms"]) for k,v in d["extra"]["kernels"].items()})
except Exception as e: print($br, "failed", e)
PY
done
echo "== pytest gpu (full) =="; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu.log
echo "== bench default =="; timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json | cut -c1-1500
