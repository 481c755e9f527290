#!/bin/bash
# What the driver runs at round end, plus the two runs it cannot do for us: GPU test suite, smoke, the bench line with the
# driver's flags, one 10^10-row single-GPU run (north_star's target size) and a 2-rank dry run of the multi-GPU code path
# on this one GPU (gloo, host-driven exchange: plumbing only).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu_round_check.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 800 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_default.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_default.json"))
    print("verified_vs_oracle:", d["extra"].get("verified_vs_oracle"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "traffic/algo", d["roofline"].get("traffic_over_algorithmic"), "cold", d["extra"].get("cold_first_step_ms"))
    print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
    print(d["extra"]["kernels"])
    for k, v in d["extra"].items():
        if isinstance(v, dict) and "rows_per_s" in v: print("  ", k, f"{v['rows_per_s']/1e9:.1f} Grows/s", v["roofline"]["frac"])
        elif isinstance(v, dict) and "error" in v: print("  ", k, v)
except Exception as e:
    print("no bench line:", e)
PY
timeout 600 python bench.py --rows 1e10 --steps 3 --warmup 1 --prewarm-steps 3 --no-extras --cpu-sample-rows 1e8 > gpurun_out/bench_1e10.json 2> gpurun_out/bench_1e10.err; echo "bench 1e10 rc=$?"; tail -2 gpurun_out/bench_1e10.err; cut -c1-700 gpurun_out/bench_1e10.json
DFX_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rows 1e8 --steps 2 --warmup 1 --prewarm-steps 2 > gpurun_out/bench_2rank_dryrun.json 2> gpurun_out/bench_2rank_dryrun.err; echo "2-rank dry run rc=$?"; tail -3 gpurun_out/bench_2rank_dryrun.err; cut -c1-900 gpurun_out/bench_2rank_dryrun.json
