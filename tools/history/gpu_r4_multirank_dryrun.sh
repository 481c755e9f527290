#!/bin/bash
# round 4: 2- and 3-rank dry runs of bench.py on ONE GPU -- the multi-rank code path of the final tree (config 4's row ranges, the
# config-5 leg, config.rccl_ranks) with the library's own exchange between the ranks over the host-staged RCCL stand-in; plumbing
# only, the numbers mean nothing.  Then the new split-scan robustness test (added after the closing suite).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4dry; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_plan.py -m gpu -q --timeout 600 -x -k "skew_growth" > $OUT/pytest_split.log 2>&1; echo "split robustness rc=$?"; tail -n 3 $OUT/pytest_split.log | cut -c1-300
export DFX_BENCH_SHARED_GPU=1 DFX_RCCL_LIB=$R/tests/native/librccl_stub.so
for n in 2 3; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --rows 1e8 --steps 2 --warmup 1 > $OUT/bench_${n}rank.json 2> $OUT/bench_${n}rank.err; echo "$n-rank dry run rc=$?"; tail -3 $OUT/bench_${n}rank.err | cut -c1-300; cut -c1-1200 $OUT/bench_${n}rank.json
done
