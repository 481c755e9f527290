#!/bin/bash
# round 3: 2- and 3-rank dry runs of bench.py on ONE GPU -- the multi-rank code path (config 4 row ranges, the config-5 leg)
# with the library's own exchange between the ranks over the host-staged RCCL stand-in (plumbing only: the numbers mean nothing)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c8; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_BENCH_SHARED_GPU=1 DFX_RCCL_LIB=$R/tests/native/librccl_stub.so
for n in 2 3; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --rows 1e8 --steps 2 --warmup 1 > $OUT/bench_${n}rank.json 2> $OUT/bench_${n}rank.err; echo "$n-rank dry run rc=$?"; tail -4 $OUT/bench_${n}rank.err; cut -c1-1500 $OUT/bench_${n}rank.json
done
unset DFX_RCCL_LIB
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rows 1e8 --steps 2 --warmup 1 > $OUT/bench_2rank_host.json 2> $OUT/bench_2rank_host.err; echo "2-rank host-exchange dry run rc=$?"; tail -3 $OUT/bench_2rank_host.err; cut -c1-700 $OUT/bench_2rank_host.json
