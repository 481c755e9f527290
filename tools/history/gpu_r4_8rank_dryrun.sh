#!/bin/bash
# round 4: an 8-rank dry run of bench.py on ONE GPU (the rank count of the driver's scaling run): the library's own exchange between
# eight processes over the host-staged RCCL stand-in; plumbing only, the numbers mean nothing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4dry8; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
export DFX_BENCH_SHARED_GPU=1 DFX_RCCL_LIB=$R/tests/native/librccl_stub.so
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --rows 5e7 --steps 2 --warmup 1 > $OUT/bench_8rank.json 2> $OUT/bench_8rank.err; echo "8-rank dry run rc=$?"; tail -2 $OUT/bench_8rank.err | cut -c1-300; cut -c1-900 $OUT/bench_8rank.json
