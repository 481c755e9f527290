#!/bin/bash
# rocprofv3 kernel stats of the bench command alone (the first step of tools/gpu_profile_r3.sh), for the final tree
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r3b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o out -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/stats/out_kernel_stats.csv $OUT/bench_kernel_stats.csv
rm -f $OUT/stats/out_kernel_trace.csv
head -5 $OUT/bench_kernel_stats.csv | cut -c1-200; cut -c1-300 $OUT/bench_under_rocprof.json
