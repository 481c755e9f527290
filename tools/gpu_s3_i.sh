#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/gpurun_out/csvprof
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/csvprof -o out -- python $R/tools/csv_bench.py 1024 4194304 2>&1 | tail -2
cd $R; head -12 gpurun_out/csvprof/out_kernel_stats.csv | cut -c1-200; cp gpurun_out/csvprof/out_kernel_stats.csv gpurun_out/csv_kernel_stats.csv; rm -rf gpurun_out/csvprof
