#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_csv.py -q -x --timeout 600 2>&1 | tail -n 3
cd /tmp; rm -rf $R/gpurun_out/csvprof
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/csvprof -o out -- python $R/tools/csv_bench.py 1024 4194304 2>&1 | grep "MB,"
cd $R; head -7 gpurun_out/csvprof/out_kernel_stats.csv | cut -c1-160; cp gpurun_out/csvprof/out_kernel_stats.csv gpurun_out/csv_kernel_stats.csv; rm -rf gpurun_out/csvprof
