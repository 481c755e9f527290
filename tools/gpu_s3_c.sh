#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/tl; NOPROF=1 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $R/gpurun_out/tl -o out -- python $R/tools/prof_query.py headline 1000000000 3 > /dev/null 2>&1
cd $R; python tools/timeline.py gpurun_out/tl 64 | awk 'NR<=16 || NR>=50'
rm -rf gpurun_out/tl
