#!/bin/bash
# build here (hipcc cross-compiles), then run a script on the GPU box: tools/gpu.sh <timeout-seconds> <script> [args...]
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/gpu_sh_build.log 2>&1 || { tail -20 /tmp/gpu_sh_build.log; exit 1; }
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "bash $*"
