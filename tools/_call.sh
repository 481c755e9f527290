mkdir -p gpurun_out; export TMPDIR=/tmp
for wl in neighbour headline; do for o in "" "scan.fast=0"; do echo "== $wl $o"; timeout 200 python tools/prof_query.py $wl 1e9 3 $o 2>&1 | grep -v amdgpu.ids | tail -2; done; done
