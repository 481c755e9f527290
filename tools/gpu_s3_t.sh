#!/bin/bash
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/datafusion_archive_amd/lib
for i in 1 2; do for v in libdfx_hip.so libdfx_hip_d2.so libdfx_hip_d3.so; do echo "-- $v"; for wl in headline cfg3; do DFX_LIB=$L/$v timeout 300 python tools/prof_query.py $wl 1000000000 3 2>&1 | tail -2 | head -1; done; done; done
