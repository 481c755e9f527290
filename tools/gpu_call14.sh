#!/bin/bash
# round 3: A/B of non-temporal chunk stores in the wave-specialised pass 1 (variant library built by tools/build_variant.py)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp DFX_NO_TORCH=1 KPROBE_BATCH_LOG2=27
cd $R
V=$R/datafusion_archive_amd/lib/variants
for i in 1 2 3; do for lib in "" $V/libdfx_nt.so; do
  echo "== lib ${lib:-default}"; DFX_LIB=$lib timeout 120 python tools/kprobe.py 1e9 1e6 1 2>&1 | tail -1
done; done
