#!/bin/bash
# round 3, closing run (one gpurun call, ~13 min): smoke, the whole GPU suite, the bench line with the driver's flags, then the
# evidence of tools/gpu_profile_r3.sh (rocprofv3 kernel stats of the bench command, counters, launch-size decomposition)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3final; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest.log
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall ${SECONDS}s"; cut -c1-400 $OUT/bench.json
bash $R/tools/gpu_profile_r3.sh > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -n 30 $OUT/profile.log | cut -c1-300
