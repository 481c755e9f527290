#!/bin/bash
# round 5, call 10: the container's CPU budget on the GPU box, then the bench line
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
bash $GRAFT_REPO_ROOT/tools/gpu_round.sh bench r05c
