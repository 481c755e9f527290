cd $GRAFT_REPO_ROOT
for r in 1 2; do DFX_NO_TORCH=1 timeout 400 python tools/placement_probe.py 1e9 2>&1 | grep -v amdgpu.ids; echo ----; done
