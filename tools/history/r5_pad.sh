cd $GRAFT_REPO_ROOT
DFX_NO_TORCH=1 timeout 300 python tools/pad_probe.py 1e9 0 2>&1 | grep -v amdgpu.ids
DFX_NO_TORCH=1 timeout 300 python tools/pad_probe.py 1e9 1 2>&1 | grep -v amdgpu.ids
