#!/bin/bash
# tools only (round 5): assembly forward / backward at the C5 geometry (q = 6, d = 50): pair kernels with KSM = 16 against the generic kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_ops.py -x -q -k "kernel" 2>&1 | tail -3
export BWD_GEOM="1024,512,50,5" FWD_GEOM="1024,512,50,5"
python tools/assemble_bwd_probe.py 2>&1 | grep -v amdgpu.ids
python tools/assemble_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
