#!/bin/bash
# round 6: the both-sides one-hot assembly kernels: op tests, then the C3-geometry probe
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "canon2" > $O/tests6.txt 2>&1 || { tail -40 $O/tests6.txt; exit 1; }
tail -2 $O/tests6.txt
for rep in 1 2; do timeout -k 10 120 python tools/canon2_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/canon2.txt; done
