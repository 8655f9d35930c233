#!/bin/bash
# round 6: canonical kernels at HEAD: tests, then the probe over variants given as arguments ("name:-DFLAG=..,-DOTHER=..")
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "${CANON_K:-canon}" > $O/tests3.txt 2>&1 || { tail -30 $O/tests3.txt; exit 1; }
tail -2 $O/tests3.txt
export CANON_FILL=0
for rep in 1 2; do
  echo "== HEAD" | tee -a $O/head.txt; python tools/canon_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/head.txt
  [ $# -gt 0 ] && CANON_GEOM="${CANON_GEOM:-500,4096,20,5}" PROBE=canon_probe.py PROBE_LINES=1 tools/assemble_variants.sh "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/head.txt
done
