#!/bin/bash
# tools only (round 6): per-block stamps of the strips chain for a few variants of potrf.hip (defines in "$@", one run each)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06_potrf; mkdir -p $O
i=0
for defs in "$@"; do
  i=$((i+1))
  echo "=== variant $i: $defs"
  POTRF_DEFS="-DPOTRF_DEBUG_K=20 $defs" timeout -k 10 300 tools/potrf_clock.sh 2>&1 | grep -v amdgpu.ids | grep -v "per wave, end" | tee $O/clock_v$i.txt
done
