#!/bin/bash
# round 6: ablations / grid sizes of the canonical forward kernel, one box
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
export CANON_GEOM="500,4096,20,5" CANON_FILL=0
PROBE=canon_probe.py tools/assemble_variants.sh "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/variants.txt
