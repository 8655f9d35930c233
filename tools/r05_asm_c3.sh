#!/bin/bash
# tools only (round 5): the generic assembly backward at the C3 geometry (q = 11): runtime-q against the compile-time instantiation, ablations
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export BWD_GEOM="300,512,10,10" PROBE=assemble_bwd_probe.py PROBE_LINES=2
tools/assemble_variants.sh "q11:" "runtime_q:-DBWD_NO_Q11" "q11_nopass:-DBWD_ABLATE=1" "q11_nodp1:-DBWD_ABLATE=2" "q11_noT:-DBWD_ABLATE=4" "q11_noG:-DBWD_ABLATE=8" "q11_all:-DBWD_ABLATE=15" 2>&1 | grep -v amdgpu.ids | sed 's#/tmp/asm_build_[A-Za-z0-9]*/##'
