#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export BWD_GEOM="300,512,10,10" PROBE=assemble_bwd_probe.py PROBE_LINES=2
tools/assemble_variants.sh "w4:-DBWDS_WGS=1024" "w4_notransform:-DBWDS_WGS=1024,-DBWDS_ABL=1" "w4_nodp1:-DBWDS_WGS=1024,-DBWDS_ABL=2" "w4_noT:-DBWDS_WGS=1024,-DBWDS_ABL=4" "w4_noG:-DBWDS_WGS=1024,-DBWDS_ABL=8" "w4_none:-DBWDS_WGS=1024,-DBWDS_ABL=15" 2>&1 | grep -v amdgpu.ids | sed 's#/tmp/asm_build_[A-Za-z0-9]*/##' | grep "K_ZX"
