#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp64.py -x -q -k "kernel" 2>&1 | tail -3
export FWD_GEOM="300,512,10,10" PROBE=assemble_probe.py PROBE_LINES=1
tools/assemble_variants.sh "split:" "generic:-DFWD_NO_SPLIT" 2>&1 | grep -v amdgpu.ids | sed 's#/tmp/asm_build_[A-Za-z0-9]*/##'
