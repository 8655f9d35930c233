#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_reftext.py -x -q -m gpu -k "one_call or committed or c3_full or c4 or reftext or stated" > $O/tests_pc.txt 2>&1 || { tail -40 $O/tests_pc.txt; exit 1; }
tail -1 $O/tests_pc.txt
tools/r06_envab.sh DSVGP_PRECLEAR "0 1" "c4 c3" 3
