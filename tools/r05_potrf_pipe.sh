#!/bin/bash
# tools only (round 5): A/B of the software-pipelined strip kernel (POTRF_PIPE) on one box: chain time + accuracy at
# n = 3000 / 3300 / 600 (potrf_inv_probe), per-launch trace at n = 3000
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r05_pipe; mkdir -p $O
tools/potrf_variants.sh "pipe1:-DPOTRF_PIPE=1" "pipe0:-DPOTRF_PIPE=0" "pipe1:-DPOTRF_PIPE=1" "pipe0:-DPOTRF_PIPE=0" ${EXTRA_VARIANTS} > $O/ab.txt 2>&1
cat $O/ab.txt
tools/potrf_inv_trace.sh 3000 > $O/trace_3000.txt 2>&1; tail -60 $O/trace_3000.txt
