#!/bin/bash
# tools only (round 5): the interleaved-prefetch experiment of the Cholesky strips (POTRF_IL) alone: chain A/B, per-workgroup trace, per-launch trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/r05_potrf
tools/potrf_variants.sh "il1:" "il0:-DPOTRF_IL=0" "il1:" "il0:-DPOTRF_IL=0" > gpurun_out/r05_potrf/il_ab.txt 2>&1; cat gpurun_out/r05_potrf/il_ab.txt
for k in 5 25; do tools/potrf_wgtrace.sh $k 3000; done > gpurun_out/r05_potrf/wgtrace_il.txt 2>&1; cat gpurun_out/r05_potrf/wgtrace_il.txt
tools/potrf_inv_trace.sh 3000 > gpurun_out/r05_potrf/trace_3000_il.txt 2>&1; cat gpurun_out/r05_potrf/trace_3000_il.txt
