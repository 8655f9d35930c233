#!/bin/bash
# tools only (round 5): A/B of the critical tile's strip-ordered update (POTRF_CRIT_SLIVER) on one box: chain time + accuracy at
# n = 3000 / 3300 / 600, in-kernel stamps of the critical workgroup at two block columns, per-launch trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r05_potrf; mkdir -p $O
tools/potrf_variants.sh "sliver1:-DPOTRF_CRIT_SLIVER=1" "sliver0:-DPOTRF_CRIT_SLIVER=0" "sliver1:-DPOTRF_CRIT_SLIVER=1" "sliver0:-DPOTRF_CRIT_SLIVER=0" > $O/ab.txt 2>&1
cat $O/ab.txt
for K in 20 40; do
  POTRF_DEFS="-DPOTRF_DEBUG_K=$K" tools/potrf_clock.sh > $O/clock_k$K.txt 2>&1; cat $O/clock_k$K.txt
  POTRF_DEFS="-DPOTRF_DEBUG_K=$K -DPOTRF_CRIT_SLIVER=0" tools/potrf_clock.sh > $O/clock_k${K}_old.txt 2>&1; cat $O/clock_k${K}_old.txt
done
POTRF_CLOCK_N=600 POTRF_DEFS="-DPOTRF_DEBUG_K=4" tools/potrf_clock.sh > $O/clock_n600.txt 2>&1; cat $O/clock_n600.txt
tools/potrf_inv_trace.sh 3000 > $O/trace_3000.txt 2>&1; cat $O/trace_3000.txt
tools/potrf_inv_trace.sh 600 > $O/trace_600.txt 2>&1; cat $O/trace_600.txt
