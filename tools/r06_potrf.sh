#!/bin/bash
# tools only (round 6): the register row-strip chain of the critical workgroup (POTRF_CRIT_STRIPS) against round 5's factor64_lds, one box, alternating:
# tests, chain time + accuracy at n = 3000 / 3300 / 600, in-kernel stamps of the critical workgroup, per-launch traces
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06_potrf; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "potrf" > $O/tests.txt 2>&1; echo "potrf tests rc=$?"; tail -3 $O/tests.txt
if [ "$1" = "quick" ]; then exit 0; fi
timeout -k 10 900 tools/potrf_variants.sh "strips1:" "strips0:-DPOTRF_CRIT_STRIPS=0" "strips1:" "strips0:-DPOTRF_CRIT_STRIPS=0" > $O/ab.txt 2>&1; cat $O/ab.txt
for K in 20 40; do
  POTRF_DEFS="-DPOTRF_DEBUG_K=$K" timeout -k 10 300 tools/potrf_clock.sh > $O/clock_k$K.txt 2>&1; cat $O/clock_k$K.txt
done
POTRF_DEFS="-DPOTRF_DEBUG_K=20 -DPOTRF_CRIT_STRIPS=0" timeout -k 10 300 tools/potrf_clock.sh > $O/clock_k20_old.txt 2>&1; cat $O/clock_k20_old.txt
POTRF_CLOCK_N=600 POTRF_DEFS="-DPOTRF_DEBUG_K=4" timeout -k 10 300 tools/potrf_clock.sh > $O/clock_n600.txt 2>&1; cat $O/clock_n600.txt
timeout -k 10 300 tools/potrf_inv_trace.sh 3000 > $O/trace_3000.txt 2>&1; cat $O/trace_3000.txt
timeout -k 10 300 tools/potrf_inv_trace.sh 600 > $O/trace_600.txt 2>&1; cat $O/trace_600.txt
