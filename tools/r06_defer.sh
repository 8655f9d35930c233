#!/bin/bash
# round 6: the eager loop with the factorisation status read one step late: tests, then C2 / C3 / C4 with DSVGP_DEFER_STATUS 0 / 1 alternating
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_train.py tests/test_gpu_ops.py -x -q -m gpu -k "deferred or guarded or graph or drop_in or train or adam" > $O/tests_defer.txt 2>&1 || { tail -40 $O/tests_defer.txt; exit 1; }
tail -1 $O/tests_defer.txt
for v in 0 1; do DSVGP_DEFER_STATUS=$v timeout -k 10 200 python tools/host_trace.py c2 2>&1 | grep "host per step" | sed "s/^/DEFER=$v /" | tee -a $O/defer.txt; done
tools/r06_envab.sh DSVGP_DEFER_STATUS "0 1" "${CONFIGS:-c2 c3 c4}" 3
