#!/bin/bash
# tools only: tools/potrf_wgtrace.sh <k> [n] [extra potrf.hip defines]   -- builds -DPOTRF_TRACE -DPOTRF_DEBUG_K=k and runs potrf_wgtrace.py
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${1:-20}; N=${2:-3000}
B=$(mktemp -d /tmp/potrf_tr_XXXX)
$R/tools/build_variant.sh $B "potrf.hip:-DPOTRF_TRACE -DPOTRF_DEBUG_K=$K $3" > /dev/null 2>&1
DSVGP_LIB_PATH=$B/libdsvgp_hip.so python3 $R/tools/potrf_wgtrace.py $N $K 2>&1 | grep -v amdgpu.ids
