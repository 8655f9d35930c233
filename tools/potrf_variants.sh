#!/bin/bash
# usage: tools/potrf_variants.sh "name:-DFLAG=..,-DOTHER=.." ...   (builds variants of potrf.hip on the GPU box, times + checks the factor+inverse chain)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$(mktemp -d /tmp/potrf_build_XXXX)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; [ "$defs" = "$spec" ] && defs=""; defs=${defs//,/ }
  $R/tools/build_variant.sh $B/$name "potrf.hip:$defs"
  DSVGP_LIB_PATH=$B/$name/libdsvgp_hip.so python $R/tools/${PROBE:-potrf_inv_probe.py} 2>&1 | tail -${PROBE_LINES:-1}
done
