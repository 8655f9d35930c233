#!/bin/bash
# usage: [PROBE=assemble_bwd_probe.py PROBE_LINES=2] tools/assemble_variants.sh "name:-DFLAG=..,-DOTHER=.." ...   (builds variants of assemble.hip on the GPU box and times them)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$(mktemp -d /tmp/asm_build_XXXX)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; [ "$defs" = "$spec" ] && defs=""; defs=${defs//,/ }
  $R/tools/build_variant.sh $B/$name "assemble.hip:$defs"
  DSVGP_LIB_PATH=$B/$name/libdsvgp_hip.so python $R/tools/${PROBE:-assemble_probe.py} 2>&1 | tail -${PROBE_LINES:-1}
done
