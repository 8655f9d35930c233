#!/bin/bash
# Build GEMM variants on the GPU box and time them with tools/gemm_probe.cpp.
# usage: tools/probe_variants.sh "name1:-DFLAG=..,-DFLAG2=.." "name2:..."
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$(mktemp -d /tmp/probe_build_XXXX)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; [ "$defs" = "$spec" ] && defs=""; defs=${defs//,/ }
  $R/tools/build_variant.sh $B/$name "gemm.hip:$defs" "gemm64.hip:$defs"      # (G64DEFS ride in the same -D list)
  hipcc -O2 $defs $R/tools/gemm_probe.cpp -I$R/include -L$B/$name -ldsvgp_hip -Wl,-rpath,$B/$name -o $B/$name/probe
  echo "=== variant $name  ($defs)"
  $B/$name/probe ${PROBE_ARGS}
done
