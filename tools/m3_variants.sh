#!/bin/bash
# tools only: the M' x M' x M' fp64 products (tools/m3_probe.py) against library variants of gemm64.hip / gemm.hip
# usage: tools/m3_variants.sh "name:-DG64_KCHUNK=512 -DG64_MINW=5" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; [ "$defs" = "$spec" ] && defs=""
  B=$(mktemp -d /tmp/m3v_XXXX)
  $R/tools/build_variant.sh $B "gemm64.hip:$defs" "gemm.hip:$defs" > /dev/null 2>&1 || { echo "build failed: $name"; continue; }
  echo "=== $name ($defs)"
  DSVGP_LIB_PATH=$B/libdsvgp_hip.so python3 $R/tools/m3_probe.py ${M3_N:-3000} 20
done
