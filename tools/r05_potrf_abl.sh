#!/bin/bash
# tools only (round 5): timing ablations of the tile roles of the Cholesky step launches (wrong numbers): per-launch trace at n = 3000
# base / NOC (no C / R tile loads and stores) / NOP (no tile products) / both
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r05_potrf; mkdir -p $O
for v in "base:" "noc:-DPOTRF_ABL_NOC" "nop:-DPOTRF_ABL_NOP" "nocnop:-DPOTRF_ABL_NOC -DPOTRF_ABL_NOP" $EXTRA_VARIANTS; do
  name=${v%%:*}; defs=${v#*:}
  B=$(mktemp -d /tmp/potrf_abl_XXXX)
  tools/build_variant.sh $B "potrf.hip:$defs" > /dev/null 2>&1
  echo "== $name ($defs)"
  DSVGP_LIB_PATH=$B/libdsvgp_hip.so tools/potrf_inv_trace.sh 3000 2>&1 | grep -v amdgpu.ids
done
