#!/bin/bash
# Build csrc/gemm32.hip with variant -D flags together with tools/gemm32_probe.cpp and run the probe (GPU box).
# usage: tools/gemm32_variants.sh "name1:-DG32_BK=16,-DG32_MINW=4" "name2:..."      (always rebuilt from HEAD's sources)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/gp-derivatives-variational-inference_amd/csrc
B=$(mktemp -d /tmp/g32_XXXX)
FL="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I$R/include -I$C"
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; [ "$defs" = "$spec" ] && defs=""; defs=${defs//,/ }
  hipcc $FL $defs $C/gemm32.hip $R/tools/gemm32_probe.cpp -L/opt/rocm/lib -lrocblas -Wl,-rpath,/opt/rocm/lib -o $B/probe_$name
  echo "=== variant $name  ($defs)"
  $B/probe_$name ${PROBE_REPS:-5}
done
