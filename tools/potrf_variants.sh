#!/bin/bash
# usage: tools/potrf_variants.sh "name:-DFLAG=.." ...   (builds variants of potrf.hip on the GPU box, times + checks the factor+inverse chain)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/gp-derivatives-variational-inference_amd/csrc
B=$(mktemp -d /tmp/potrf_build_XXXX)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -I$R/include -I$C"
for f in gemm gemm64 gemm32 elbo assemble ciq api; do hipcc $FL -c $C/$f.hip -o $B/$f.o & done; wait
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; [ "$defs" = "$spec" ] && defs=""; defs=${defs//,/ }
  mkdir -p $B/$name
  hipcc $FL $defs -c $C/potrf.hip -o $B/$name/potrf.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $B/$name/libdsvgp_hip.so $B/$name/potrf.o $B/gemm.o $B/elbo.o $B/assemble.o $B/ciq.o $B/gemm64.o $B/gemm32.o $B/api.o -L/opt/rocm/lib -lrocsolver -lrocblas
  DSVGP_LIB_PATH=$B/$name/libdsvgp_hip.so python $R/tools/potrf_inv_probe.py 2>&1 | tail -1
done
