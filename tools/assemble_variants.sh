#!/bin/bash
# usage: [PROBE=assemble_bwd_probe.py PROBE_LINES=2] tools/assemble_variants.sh "name:-DFLAG=.." ...   (builds variants of assemble.hip on the GPU box and times them)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/gp-derivatives-variational-inference_amd/csrc
B=$(mktemp -d /tmp/asm_build_XXXX)      # always rebuilt from HEAD's sources
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -I$R/include -I$C"
for f in gemm gemm64 gemm32 elbo potrf ciq api; do hipcc $FL -c $C/$f.hip -o $B/$f.o & done; wait
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; defs=${defs//,/ }
  mkdir -p $B/$name
  hipcc $FL $defs -c $C/assemble.hip -o $B/$name/assemble.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $B/$name/libdsvgp_hip.so $B/$name/assemble.o $B/gemm.o $B/elbo.o $B/potrf.o $B/ciq.o $B/gemm64.o $B/gemm32.o $B/api.o -L/opt/rocm/lib -lrocsolver -lrocblas
  DSVGP_LIB_PATH=$B/$name/libdsvgp_hip.so python $R/tools/${PROBE:-assemble_probe.py} 2>&1 | tail -${PROBE_LINES:-1}
done
