#!/bin/bash
# Build GEMM variants on the GPU box and time them with tools/gemm_probe.cpp.
# usage: tools/probe_variants.sh "name1:-DFLAG=..,-DFLAG2=.." "name2:..."
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/gp-derivatives-variational-inference_amd/csrc
B=/tmp/probe_build; mkdir -p $B
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -I$R/include -I$C"
for f in assemble elbo potrf ciq api; do hipcc $FL -c $C/$f.hip -o $B/$f.o & done; wait
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}; defs=${defs//,/ }
  mkdir -p $B/$name
  hipcc $FL $defs -c $C/gemm.hip -o $B/$name/gemm.o
  hipcc $FL $defs -c $C/gemm64.hip -o $B/$name/gemm64.o   # (G64DEFS ride in the same -D list)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $B/$name/libdsvgp_hip.so $B/$name/gemm.o $B/assemble.o $B/elbo.o $B/potrf.o $B/ciq.o $B/$name/gemm64.o $B/api.o -L/opt/rocm/lib -lrocsolver -lrocblas
  hipcc -O2 $defs $R/tools/gemm_probe.cpp -I$R/include -L$B/$name -ldsvgp_hip -Wl,-rpath,$B/$name -o $B/$name/probe
  echo "=== variant $name  ($defs)"
  $B/$name/probe ${PROBE_ARGS}
done
