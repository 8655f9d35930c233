#!/bin/bash
# tools only: build a library that routes EVERY eligible fp64 product through gemm64.hip and run the GPU test suite against it
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/gp-derivatives-variational-inference_amd/csrc; B=/tmp/g64_build; mkdir -p $B
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -I$R/include -I$C"
for f in assemble elbo potrf ciq gemm64 api; do hipcc $FL -c $C/$f.hip -o $B/$f.o & done
hipcc $FL -DGEMM64=1 -DGEMM64_MIN_TILES=1 -DGEMM64_MIN_K=1 -c $C/gemm.hip -o $B/gemm.o & wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libdsvgp_hip.so $B/gemm.o $B/gemm64.o $B/assemble.o $B/elbo.o $B/potrf.o $B/ciq.o $B/api.o -L/opt/rocm/lib -lrocsolver -lrocblas
DSVGP_LIB_PATH=$B/libdsvgp_hip.so python -m pytest $R/tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
