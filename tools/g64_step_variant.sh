#!/bin/bash
# tools only: build a library variant (extra -D flags for gemm.hip / gemm64.hip) and run bench.py configs against it
# usage: tools/g64_step_variant.sh "<defs>" "<bench args 1>" "<bench args 2>" ...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/gp-derivatives-variational-inference_amd/csrc; B=/tmp/g64v_build; mkdir -p $B
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -I$R/include -I$C"
defs=$1; shift
for f in assemble elbo potrf ciq api; do [ -f $B/$f.o ] || hipcc $FL -c $C/$f.hip -o $B/$f.o & done
hipcc $FL $defs -c $C/gemm.hip -o $B/gemm.o & hipcc $FL $defs -c $C/gemm64.hip -o $B/gemm64.o & wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libdsvgp_hip.so $B/gemm.o $B/gemm64.o $B/assemble.o $B/elbo.o $B/potrf.o $B/ciq.o $B/api.o -L/opt/rocm/lib -lrocsolver -lrocblas
for a in "$@"; do
  DSVGP_LIB_PATH=$B/libdsvgp_hip.so python $R/bench.py $a --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$defs] $a', round(j['ms_per_step'],3))"
done
