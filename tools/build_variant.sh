#!/bin/bash
# tools only: build a variant of libdsvgp_hip.so into a FRESH directory, every source compiled from the working tree
# (no object reuse between runs; the source list is build_ext.py's).
# usage: tools/build_variant.sh <outdir> ["file.hip:-DFLAG=1 -DOTHER=2" ...]     -> <outdir>/libdsvgp_hip.so
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/gp-derivatives-variational-inference_amd/csrc
out=$1; shift
rm -rf "$out"; mkdir -p "$out"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -I$R/include -I$C"
srcs=$(python3 -c "import re,sys; s=open('$R/gp-derivatives-variational-inference_amd/build_ext.py').read(); print(' '.join(re.findall(r'\"(\w+\.hip)\"', s[s.index('SOURCES'):s.index(']', s.index('SOURCES'))])))")
objs=""
for f in $srcs; do
  defs=""
  for spec in "$@"; do [ "${spec%%:*}" = "$f" ] && defs="$defs ${spec#*:}"; done
  hipcc $FL $defs -c $C/$f -o $out/${f%.hip}.o &
  objs="$objs $out/${f%.hip}.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdsvgp_hip.so $objs -L/opt/rocm/lib -lrocsolver -lrocblas
