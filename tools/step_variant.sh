#!/bin/bash
# tools only: build a library variant (one "file.hip:-Dflags" spec for tools/build_variant.sh) and run bench.py configs against it
# usage: tools/step_variant.sh "gemm32.hip:-DG32_SK_BELOW=1024" "--config c5 --steps 8" "--config c3" ...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$(mktemp -d /tmp/stepv_build_XXXX)
spec=$1; shift
$R/tools/build_variant.sh $B "$spec" > /dev/null
for a in "$@"; do
  case "$a" in *--steps*) st="";; *) st="--steps 30";; esac
  DSVGP_LIB_PATH=$B/libdsvgp_hip.so python $R/bench.py $a $st --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$spec] $a', round(j['ms_per_step'],3))"
done
