#!/bin/bash
# tools only: build a library variant (extra -D flags for gemm.hip / gemm64.hip) and run bench.py configs against it
# usage: tools/g64_step_variant.sh "<defs>" "<bench args 1>" "<bench args 2>" ...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$(mktemp -d /tmp/g64v_build_XXXX)
defs=$1; shift
$R/tools/build_variant.sh $B "gemm.hip:$defs" "gemm64.hip:$defs"
for a in "$@"; do
  DSVGP_LIB_PATH=$B/libdsvgp_hip.so python $R/bench.py $a --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$defs] $a', round(j['ms_per_step'],3), 'dominant kernel avg_ms', round((j.get('roofline') or {}).get('avg_ms', 0), 4))"
done
