#!/bin/bash
# tools only: the piecewise C2 step with variants of the library (build_variant.sh specs "name:file.hip:-DFLAG=..")
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
B=$(mktemp -d /tmp/pwvar_XXXX)
for spec in "$@"; do name=${spec%%:*}; rest=${spec#*:}; tools/build_variant.sh $B/$name "$rest" > /dev/null 2>&1; done
for rep in 1 2; do for spec in head "$@"; do name=${spec%%:*}
  L=$B/$name/libdsvgp_hip.so; [ $name = head ] && L=$R/gp-derivatives-variational-inference_amd/libdsvgp_hip.so
  DSVGP_LIB_PATH=$L DSVGP_C_STEP=0 python bench.py --config c2 --steps 300 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('$name', round(json.loads(sys.stdin.read())['ms_per_step'], 4))"
done; done
