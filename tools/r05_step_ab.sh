#!/bin/bash
# tools only (round 5): whole-step A/B of potrf.hip variants, alternating, one box: usage tools/r05_step_ab.sh "<defs A>" "<defs B>"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ARG_A=$1; ARG_B=$2
BA=$(mktemp -d /tmp/sa_XXXX); BB=$(mktemp -d /tmp/sb_XXXX)
tools/build_variant.sh $BA "potrf.hip:$1" > /dev/null 2>&1 &
tools/build_variant.sh $BB "potrf.hip:$2" > /dev/null 2>&1 &
wait
run() { DSVGP_LIB_PATH=$1/libdsvgp_hip.so python3 bench.py --config $2 --steps $3 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$4 $2', round(j['ms_per_step'],4))"; }
for rep in 1 2; do
  for cfg in "c4 20" "c3 30" "c2 300" "c4shard8 40"; do set -- $cfg
    run $BA $1 $2 "A[$ARG_A]"; run $BB $1 $2 "B[$ARG_B]"
  done
done
