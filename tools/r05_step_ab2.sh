#!/bin/bash
# tools only (round 5): whole-step A/B of library variants, alternating, one box:
#   usage: tools/r05_step_ab2.sh "<file.hip:defs for A>" "<file.hip:defs for B>" [config:steps ...]     (default: c4:20 c3:30 c2:300 c4shard8:40)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
BA=$(mktemp -d /tmp/sa_XXXX); BB=$(mktemp -d /tmp/sb_XXXX)
tools/build_variant.sh $BA "$1" > /dev/null 2>&1 &
tools/build_variant.sh $BB "$2" > /dev/null 2>&1 &
wait
shift 2
CFGS="$@"; [ -z "$CFGS" ] && CFGS="c4:20 c3:30 c2:300 c4shard8:40"
run() { DSVGP_LIB_PATH=$1/libdsvgp_hip.so python3 bench.py --config $2 --steps $3 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print('$4 $2', round(j['ms_per_step'],4), 'roofline', r.get('frac'), r.get('kernel_avg_ms', r.get('avg_ms')))"; }
for rep in 1 2; do
  for cs in $CFGS; do
    run $BA ${cs%%:*} ${cs#*:} "A"; run $BB ${cs%%:*} ${cs#*:} "B"
  done
done
