#!/bin/bash
# round 6: A/B of one environment switch of the one-call step on one box, alternating: tools/r06_envab.sh NAME "v1 v2 .." "c4 c3 .." [reps]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
name=$1; vals=$2; cfgs=${3:-c4}; reps=${4:-2}
for rep in $(seq $reps); do for v in $vals; do for c in $cfgs; do
env $name=$v timeout -k 10 300 python bench.py --config $c --no-extras --no-cpu-baseline > $O/ab_$c.json 2> $O/ab_$c.err || { tail -20 $O/ab_$c.err; exit 1; }
python - $c $name $v <<'PY' | tee -a $O/envab_$name.txt
import json, sys
r = json.loads(open("gpurun_out/r06_tail/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2] + "=" + sys.argv[3], "%.1f steps/s  %.4f ms" % (r["value"], r["ms_per_step"]))
PY
done; done; done
