#!/bin/bash
# round 6: C2 / C3 steps with and without the stated directions, one box, alternating
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
for rep in 1 2 3; do for v in 0 1; do for c in ${CONFIGS:-c2 c3}; do
DSVGP_NO_CANON=$v timeout -k 10 300 python bench.py --config $c --no-extras > $O/ab_$c.json 2> $O/ab_$c.err || { tail -20 $O/ab_$c.err; exit 1; }
python - $c $v <<'PY' | tee -a $O/c2ab.txt
import json, sys
r = json.loads(open("gpurun_out/r06_canon/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "general" if sys.argv[2] == "1" else "stated ", "%.1f steps/s  %.4f ms" % (r["value"], r["ms_per_step"]))
PY
done; done; done
