#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
for v in 0 1 0 1; do
DSVGP_C_STEP=0 DSVGP_NO_CANON=$v timeout -k 10 300 python bench.py --config c2 --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/pw.json 2> $O/pw.err || { tail -20 $O/pw.err; exit 1; }
python - $v <<'PY'
import json, sys
r = json.loads(open("gpurun_out/r06_tail/pw.json").read().strip().splitlines()[-1])
print("piecewise c2 NO_CANON=" + sys.argv[1], "%.4f ms" % r["ms_per_step"])
PY
done
