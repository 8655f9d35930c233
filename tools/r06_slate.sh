#!/bin/bash
# round 6: [S - I | m'] under the chain (0) / behind it as a filler (1) / behind it at full grid (2): one box, alternating
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "one_call or stated or committed or c3_full" > $O/tests4.txt 2>&1 || { tail -40 $O/tests4.txt; exit 1; }
tail -1 $O/tests4.txt
for rep in 1 2; do for v in 0 1 2; do for c in ${CONFIGS:-c4 c3 c2}; do
DSVGP_S_LATE=$v timeout -k 10 300 python bench.py --config $c --no-extras --no-cpu-baseline > $O/sl_$c.json 2> $O/sl_$c.err || { tail -20 $O/sl_$c.err; exit 1; }
python - $c $v <<'PY' | tee -a $O/slate.txt
import json, sys
r = json.loads(open("gpurun_out/r06_tail/sl_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "S_LATE=" + sys.argv[2], "%.1f steps/s  %.4f ms" % (r["value"], r["ms_per_step"]))
PY
done; done; done
