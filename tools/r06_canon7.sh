#!/bin/bash
# round 6: the C3 / C2 / C4 bench lines with the stated directions (short: no extras)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "drop_in or reference or committed" > $O/tests9.txt 2>&1 || { tail -40 $O/tests9.txt; exit 1; }
tail -2 $O/tests9.txt
for c in c3 c2 c4; do
timeout -k 10 300 python bench.py --config $c --no-extras > $O/bench_$c.json 2> $O/bench_$c.err || { tail -20 $O/bench_$c.err; exit 1; }
python - $c <<'PY'
import json, sys
r = json.loads(open("gpurun_out/r06_canon/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
a = r.get("roofline_assembly") or {}
print(sys.argv[1], r["value"], r["ms_per_step"], {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in a[k].items() if kk in ("kernel", "avg_ms", "frac", "alone_avg_ms", "alone_frac")} for k in ("forward", "backward") if k in a})
PY
done
