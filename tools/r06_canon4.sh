#!/bin/bash
# round 6: the step with stated directions: tests, then the default bench line
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "stated or statement" -rP > $O/tests4.txt 2>&1 || { tail -40 $O/tests4.txt; exit 1; }
grep "parity\|passed\|failed" $O/tests4.txt | tail -25
python -m pytest tests/test_gpu_step.py tests/test_gpu_train.py -x -q -m gpu > $O/tests5.txt 2>&1 || { tail -40 $O/tests5.txt; exit 1; }
tail -2 $O/tests5.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_canon/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], json.dumps(r.get("roofline_assembly")))
PY
