#!/bin/bash
# round 6: the step with directions stated on both sides (full-gradient SVGP): tests, the whole step / train / ops files, the C3 bench line
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "stated or statement" -rP > $O/tests7.txt 2>&1 || { tail -40 $O/tests7.txt; exit 1; }
grep "full gradient\|passed\|failed" $O/tests7.txt | cut -c1-400 | tail -12
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu > $O/tests8.txt 2>&1 || { tail -40 $O/tests8.txt; exit 1; }
tail -2 $O/tests8.txt
timeout -k 10 300 python bench.py --config c3 --no-extras > $O/bench_c3.json 2> $O/bench_c3.err || { tail -20 $O/bench_c3.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_canon/bench_c3.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], json.dumps(r.get("roofline_assembly"))[:1200])
PY
