#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_ngd.py tests/test_ciq.py -x -q -m gpu > $O/tests3.txt 2>&1 || { tail -40 $O/tests3.txt; exit 1; }
tail -1 $O/tests3.txt
for c in c4 c4 c2; do
timeout -k 10 300 python bench.py --config $c --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err || { tail -20 $O/bench_$c.err; exit 1; }
python - $c <<'PY' | tee -a $O/lines3.txt
import json, sys
r = json.loads(open("gpurun_out/r06_tail/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.1f steps/s  %.4f ms" % (r["value"], r["ms_per_step"]))
PY
done
