#!/bin/bash
# round 6: the tail folds of the small-problem step: the step / parallel test files, the C2 timeline (launch count), C2 / C3 / C4 bench lines
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_step.py tests/test_gpu_train.py tests/test_gpu_parallel.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
tools/timeline.sh c2 40 c2
grep -c "dur" gpurun_out/timeline/timeline_c2.txt || true
cp gpurun_out/timeline/timeline_c2.txt $O/timeline_c2.txt
cd $R
for c in c2 c2 c3 c4; do
timeout -k 10 300 python bench.py --config $c --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err || { tail -20 $O/bench_$c.err; exit 1; }
python - $c <<'PY' | tee -a $O/lines.txt
import json, sys
r = json.loads(open("gpurun_out/r06_tail/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.1f steps/s  %.4f ms" % (r["value"], r["ms_per_step"]))
PY
done
