#!/bin/bash
# round 6: C2: one-call tests, host trace, bench lines (three), timeline
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_tail; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "one_call or stated or committed" > $O/tests2.txt 2>&1 || { tail -40 $O/tests2.txt; exit 1; }
tail -1 $O/tests2.txt
timeout -k 10 200 python tools/host_trace.py c2 2>&1 | grep "host per step" | tee -a $O/lines2.txt
for c in c2 c2 c2 c4; do
timeout -k 10 300 python bench.py --config $c --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err || { tail -20 $O/bench_$c.err; exit 1; }
python - $c <<'PY' | tee -a $O/lines2.txt
import json, sys
r = json.loads(open("gpurun_out/r06_tail/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.1f steps/s  %.4f ms" % (r["value"], r["ms_per_step"]))
PY
done
tools/timeline.sh c2 40 c2 > /dev/null 2>&1; cp gpurun_out/timeline/timeline_c2.txt $O/timeline_c2.txt; head -8 $O/timeline_c2.txt | cut -c1-120
