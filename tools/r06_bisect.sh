#!/bin/bash
# tools only: the piecewise C2 step of several trees on one box (git worktrees gpurun_scratch_<commit>, built before the call)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for t in "$@"; do
  (cd $t && DSVGP_C_STEP=0 python bench.py --config c2 --steps 300 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('$t', round(json.loads(sys.stdin.read())['ms_per_step'], 4))")
done; done
