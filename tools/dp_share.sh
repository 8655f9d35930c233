#!/bin/bash
# one rank's kernels of the C4 step at 8 / 4 / 2 ranks (collectives skipped, bench.py --emulate-world): five-piece C entry point
# (DSVGP_C_STEP=1, default) against the piecewise Python orchestration (DSVGP_C_STEP=0), same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dp_share; mkdir -p $O
cd $R
for rep in 1 2; do for w in 8 4 2; do for c in 1 0; do
  DSVGP_C_STEP=$c python3 bench.py --config c4shard$w --emulate-world $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/b_w${w}_c${c}_$rep.json 2>$O/err.txt && tail -1 $O/b_w${w}_c${c}_$rep.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('c4shard$w one_call=$c rep $rep', round(j['ms_per_step'],4), 'one_call_step', j['config']['one_call_step'])"
done; done; done
