#!/bin/bash
# A/B of the one-call step's tail placement (flag 16 of dsvgp_elbo_step_f32): DSVGP_TAIL_SIDE=0/1, same box, alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_tail; mkdir -p $O
cd $R
for rep in 1 2; do for t in 0 1; do for cfg in "c4 20" "c3 40" "c2 300"; do set -- $cfg
  DSVGP_TAIL_SIDE=$t python3 bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline --no-extras > $O/b_$1_t${t}_$rep.json 2>$O/err.txt && tail -1 $O/b_$1_t${t}_$rep.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 tail_side=$t rep $rep', round(j['ms_per_step'],4))"
done; done; done
