#!/bin/bash
# tools only (round 5): A/B of flag 128 of the one-call step (forward solve in row ranges under the Cholesky chain), same box,
# alternating; DSVGP_PIPE_K1 / K2 (per mille of the block rows) and DSVGP_PIPE_PAD (bytes of unused LDS) select the variant
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pipe; mkdir -p $O
cd $R
one() { # label cfg steps env-assignments...
  label=$1; cfg=$2; steps=$3; shift 3
  env "$@" python3 bench.py --config $cfg --steps $steps --warmup 4 --no-cpu-baseline --no-extras > $O/b_${cfg}_$label.json 2>$O/err_${cfg}_$label.txt \
    && tail -1 $O/b_${cfg}_$label.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$cfg $label', round(j['ms_per_step'],4), 'loss', j['config'].get('final_loss'))" \
    || { echo "$cfg $label FAILED"; tail -5 $O/err_${cfg}_$label.txt; }
}
for rep in 1 2; do
  one off_$rep c4 20 DSVGP_SOLVE_PIPE=0
  one on_$rep c4 20 DSVGP_SOLVE_PIPE=1
done
one k300_700 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_K1=300 DSVGP_PIPE_K2=700
one k500_800 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_K1=500 DSVGP_PIPE_K2=800
one k600_850 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_K1=600 DSVGP_PIPE_K2=850
one nopad c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=0
one pad20k c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=20480
one off_3 c4 20 DSVGP_SOLVE_PIPE=0
one off c3 30 DSVGP_SOLVE_PIPE=0
one on c3 30 DSVGP_SOLVE_PIPE=1
