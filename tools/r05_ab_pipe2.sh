#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pipe; mkdir -p $O
cd $R
one() { label=$1; cfg=$2; steps=$3; shift 3
  env "$@" python3 bench.py --config $cfg --steps $steps --warmup 4 --no-cpu-baseline --no-extras > $O/b_${cfg}_$label.json 2>$O/err_${cfg}_$label.txt \
    && tail -1 $O/b_${cfg}_$label.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$cfg $label', round(j['ms_per_step'],4), 'loss', j['config'].get('final_loss'))" \
    || { echo "$cfg $label FAILED"; tail -5 $O/err_${cfg}_$label.txt; }
}
one off_a c4 20 DSVGP_SOLVE_PIPE=0
one late_nopad_450_750 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=0
one late_nopad_350 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=0 DSVGP_PIPE_K1=350 DSVGP_PIPE_K2=350
one late_nopad_250 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=0 DSVGP_PIPE_K1=250 DSVGP_PIPE_K2=250
one late_nopad_600 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=0 DSVGP_PIPE_K1=600 DSVGP_PIPE_K2=600
one late_pad_350 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_K1=350 DSVGP_PIPE_K2=350
one late_pad20k_350_600 c4 20 DSVGP_SOLVE_PIPE=1 DSVGP_PIPE_PAD=20480 DSVGP_PIPE_K1=350 DSVGP_PIPE_K2=600
one off_b c4 20 DSVGP_SOLVE_PIPE=0
