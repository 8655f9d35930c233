#!/bin/bash
# tools only: A/B of the fp32-MFMA form of tril(L^T L-bar) (default) against its fp64-accumulated form (DSVGP_PHI_ARG_FP64=1):
# step time at C4 / C3 / C2 (twice each, interleaved) and the reference-text parity figures of both.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { DSVGP_PHI_ARG_FP64=$1 python bench.py --config $2 --no-cpu-baseline --no-extras --steps $3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('phi_arg_fp64=%s %s %.4f ms/step loss %.6f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['config']['final_loss']))" $1 $2; }
for rep in 1 2; do for v in 0 1; do run $v c4 30; run $v c3 30; run $v c2 300; done; done
for v in 0 1; do echo "== reference-text parity, DSVGP_PHI_ARG_FP64=$v"; DSVGP_PHI_ARG_FP64=$v python -m pytest tests/test_gpu_reftext.py -m gpu -q -s -k "not fp64" 2>&1 | grep -E "parity|passed|failed"; done
