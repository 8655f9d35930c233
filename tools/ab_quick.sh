#!/bin/bash
# tools only: step time at C4 / C3 / C2 (twice each) for the library in the tree; extra bench arguments after "--"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --config $1 --no-cpu-baseline --no-extras --steps $2 "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s %.4f ms/step loss %.6f' % (sys.argv[1], d['ms_per_step'], d['config']['final_loss']))" $1; }
for rep in 1 2; do run c4 30 "$@"; run c3 30 "$@"; run c2 300 "$@"; done
