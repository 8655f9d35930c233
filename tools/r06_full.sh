#!/bin/bash
# tools only (round 6): the whole GPU suite + the C4 / C2 / C3 bench lines of the library in the tree
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06_full; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "gpu tests rc=$?"; tail -4 $O/tests.txt
for cfg in "c4 20" "c2 300" "c3 30" "c4shard8 40"; do set -- $cfg
  EW=""; [ "$1" = "c4shard8" ] && EW="--emulate-world 8"
  python3 bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline --no-extras $EW > $O/bench_$1.json 2>/dev/null && tail -1 $O/bench_$1.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],4))"
done
