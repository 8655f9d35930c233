#!/bin/bash
# Round-end measurement pass (GPU box): bench lines of every config, rocprofv3 kernel stats of C4 / C3 / C5 / C2, the C4 timeline.
# Outputs under gpurun_out/final/ (copy what is to be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd $R
if [ "$SKIP_BENCH" != "1" ]; then
python3 bench.py > $O/bench_c4.json 2> $O/bench_c4.err && tail -1 $O/bench_c4.json | cut -c1-400
for cfg in "c2 300" "c3 40" "c5 8"; do set -- $cfg
  python3 bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$1.json 2>/dev/null && tail -1 $O/bench_$1.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],4), (j.get('roofline') or {}).get('frac'))"
done
# one rank's kernels of the C4 step at 8 / 4 / 2 ranks (collectives skipped), with the replicated stage sharded (default) and replicated
for w in 8 4 2; do for s in 1 0; do
  DSVGP_SHARD_REPLICATED=$s python3 bench.py --config c4shard$w --emulate-world $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c4shard${w}_shard$s.json 2>/dev/null && tail -1 $O/bench_c4shard${w}_shard$s.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('c4shard$w sharded=$s', round(j['ms_per_step'],4))"
done; done
# deterministic mode and the piecewise (Python-orchestrated) path, for the record
DSVGP_DETERMINISTIC=1 python3 bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c4_deterministic.json 2>/dev/null && tail -1 $O/bench_c4_deterministic.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('c4 deterministic', round(j['ms_per_step'],4))"
DSVGP_DETERMINISTIC=1 python3 bench.py --config c2 --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_c2_deterministic.json 2>/dev/null && tail -1 $O/bench_c2_deterministic.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2 deterministic', round(j['ms_per_step'],4))"
DSVGP_C_STEP=0 python3 bench.py --config c2 --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_c2_piecewise.json 2>/dev/null && tail -1 $O/bench_c2_piecewise.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2 piecewise path', round(j['ms_per_step'],4))"
for cfg in "c4 10" "c3 20" "c2 300"; do set -- $cfg      # the double-precision model mode of the reference's experiment scripts
  python3 bench.py --fp64 --config $1 --steps $2 > $O/bench_$1_fp64.json 2>/dev/null && tail -1 $O/bench_$1_fp64.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 fp64', round(j['ms_per_step'],4))"
done
fi
cd /tmp && export TMPDIR=/tmp
for cfg in "c4 9" "c3 9" "c5 4" "c2 40"; do set -- $cfg
  rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python3 $R/bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ks.log 2>&1
  db=$(find /tmp/ks -name "*.db" | head -1)
  extra=6; [ "$1" = "c5" ] && extra=3      # warm-up 3 (+ the 3 untimed steps bench.py adds for the assembly-alone timing; not for CIQ)
  python3 $R/tools/kernel_stats.py $db --steps $(($2 + extra)) > $O/kernel_stats_$1.txt 2>&1 || cp /tmp/ks.log $O/kernel_stats_$1.err
  if [ "$1" = "c4" ] || [ "$1" = "c2" ]; then python3 $R/tools/step_timeline.py $db > $O/timeline_$1.txt 2>&1; fi
  head -3 $O/kernel_stats_$1.txt
done
rm -rf /tmp/ks
