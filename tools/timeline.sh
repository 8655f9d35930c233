#!/bin/bash
# rocprofv3 kernel timeline of one bench step: tools/timeline.sh <config> <steps> <out-name> [env assignments...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cfg=$1; steps=$2; name=$3; shift 3
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/timeline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$name
rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o run -- python3 $R/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-extras > /tmp/ks_$name.log 2>&1
db=$(find /tmp/ks_$name -name "*.db" | head -1)
python3 $R/tools/step_timeline.py $db > $O/timeline_$name.txt 2>&1
python3 $R/tools/kernel_stats.py $db --steps $(($steps + 6)) > $O/kernel_stats_$name.txt 2>&1
tail -1 /tmp/ks_$name.log | cut -c1-200
rm -rf /tmp/ks_$name
