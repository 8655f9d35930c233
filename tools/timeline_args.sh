#!/bin/bash
# rocprofv3 kernel timeline of one bench step with arbitrary bench arguments: tools/timeline_args.sh <out-name> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/timeline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$name
rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o run -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extras > /tmp/ks_$name.log 2>&1
db=$(find /tmp/ks_$name -name "*.db" | head -1)
python3 $R/tools/step_timeline.py $db > $O/timeline_$name.txt 2>&1
tail -1 /tmp/ks_$name.log | cut -c1-200
rm -rf /tmp/ks_$name
