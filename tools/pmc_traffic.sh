#!/bin/bash
# HBM traffic per kernel of the C4 step: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over bench.py,
# summarised with the gfx950 corrections (tools/summarize_pmc.py) -> gpurun_out/<tag>_pmc_hbm_traffic.json
# usage (GPU box): bash tools/pmc_traffic.sh [tag] [bench args]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}; shift || true
ARGS=${@:---config c4 --steps 6 --warmup 2 --no-cpu-baseline --no-extras}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o run -- python3 $R/bench.py $ARGS > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o run -- python3 $R/bench.py $ARGS > /tmp/pmc_w.log 2>&1
python3 $R/tools/summarize_pmc.py /tmp/pmc_f /tmp/pmc_w $O/${TAG}_pmc_hbm_traffic.json
rm -rf /tmp/pmc_f /tmp/pmc_w
