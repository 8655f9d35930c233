#!/bin/bash
# Matrix-pipe / stall / L2 counters of the kernels of the REAL step (bench.py, not a probe): rocprofv3 --pmc passes over
# `python3 bench.py --config c4`, summarised per kernel (tools/summarize_pmc_step.py).  The forward solve
# (gemm64_kernel<float>, the roofline entry of bench.py) is told apart from the Q' solve by its grid size.
# usage (GPU box): bash tools/pmc_step.sh [tag] [bench args]   ->  gpurun_out/<tag>_pmc_step_mfma_busy.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}; shift || true
ARGS=${@:---config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-extras}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_s; mkdir -p /tmp/pmc_s
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_s/p$i -o run -- python3 $R/bench.py $ARGS > /tmp/pmc_s/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 /tmp/pmc_s/p$i.log; }
done
python3 $R/tools/summarize_pmc_step.py /tmp/pmc_s $O/${TAG}_pmc_step_mfma_busy.txt
rm -rf /tmp/pmc_s; mkdir -p /tmp/pmc_s
