#!/bin/bash
# PMC passes over tools/gemm_probe (shipped library): per-kernel averages of the matrix-pipe / stall / cache counters.
# usage (on the GPU box): bash tools/pmc_gemm.sh   -> gpurun_out/pmc_gemm/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_gemm; mkdir -p $O
export TMPDIR=/tmp
hipcc -O2 $R/tools/gemm_probe.cpp -I$R/include -L$R/gp-derivatives-variational-inference_amd -ldsvgp_hip -Wl,-rpath,$R/gp-derivatives-variational-inference_amd -o /tmp/gemm_probe
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o run -- /tmp/gemm_probe > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_kernel" not in k: continue
        k = k[k.index("gemm_kernel"):][:60] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/summary.txt", "w") as out:
    for k in sorted(acc):
        out.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            out.write("    %-40s n=%2d avg=%.4g\n" % (c, len(v), sum(v) / len(v)))
print(open("$O/summary.txt").read()[:200])
PY
