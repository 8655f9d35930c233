#!/bin/bash
# PMC passes over tools/gemm32_probe (csrc/gemm32.hip built with the given -D flags) and rocBLAS sgemm next to it:
# matrix-pipe busy fraction, wait / issue-stall fractions, L2 hit rate, clock.  usage (GPU box): bash tools/pmc_gemm32.sh [-Dflags...]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/gp-derivatives-variational-inference_amd/csrc
O=$R/gpurun_out/pmc_gemm32; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I$R/include -I$C "$@" $C/gemm32.hip $R/tools/gemm32_probe.cpp \
  -L/opt/rocm/lib -lrocblas -Wl,-rpath,/opt/rocm/lib -o /tmp/g32probe
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o run -- /tmp/g32probe 2 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$O/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size", "?")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/summary.txt", "w") as out:
    for k in sorted(acc):
        if not any(s in k for s in ("gemm32", "Cijk")): continue
        a = {c: sum(v) / len(v) for c, v in acc[k].items()}
        d = sum(dur[k]) / max(len(dur[k]), 1)
        out.write("%s  launches=%d avg_ms=%.3f\n" % (k, len(dur[k]), d))
        g = a.get
        if g("SQ_BUSY_CU_CYCLES"):
            out.write("    mfma_busy=%.3f  wait_any/wave=%.3f  wait_inst_any/wave=%.3f  wait_inst_lds/wave=%.3f  clk=%.2f GHz\n" % (
                g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * g("SQ_BUSY_CU_CYCLES")), g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES", 1),
                g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES", 1), g("SQ_WAIT_INST_LDS", 0) / g("SQ_WAVE_CYCLES", 1),
                g("GRBM_GUI_ACTIVE", 0) / 8 / (d * 1e6) if d else 0))
        if g("TCC_REQ_sum"):
            out.write("    L2 hit=%.3f  L2 req=%.3g  TCP->TCC read latency=%.0f cyc  tcp_pending_stall=%.3g\n" % (
                g("TCC_HIT_sum", 0) / max(g("TCC_HIT_sum", 0) + g("TCC_MISS_sum", 0), 1), g("TCC_REQ_sum"),
                g("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(g("TCP_TCC_READ_REQ_sum", 1), 1), g("TCP_PENDING_STALL_CYCLES_sum", 0)))
        if g("SQ_ACTIVE_INST_ANY"):
            out.write("    active: valu=%.3g lds=%.3g vmem=%.3g any=%.3g  lds_bank_conflict/idx_active=%.3f\n" % (
                g("SQ_ACTIVE_INST_VALU", 0), g("SQ_ACTIVE_INST_LDS", 0), g("SQ_ACTIVE_INST_VMEM", 0), g("SQ_ACTIVE_INST_ANY", 0),
                g("SQ_LDS_BANK_CONFLICT", 0) / max(g("SQ_LDS_IDX_ACTIVE", 1), 1)))
print(open("$O/summary.txt").read())
PY
