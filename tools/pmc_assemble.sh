#!/bin/bash
# PMC passes over the kernel-assembly probes (tools/assemble_probe.py forward, tools/assemble_bwd_probe.py backward): per-kernel
# averages of instruction counts, issue / wait cycles and LDS counters.   usage (GPU box): bash tools/pmc_assemble.sh [fwd|bwd]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-fwd}
P=$R/tools/assemble_probe.py; [ "$W" = bwd ] && P=$R/tools/assemble_bwd_probe.py
O=$R/gpurun_out/pmc_assemble_$W; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o run -- python3 $P > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "kernel_fwd" not in k and "kernel_bwd" not in k: continue
        k = k[k.index("kernel_"):][:50] + " grid=" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/summary.txt", "w") as out:
    for k in sorted(acc):
        out.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            out.write("    %-32s n=%3d avg=%.5g\n" % (c, len(v), sum(v) / len(v)))
print(open("$O/summary.txt").read())
PY
