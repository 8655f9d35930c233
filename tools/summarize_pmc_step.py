#!/usr/bin/env python3
"""Per-kernel averages of the SQ / TCC counters collected by tools/pmc_step.sh over the real step.

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the 4 SIMDs of a CU
and over all CUs; SQ_BUSY_CU_CYCLES counts CU-busy cycles (quad-cycle granularity x 4), so
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)
is the fraction of the busy time in which a SIMD's matrix pipe executes (the convention of profiles/r01_i / r02_a);
SQ_WAIT_* / SQ_ACTIVE_* / SQ_WAVE_CYCLES are quad-cycles per wave and are reported as fractions of SQ_WAVE_CYCLES.
Durations come from the same counter-collection rows (profiled passes run 2-5 % slower than un-profiled ones).
usage: tools/summarize_pmc_step.py <dir with p1.. p2.. p3..> <out.txt>"""
import collections
import csv
import glob
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            name = name[:name.rfind("(")] if name.endswith(")") and "(" in name else name
            key = "%s grid=%s" % (name[:80], r.get("Grid_Size", "?"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "TCC_HIT_sum", "SQ_WAVES"):
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    rows = []
    for k, c in acc.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        d = sum(dur[k]) / max(len(dur[k]), 1)
        n = max(len(v) for v in c.values())
        rows.append((d * n, k, a, d, n))
    rows.sort(reverse=True)
    with open(out, "w") as fo:
        fo.write(__doc__.split("usage")[0].strip() + "\n\n")
        for _, k, a, d, n in rows[:28]:
            g = a.get
            fo.write("%s  launches=%d avg_ms=%.3f\n" % (k, n, d))
            if g("SQ_BUSY_CU_CYCLES"):
                wc = g("SQ_WAVE_CYCLES", 1.0) or 1.0
                fo.write("    mfma_busy=%.3f  wait_any/wave=%.3f  wait_inst_any/wave=%.3f  wait_inst_lds/wave=%.3f  clk=%.2f GHz\n" % (
                    g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * g("SQ_BUSY_CU_CYCLES")), g("SQ_WAIT_ANY", 0) / wc,
                    g("SQ_WAIT_INST_ANY", 0) / wc, g("SQ_WAIT_INST_LDS", 0) / wc,
                    g("GRBM_GUI_ACTIVE", 0) / 8 / (d * 1e-3) / 1e9 if d > 0 else 0))
            if g("SQ_ACTIVE_INST_ANY"):
                fo.write("    active_valu=%.0f active_lds=%.0f active_vmem=%.0f active_any=%.0f (quad-cycles, summed over waves)  lds_conflict/lds_active=%.4f  waves=%.0f\n" % (
                    g("SQ_ACTIVE_INST_VALU", 0), g("SQ_ACTIVE_INST_LDS", 0), g("SQ_ACTIVE_INST_VMEM", 0), g("SQ_ACTIVE_INST_ANY", 0),
                    g("SQ_LDS_BANK_CONFLICT", 0) / max(g("SQ_LDS_IDX_ACTIVE", 1), 1), g("SQ_WAVES", 0)))
            if g("TCC_REQ_sum"):
                fo.write("    L2 hit rate=%.3f  (hit %.3g, miss %.3g, req %.3g)\n" % (
                    g("TCC_HIT_sum", 0) / max(g("TCC_HIT_sum", 0) + g("TCC_MISS_sum", 0), 1), g("TCC_HIT_sum", 0), g("TCC_MISS_sum", 0), g("TCC_REQ_sum", 0)))
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
