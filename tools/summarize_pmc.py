#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) into per-kernel HBM traffic.

gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; FETCH_SIZE
reports exactly 1/2 of the bytes of coalesced streaming reads (re-checked here on kernels with a known byte
count: kernel_bwd_kernel<float> reads the 288000 KiB K_ZX-bar once and reports 150100 KiB; rowdot / colstats
read the 288000 KiB A once and report 144900 / 144100 KiB), WRITE_SIZE is exact (kernel_fwd_kernel<float>
writes K_ZX = 288000 KiB and reports 288000.0).
usage: tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = name[:name.rfind("(")] if name.endswith(")") and "(" in name else name
        key = (name, int(r["Grid_Size"]))
        agg[key].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return agg


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = []
    for key in fetch:
        fv = fetch[key]
        wv = write.get(key, [(0.0, 0.0)])
        f_kib = sum(v for v, _ in fv) / len(fv)
        w_kib = sum(v for v, _ in wv) / len(wv)
        ms = sum(t for _, t in fv) / len(fv)
        hbm = 2 * f_kib * 1024 + w_kib * 1024
        res.append(dict(kernel=key[0], grid_threads=key[1], launches=len(fv), avg_ms=round(ms, 3),
                        FETCH_SIZE_KiB=round(f_kib, 1), WRITE_SIZE_KiB=round(w_kib, 1),
                        hbm_bytes_per_launch=int(hbm), hbm_GBps=round(hbm / (ms * 1e-3) / 1e9, 1) if ms > 0 else None))
    res.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
    json.dump(dict(note=__doc__.split("usage")[0].strip(), kernels=res[:25]), open(out, "w"), indent=1)
    for r in res[:12]:
        print("%-60s grid=%8d n=%3d %8.3f ms  %8.1f MB  %7.1f GB/s" % (r["kernel"][:60], r["grid_threads"], r["launches"],
                                                                      r["avg_ms"], r["hbm_bytes_per_launch"] / 1e6, r["hbm_GBps"] or 0))


if __name__ == "__main__":
    main()
