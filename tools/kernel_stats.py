#!/usr/bin/env python3
"""Per-kernel time per bench step out of a rocprofv3 --kernel-trace run (rocpd sqlite database).

  python tools/kernel_stats.py gpurun_out/prof/run_results.db --steps 12 [--csv out.csv]

``--steps`` = warm-up + timed steps of the profiled ``bench.py`` command (kernel time is divided by it).
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by %s "
                     "order by 3 desc" % (namecol, namecol)).fetchall()
    tot = sum(r[2] for r in rows)
    print("total kernel ms/step %.3f" % (tot / a.steps / 1e6))
    out = ["kernel,calls,calls_per_step,ms_per_step,avg_us,min_us,max_us"]
    for n, cnt, ns, mn, mx in rows:
        out.append('"%s",%d,%.1f,%.4f,%.1f,%.1f,%.1f' % (n, cnt, cnt / a.steps, ns / a.steps / 1e6, ns / cnt / 1e3,
                                                       mn / 1e3, mx / 1e3))
    for n, cnt, ns, mn, mx in rows[:a.top]:
        print("%-100s calls/step=%6.1f ms/step=%7.3f avg_us=%9.1f" % (n[:100], cnt / a.steps, ns / a.steps / 1e6,
                                                                     ns / cnt / 1e3))
    if a.csv:
        open(a.csv, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
