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
    # (bench.py's MFMA-rate probe runs once after the timed region: not part of a step)
    notprobe = "%s not like '%%mfma_rate_%%'" % namecol
    by_name = c.execute("select %s, count(*), sum(end-start), min(end-start), max(end-start) from kernels where %s group by %s "
                        "order by 3 desc" % (namecol, notprobe, namecol)).fetchall()
    tot = sum(r[2] for r in by_name)
    print("total kernel ms/step %.3f" % (tot / a.steps / 1e6))
    out = ["kernel,calls,calls_per_step,ms_per_step,avg_us,min_us,max_us"]
    for n, cnt, ns, mn, mx in by_name:
        out.append('"%s",%d,%.1f,%.4f,%.1f,%.1f,%.1f' % (n, cnt, cnt / a.steps, ns / a.steps / 1e6, ns / cnt / 1e3,
                                                       mn / 1e3, mx / 1e3))
    print("-- by kernel")
    for n, cnt, ns, mn, mx in by_name[:a.top]:
        print("%-100s calls/step=%6.1f ms/step=%7.3f avg_us=%9.1f" % (n[:100], cnt / a.steps, ns / a.steps / 1e6,
                                                                     ns / cnt / 1e3))
    # one line per (kernel, launch size): the same instantiation is launched on very different shapes in one step (e.g. the
    # forward solve and the Q' solve both run gemm64_kernel<float>), and the bench's roofline entry is about ONE of them
    gridcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    if gridcol:
        rows = c.execute("select %s, %s, count(*), sum(end-start) from kernels where %s group by %s, %s order by 4 desc"
                         % (namecol, gridcol, notprobe, namecol, gridcol)).fetchall()
        print("-- by kernel and launch size (grid threads), the 12 largest")
        for n, g, cnt, ns in rows[:12]:
            print("%-88s grid=%-9d calls/step=%6.1f ms/step=%7.3f avg_us=%9.1f" % (n[:88], g, cnt / a.steps, ns / a.steps / 1e6,
                                                                                  ns / cnt / 1e3))
    if a.csv:
        open(a.csv, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
