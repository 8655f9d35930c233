#!/usr/bin/env python3
"""Kernel timeline of one bench step in a rocprofv3 --kernel-trace database (start offset, duration, stream).
usage: step_timeline.py run.db [--back N] [name fragments to leave out ...]
--back N: the step that ends N steps before the last one (default 4: bench.py appends three untimed steps with the side stream OFF
for the assembly-alone timing, so the last TIMED step -- side stream on -- is four from the end)"""
import sqlite3, sys
args = sys.argv[2:]
back = 4
if args[:1] == ["--back"]:
    back = int(args[1]); args = args[2:]
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,stream_id,grid_x,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'hyp_forward' in r[0] or 'column_mean_hyp' in r[0] or 'pack_both' in r[0]]     # (a step's first launch)
back = min(back, len(idx) - 2)
a, b = idx[-2 - back], idx[-1 - back]
t0 = rows[a][1]
skip = tuple(args)
prev_end = t0
for r in rows[a:b]:
    nm = r[0].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    if any(k in nm for k in skip):
        continue
    print("%9.1f us  dur %8.1f  gap %7.1f  s%-3s grid %-8d %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[3], r[4], nm))
    prev_end = max(prev_end, r[2])
print("step span %.1f us" % ((rows[b][1] - t0) / 1e3))
