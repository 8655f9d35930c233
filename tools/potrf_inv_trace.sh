#!/bin/bash
# per-launch durations of the fused Cholesky + inverse chain (rocprofv3 kernel trace of tools/potrf_inv_trace.py)
# usage: tools/potrf_inv_trace.sh [n] ; DSVGP_LIB_PATH selects a variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3000}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/potrf_inv_trace
rocprofv3 --kernel-trace -d /tmp/potrf_inv_trace -o run -- python3 $R/tools/potrf_inv_trace.py $N 3 > /tmp/potrf_inv_trace.log 2>&1
python3 - <<'PY'
import sqlite3, glob, re
m = re.search(r"probe: n=(\d+) reps=(\d+)", open('/tmp/potrf_inv_trace.log').read())
n, reps = int(m.group(1)), int(m.group(2))
db = glob.glob('/tmp/potrf_inv_trace/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
names = [r for r in rows if any(p in r[0] for p in ("chol_step_kernel", "chol_yrow_kernel", "chol_panels_kernel"))]
per = len(names) // reps
last = names[-per:]
nblk = (n + 63) // 64
print("n=%d: %d launches per factorisation, chain %.3f ms (first launch start -> last launch end)" % (n, per, (last[-1][2] - last[0][1]) / 1e6))
steps = [r for r in last if "chol_step" in r[0]]
out = []
for k, (nm, s, e) in enumerate(steps):
    kk = k - 1
    nt = nblk - (kk + 1)
    nA = 1 if kk < 0 else nt * (nt + 1) // 2
    nI = 0 if kk < 0 else nt * (kk + 1) + (kk + 1)
    out.append((kk, nA + nI, (e - s) / 1e3))
print("  k: tiles: us   " + "  ".join("%d:%d:%.1f" % o for o in out))
gaps = [(steps[i + 1][1] - steps[i][2]) / 1e3 for i in range(len(steps) - 1)]
print("  mean gap between step launches %.2f us; sum of step durations %.3f ms" % (sum(gaps) / max(len(gaps), 1), sum(o[2] for o in out) / 1e3))
for r in last:
    if "chol_step" not in r[0]:
        print("  %s %.1f us" % (r[0][:40], (r[2] - r[1]) / 1e3))
PY
rm -rf /tmp/potrf_inv_trace
