#!/bin/bash
# per-launch durations of the Cholesky kernels (rocprofv3 kernel trace of tools/potrf_probe.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/potrf_trace
rocprofv3 --kernel-trace -d /tmp/potrf_trace -o run -- python3 $R/tools/potrf_probe.py > /tmp/potrf_trace.log 2>&1
python3 - <<'PY'
import sqlite3, glob, re
m = re.search(r"probe: runs_algo1=(\d+) reps=(\d+)", open('/tmp/potrf_trace.log').read())
runs, reps = int(m.group(1)), int(m.group(2))        # factorisations traced = runs * reps (printed by the probe)
db = glob.glob('/tmp/potrf_trace/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
for pat in ("chol_step_kernel", "chol_panels_kernel"):
    d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
    if not d: continue
    print(pat, "launches", len(d), "ms per factorisation %.3f" % (sum(d) / 1e3 / (runs * reps)))
    if pat == "chol_step_kernel":
        per = len(d) // (runs * reps)
        last = d[-per:]
        print("  per-k durations (us), last run:", " ".join("%.1f" % x for x in last))
        st = [s for n, s, e in rows if pat in n][-per:]; en = [e for n, s, e in rows if pat in n][-per:]
        print("  gaps between launches (us):", " ".join("%.1f" % ((st[i + 1] - en[i]) / 1e3) for i in range(0, per - 1, 4)))
PY
