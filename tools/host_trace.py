#!/usr/bin/env python3
"""Where the HOST's time goes in the one-call step (tools only): per step, seconds between consecutive calls of
dsvgp_elbo_step_f32, the time inside the call (queueing ~65 launches), the wait for the factorisation status, and the rest of
the Python loop (gather, model / likelihood / mll wrappers, optimizers, schedulers).  usage: python3 tools/host_trace.py [config [more bench.py arguments]]"""
import os, sys, runpy
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
import dsvgp_amd
from dsvgp_amd import _step
orig = _step.ElboEngine.__init__
traces = []
def init(self, *a, **k):
    orig(self, *a, **k)
    self.host_trace = []
    traces.append(self.host_trace)
_step.ElboEngine.__init__ = init
extra = sys.argv[2:]                 # e.g. --emulate-world 8 with config c4shard8
sys.argv = ["bench.py", "--config", cfg, "--steps", "400", "--warmup", "30", "--no-cpu-baseline", "--no-extras"] + extra
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
tr = max(traces, key=len)[40:400]
n = len(tr) - 1
call = sum(b - a for a, b, c in tr) / len(tr)
wait = sum(c - b for a, b, c in tr) / len(tr)
period = (tr[-1][0] - tr[0][0]) / n
print("host per step: period %.1f us = call (queue launches) %.1f + status wait %.1f + rest of the Python loop %.1f" % (
    period * 1e6, call * 1e6, wait * 1e6, (period - call - wait) * 1e6))
