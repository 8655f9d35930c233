#!/usr/bin/env python3
"""tools only: host time per C-ABI call of the Python-orchestrated step (DSVGP_C_STEP=0) -- every function of the ctypes library wrapped with a
timer; prints calls per step and microseconds per call, sorted by total.  usage: python3 tools/call_times.py [config]  (run inside the tree to measure)"""
import os, sys, time, runpy, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DSVGP_C_STEP"] = os.environ.get("DSVGP_C_STEP", "0")
import dsvgp_amd
from dsvgp_amd import _lib
tot, cnt = collections.Counter(), collections.Counter()
class Wrap:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_cache", {})
    def __getattr__(self, name):
        c = self._cache.get(name)
        if c is None:
            fn = getattr(self._lib, name)
            def timed(*a, _fn=fn, _n=name):
                t0 = time.perf_counter()
                r = _fn(*a)
                tot[_n] += time.perf_counter() - t0
                cnt[_n] += 1
                return r
            c = self._cache[name] = timed
        return c
w = Wrap(_lib.lib)
_lib.lib = w
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "__name__", "").startswith(("dsvgp_amd", "gp-derivatives")) and getattr(m, "lib", None) is w._lib:
        m.lib = w
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = 300
sys.argv = ["bench.py", "--config", cfg, "--steps", str(steps), "--warmup", "20", "--no-cpu-baseline", "--no-extras"]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
n = steps + 26
print("total in C calls per step: %.1f us" % (sum(tot.values()) / n * 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print("%-40s %6.2f calls/step  %7.1f us/call  %7.1f us/step" % (k, cnt[k] / n, v / cnt[k] * 1e6, v / n * 1e6))
