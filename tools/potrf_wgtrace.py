#!/usr/bin/env python3
"""Per-workgroup phase trace of ONE Cholesky step launch (potrf.hip built with -DPOTRF_TRACE -DPOTRF_DEBUG_K=k).
usage: DSVGP_LIB_PATH=<variant>/libdsvgp_hip.so tools/potrf_wgtrace.py [n] [k]   (k only labels the output: it is compiled in)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator(device=dev).manual_seed(1)
Q = torch.randn(n, n, device=dev, dtype=torch.float64, generator=g)
K = Q @ Q.t() / n + torch.eye(n, device=dev, dtype=torch.float64)
info = torch.zeros(1, dtype=torch.int32, device=dev)
ws = ops.trsm_workspace(n, n + 1, 4096, dev)
pws = ops.potrf_workspace(n, dev)
for rep in range(3):
    A = K.clone()
    ops.potrf_inverse_(ctx, A, info, 4096, ws, pws)
    torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["DSVGP_LIB_PATH"])
NWG, SL = 4096, 24
buf = (ctypes.c_ulonglong * (NWG * SL))()
lib.dsvgp_debug_potrf_trace(buf, NWG)
rows = [list(buf[i * SL:(i + 1) * SL]) for i in range(NWG)]
nblk = (n + 63) // 64
nt = nblk - (k + 1)
wgs = [(i, r) for i, r in enumerate(rows) if r[0] and r[21]]
# (s_memtime counts per XCD: absolute stamps of different workgroups do not compare; durations inside a workgroup do)
t0 = 0
print("n=%d k=%d: %d workgroups traced, longest workgroup %d cycles" % (n, k, len(wgs), max(r[21] - r[0] for _, r in wgs)))
role_name = {0: "crit", 1: "update", 2: "inverse"}
by_role = {}
cus = {}
for i, r in wgs:
    role = r[22]
    hw, xcc = r[23] & 0xffffffff, r[23] >> 32
    cu = (xcc & 0xf, (hw >> 13) & 0x7, (hw >> 12) & 1, (hw >> 8) & 0xf)
    cus.setdefault(cu, []).append((r[0] - t0, r[21] - t0, i, role))
    by_role.setdefault(role, []).append(r)
print("distinct CUs used: %d; workgroups per CU: %s" % (len(cus), sorted({len(v) for v in cus.values()})))
hist = {}
for v in cus.values():
    hist[len(v)] = hist.get(len(v), 0) + 1
print("  CUs by workgroup count:", dict(sorted(hist.items())))
for role, rs in sorted(by_role.items()):
    durs = sorted(r[21] - r[0] for r in rs)
    print("%-8s %4d WGs: duration min %d med %d max %d" % (role_name.get(role, role), len(rs), durs[0], durs[len(rs) // 2], durs[-1]))
# phase anatomy of the longest update / inverse strips
def phases(r):
    out = ["load+LDS %d" % (r[1] - r[0]), "P1 %d" % (r[2] - r[1])]
    prev = r[2]
    for cc in range(5):
        a, b, c = r[3 + 3 * cc], r[4 + 3 * cc], r[5 + 3 * cc]
        if not a:
            break
        out.append("[col %d: stage+barrier %d, product %d, stores %d]" % (cc, a - prev, b - a, c - b))
        prev = c
    out.append("exit %d" % (r[21] - prev))
    return " ".join(out)
for role in (1, 2):
    rs = sorted(by_role.get(role, []), key=lambda r: r[21] - r[0])
    if not rs:
        continue
    for tag, r in (("longest", rs[-1]), ("median", rs[len(rs) // 2]), ("shortest", rs[0])):
        print("%s %s: ncols %d row %d dur %d: %s" % (role_name[role], tag, r[20] >> 16, r[20] & 0xffff, r[21] - r[0], phases(r)))
crit = by_role.get(0, [])
if crit:
    r = crit[0]
    print("crit: update done %d factor done %d exit %d" % (r[1] - r[0], r[2] - r[0], r[21] - r[0]))
