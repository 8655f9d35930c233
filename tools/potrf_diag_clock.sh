#!/bin/bash
# tools only: in-kernel stamps of the persistent diagonal workgroup of the hybrid Cholesky chain (potrf.hip, -DPOTRF_DEBUG):
# per block column, cycles spent waiting for the row tiles, loading, forming C, factoring, publishing.   POTRF_CLOCK_N = n
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$(mktemp -d /tmp/potrf_dbg_XXXX)
$R/tools/build_variant.sh $B "potrf.hip:-DPOTRF_DEBUG ${POTRF_DEFS}" > /dev/null 2>&1
DSVGP_LIB_PATH=$B/libdsvgp_hip.so python3 - <<'PY'
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0); ctx = ops.Context.get(dev)
g = torch.Generator().manual_seed(0)
n = int(os.environ.get('POTRF_CLOCK_N', '3000'))
Q = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (Q @ Q.t() / n + torch.eye(n, dtype=torch.float64)).to(dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
nb = 4096
ws = ops.trsm_workspace(n, n + 1, nb, dev); pws = ops.potrf_workspace(n, dev)
for rep in range(3):
    A = K.clone(); ops.potrf_inverse_(ctx, A, info, nb, ws, pws); torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["DSVGP_LIB_PATH"])
buf = (ctypes.c_ulonglong * (256 * 8))()
lib.dsvgp_debug_potrf_diag(buf)
t = list(buf)
nblk = (n + 63) // 64
print("n = %d, %d blocks; per block (s_memtime ticks, 100 MHz = 10 ns): wait rows | loads->LDS | C formed | factor64 | publish | total | since previous publish" % (n, nblk))
prev = None
for kk in range(1, nblk):
    s = t[kk * 8: kk * 8 + 6]
    d = [s[i + 1] - s[i] for i in range(5)]
    print("  block %2d: %6d %6d %6d %6d %6d | %6d | %s" % (kk, d[0], d[1], d[2], d[3], d[4], s[5] - s[0], "" if prev is None else s[5] - prev))
    prev = s[5]
PY
