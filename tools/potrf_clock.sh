#!/bin/bash
# tools only: build potrf.hip with -DPOTRF_DEBUG and print in-kernel cycle stamps of the diagonal workgroup of the
# fused Cholesky step kernel (block column 20 of a 3000 x 3000 matrix)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$(mktemp -d /tmp/potrf_dbg_XXXX)
$R/tools/build_variant.sh $B "potrf.hip:-DPOTRF_DEBUG ${POTRF_DEFS}"
DSVGP_LIB_PATH=$B/libdsvgp_hip.so python - <<'PY'
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0); ctx = ops.Context.get(dev)
g = torch.Generator().manual_seed(0)
n = 3000
Q = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (Q @ Q.t() / n + torch.eye(n, dtype=torch.float64)).to(dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(3):
    A = K.clone(); ops.potrf_(ctx, A, info, 1); torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["DSVGP_LIB_PATH"])
buf = (ctypes.c_ulonglong * 16)()
lib.dsvgp_debug_potrf_clock(buf)
t = list(buf)
names = {1: "loads done", 2: "2 products + F formed", 3: "factor64 + W + stores"}
print("diag workgroup, block column 20 (shader cycles since kernel entry):")
for s in (1, 2, 3):
    print("  %-24s %8d  (+%d)" % (names[s], t[s] - t[0], t[s] - t[s - 1]))
print("  inside factor64, kb = 0:  factor16 (1 wave) %d, phase (b) %d, phase (c) %d" % (t[9] - t[8], t[10] - t[9], t[11] - t[10]))
PY
