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
n = int(os.environ.get('POTRF_CLOCK_N', '3000'))
Q = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (Q @ Q.t() / n + torch.eye(n, dtype=torch.float64)).to(dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(3):
    A = K.clone(); ops.potrf_(ctx, A, info, 1); torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["DSVGP_LIB_PATH"])
buf = (ctypes.c_ulonglong * 32)()
lib.dsvgp_debug_potrf_clock(buf)
t = list(buf)
names = {1: "loads -> LDS, barrier", 4: "product 1 (wave 0)", 5: "barrier", 6: "T, A_jk -> LDS, barrier", 7: "product 2 (wave 0)",
         2: "F formed, barrier", 3: "factor64 + W + stores"}
print("n = %d" % n); print("diag workgroup, block column POTRF_DEBUG_K (s_memtime shader cycles since kernel entry):")
prev = t[0]
for s in (1, 4, 5, 6, 7, 2, 3):
    print("  %-28s %8d  (+%d)" % (names[s], t[s] - t[0], t[s] - prev))
    prev = t[s]
print("  per wave, end of product 2:", [t[16 + w] - t[0] for w in range(4)], " F written:", [t[20 + w] - t[0] for w in range(4)], " (after the barrier before it:", [t[24 + w] - t[0] for w in range(4)], ")")
print("  factor64 sub-steps kb = 0..3 (cycles each):", [t[12] - t[8]] + [t[12 + q] - t[11 + q] for q in (1, 2, 3)])
print("  inside factor64, kb = 0:  factor16 (1 wave) %d, phase (b) %d, phase (c) %d" % (t[9] - t[8], t[10] - t[9], t[11] - t[10]))
PY
