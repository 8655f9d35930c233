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
buf = (ctypes.c_ulonglong * 64)()
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
if t[28] or t[29] or t[30] or t[31]:      # round 6 (POTRF_CRIT_STRIPS): the register row-strip chain
    print("  strips: chain start %d; diagonal wave kb done at (cycles since chain start): %s" % (t[8] - t[0], [t[12 + q] - t[8] for q in range(4)]))
    print("  strips: end of each 4-column block of the diagonal waves, cycles since chain start:", [[t[32 + 4 * q + j] - t[8] for j in range(4)] for q in range(4)])
    print("  strips: chains start at %s; the next diagonal wave has followed at %s, has its trailing update at %s (cycles since chain start)"
          % ([t[48 + q] - t[8] for q in range(4)], [t[52 + q] - t[8] for q in range(3)], [t[56 + q] - t[8] for q in range(3)]))
    print("  strips: wave 2 following block column 1: flag of block jb seen at %s, block done at %s, panel block published at %d (cycles since chain start)"
          % ([t[16 + j] - t[8] for j in range(4)], [t[20 + j] - t[8] for j in range(4)], t[24] - t[8]))
    ce = t[15]
    print("  strips: tail, cycles since the END of the last chain: wave 0: T_30 formed %d, X_33 seen %d, X_30 published %d, row 3 complete %d; wave 3: L stored %d, row 3 complete %d, W accumulated %d; wave 1: T_31 %d, X_31 %d; wave 2: T_32 %d, X_32 %d"
          % (t[60] - ce, t[61] - ce, t[62] - ce, t[63] - ce, t[25] - ce, t[26] - ce, t[27] - ce, t[59] - ce, t[56] - ce, t[58] - ce, t[57] - ce))
    print("  strips: per block column (cycles): %s; X_33 published +%d after the chain; waves done at %s (since kernel entry)"
          % ([t[12] - t[8]] + [t[12 + q] - t[11 + q] for q in (1, 2, 3)], t[9] - t[15], [t[28 + w] - t[0] for w in range(4)]))
else:
    print("  factor64 sub-steps kb = 0..3 (cycles each):", [t[12] - t[8]] + [t[12 + q] - t[11 + q] for q in (1, 2, 3)])
    print("  inside factor64, kb = 0:  factor16 (1 wave) %d, phase (b) %d, phase (c) %d" % (t[9] - t[8], t[10] - t[9], t[11] - t[10]))
PY
