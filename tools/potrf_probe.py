#!/usr/bin/env python3
"""Time dsvgp_potrf algo 0 (rocSOLVER) vs 1 (blocked, one fused MFMA launch per block column) on a C4-sized K_ZZ
(M'=3000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M, d, p = 500, 20, 5
hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
Z, V = torch.rand(M, d, device=dev), torch.eye(d, device=dev)[:p].repeat(M, 1)
pz = ops.pack_points(ctx, Z, V, p, hyp)
K = ops.kernel_fwd(ctx, pz, M, pz, M, d, p, hyp, jitter=1e-3, dtype=torch.float64)
info = torch.zeros(1, dtype=torch.int32, device=dev)
ALGOS, REPS = (0, 1, 0, 1), 6
print("probe: runs_algo1=%d reps=%d" % (ALGOS.count(1), REPS))      # read by tools/potrf_trace.sh
for algo in ALGOS:
    ts = []
    for rep in range(REPS):
        A = K.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.potrf_(ctx, A, info, algo); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("algo %d: %.2f ms (min %.2f) info=%d" % (algo, sorted(ts)[len(ts) // 2], min(ts), int(info.item())))
