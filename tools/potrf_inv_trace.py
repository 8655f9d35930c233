#!/usr/bin/env python3
"""dsvgp_potrf_inverse at one size (argv[1], default 3000), argv[2] repetitions (default 3): the workload of tools/potrf_inv_trace.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device=dev).manual_seed(1)
Q = torch.randn(n, n, device=dev, dtype=torch.float64, generator=g)
K = Q @ Q.t() / n + torch.eye(n, device=dev, dtype=torch.float64)
info = torch.zeros(1, dtype=torch.int32, device=dev)
ws = ops.trsm_workspace(n, n + 1, 4096, dev)
pws = ops.potrf_workspace(n, dev)
for rep in range(reps):
    A = K.clone()
    ops.potrf_inverse_(ctx, A, info, 4096, ws, pws)
    torch.cuda.synchronize()
print("probe: n=%d reps=%d info=%d" % (n, reps, int(info.item())))
