#!/usr/bin/env python3
"""tools only: which k row of which operand goes wrong in the pipelined wide kernel (K = 256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops, L = dsvgp_amd._ops, dsvgp_amd._lib
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M, K = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 64 * 8192 + 192
Bh = (torch.arange(K).view(K, 1) * 1000.0 + torch.arange(N).view(1, N) % 192).float()
Bd = Bh.to(dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
bad_k = []
for kt in list(range(0, K, 7)) + [K - 1]:
    A = torch.zeros(K, M, dtype=torch.float64)
    A[kt, :] = 1.0
    ops.gemm(ctx, L.TRANS_A, A.to(dev), Bd, C, alpha=1.0)
    want = Bd[kt].double().view(1, N).expand(M, N)
    nb = int((C != want).sum())
    if nb:
        bad_k.append((kt, nb, C[0, 0].item(), C[0, 1].item(), C[5, 200].item()))
print("K =", K, "bad k rows:", bad_k[:20])
