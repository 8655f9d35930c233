#!/usr/bin/env python3
"""tools only: placement check of the pipelined wide fp64 kernel (gemm64p): C = op(A) B with op(A)[m][k] = A[k][m]; identity-like operands show
which element of which operand lands where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops, L = dsvgp_amd._ops, dsvgp_amd._lib
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M = K = 64
N = 64 * 8192 + 192
def run(A, B):
    C = torch.full((M, N), float("nan"), dtype=torch.float64, device=dev)
    ops.gemm(ctx, L.TRANS_A, A.to(dev), B.to(dev), C, alpha=1.0)
    return C.cpu()
# 1. A = I: C[m][n] = B[m][n]
B = (torch.arange(K).view(K, 1) * 1000.0 + torch.arange(N).view(1, N) % 192).float()
C = run(torch.eye(K, dtype=torch.float64), B)
bad = (C != B.double())
print("A = I: wrong elements", int(bad.sum()), "of", C.numel())
if bad.any():
    idx = bad.nonzero()[:12]
    for m, n in idx.tolist():
        print("  C[%d][%d] = %s, want %s" % (m, n, C[m, n].item(), B[m, n].item()))
# 2. B = [I | 0 ...]: C[m][n < 64] = A[n][m]
A = (torch.arange(K).view(K, 1) * 1000.0 + torch.arange(M).view(1, M)).double()
B = torch.zeros(K, N)
B[:, :K] = torch.eye(K)
C = run(A, B)
want = torch.zeros(M, N, dtype=torch.float64)
want[:, :K] = A.t()
bad = (C != want)
print("B = [I 0]: wrong elements", int(bad.sum()))
if bad.any():
    idx = bad.nonzero()[:12]
    for m, n in idx.tolist():
        print("  C[%d][%d] = %s, want %s" % (m, n, C[m, n].item(), want[m, n].item()))
# 3. random operands at a few shapes (ld = shape + pad)
g = torch.Generator().manual_seed(0)
for (M_, K_, padA, padB) in [(64, 128, 0, 0), (64, 144, 0, 0), (64, 256, 0, 0), (64, 272, 0, 0), (64, 512, 0, 0), (64, 592, 0, 0), (64, 601, 0, 0), (64, 1024, 0, 0)]:
    N_ = 64 * 8192 * 64 // ((M_ + 63) // 64 * 64) + 192
    A = torch.randn(K_, M_ + padA, generator=g, dtype=torch.float64)[:, :M_]
    B = torch.randn(K_, N_ + padB, generator=g)[:, :N_]
    Ad = torch.empty(K_, M_ + padA, dtype=torch.float64, device=dev)[:, :M_]; Ad.copy_(A)
    Bd = torch.empty(K_, N_ + padB, dtype=torch.float32, device=dev)[:, :N_]; Bd.copy_(B)
    C = torch.full((M_, N_), float("nan"), dtype=torch.float64, device=dev)
    ops.gemm(ctx, L.TRANS_A, Ad, Bd, C, alpha=1.0)
    ref = Ad.t() @ Bd.double()
    err = (C - ref).abs()
    nanc = int(torch.isnan(C).sum())
    print("nan in C:", nanc, " first wrong:", (err > 1e-9).nonzero()[:3].tolist(), C[0, :3].tolist(), ref[0, :3].tolist())
    print("M %d K %d padA %d padB %d: max err %.2e; bad rows %s..., bad cols(mod 192) %s" % (M_, K_, padA, padB, err.max().item(),
          (err.max(1).values > 1e-9).nonzero().flatten()[:6].tolist(), sorted(set(((err.max(0).values > 1e-9).nonzero().flatten() % 192).tolist()))[:12]))
