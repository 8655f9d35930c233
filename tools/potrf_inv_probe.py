#!/usr/bin/env python3
"""Time + check dsvgp_potrf_inverse (blocked Cholesky with the fused inverse: the chain of the training step) at
M' = 3000 (C4), 3300 (C3) and 600 (C2) for the library named by DSVGP_LIB_PATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
out = []
for M, d, p in ((500, 20, 5), (300, 10, 10), (200, 5, 2)):
    hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    Z = torch.rand(M, d, device=dev, generator=g)
    V = torch.eye(d, device=dev)[:p].repeat(M, 1)
    pz = ops.pack_points(ctx, Z, V, p, hyp)
    K = ops.kernel_fwd(ctx, pz, M, pz, M, d, p, hyp, jitter=1e-3, dtype=torch.float64)
    n = K.shape[0]
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = 4096
    ws = ops.trsm_workspace(n, n + 1, nb, dev)
    pws = ops.potrf_workspace(n, dev)
    ts = []
    pad = int(os.environ.get("PROBE_LDA_PAD", "0"))      # leading dimension rounded up to a multiple of this many doubles (0: n)
    ldp = (n + pad - 1) // pad * pad if pad else n
    buf = torch.empty(n, ldp, dtype=torch.float64, device=dev)
    for rep in range(8):
        A = buf[:, :n]
        A.copy_(K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.potrf_inverse_(ctx, A, info, nb, ws, pws); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    L = torch.tril(A)
    Linv = torch.tril(ws[:n * n * 8].view(torch.float64).view(n, n))
    r1 = ((L @ L.t()).tril() - K.tril()).abs().max().item() / K.abs().max().item()
    r2 = (Linv @ L - torch.eye(n, dtype=torch.float64, device=dev)).abs().max().item()
    out.append("n=%d: %.3f ms (min %.3f) info=%d |LL^T-K|=%.1e |L^-1 L - I|=%.1e" % (n, sorted(ts)[len(ts) // 2], min(ts), int(info.item()), r1, r2))
print("%s: %s" % (os.path.basename(os.path.dirname(os.environ.get("DSVGP_LIB_PATH", "default/x"))), "  ".join(out)))
