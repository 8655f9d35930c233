import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0); ctx = ops.Context.get(dev)
g = torch.Generator().manual_seed(0)
n = 3000
Q = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (Q @ Q.t() / n + 1e-3 * torch.eye(n, dtype=torch.float64))
Lref = torch.linalg.cholesky(K)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for algo in (0, 1):
    A = K.to(dev).clone(); ops.potrf_(ctx, A, info, algo); torch.cuda.synchronize()
    L = A.tril().cpu()
    print("algo", algo, "max|L-Lref|/max|Lref| = %.3e" % ((L - Lref).abs().max() / Lref.abs().max()).item(),
          "resid |LL^T-K|/|K| = %.3e" % ((L @ L.t() - K).abs().max() / K.abs().max()).item())

# the DSVGP K_ZZ itself (C4 geometry, fp64 assembly + 1e-3 jitter)
M, d, p = 500, 20, 5
hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
Z, V = torch.rand(M, d, device=dev), torch.eye(d, device=dev)[:p].repeat(M, 1)
pz = ops.pack_points(ctx, Z, V, p, hyp)
Kz = ops.kernel_fwd(ctx, pz, M, pz, M, d, p, hyp, jitter=1e-3, dtype=torch.float64)
Kc = Kz.cpu(); Lref = torch.linalg.cholesky(Kc)
for algo in (0, 1):
    A = Kz.clone(); ops.potrf_(ctx, A, info, algo); torch.cuda.synchronize()
    L = A.tril().cpu()
    print("K_ZZ algo", algo, "info", int(info.item()), "max|L-Lref|/max|Lref| = %.3e" % ((L - Lref).abs().max() / Lref.abs().max()).item(),
          "resid |LL^T-K|/|K| = %.3e" % ((L @ L.t() - Kc).abs().max() / Kc.abs().max()).item())
