#!/usr/bin/env python3
"""Time the kernel-assembly forward (K_ZX; geometry FWD_GEOM="M,B,d,p", default C4: 500,4096,20,5) for the library named by DSVGP_LIB_PATH."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M, B, d, p = (int(v) for v in os.environ.get("FWD_GEOM", "500,4096,20,5").split(","))
hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
Z, V = torch.rand(M, d, device=dev), torch.eye(d, device=dev)[:p].repeat(M, 1)
X, D = torch.rand(B, d, device=dev), torch.eye(d, device=dev)[:p].repeat(B, 1)
pz, px = ops.pack_points(ctx, Z, V, p, hyp), ops.pack_points(ctx, X, D, p, hyp)
out = torch.empty(M * (p + 1), B * (p + 1), device=dev)
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.kernel_fwd(ctx, pz, M, px, B, d, p, hyp, out=out)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("%s: K_ZX fwd %.1f us  %.2f TB/s" % (os.environ.get("DSVGP_LIB_PATH", "default"), ms * 1e3, out.numel() * 4 / ms / 1e9))
