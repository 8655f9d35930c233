#!/usr/bin/env python3
"""Time the kernel-assembly backward (K_ZX-bar, fp32 upstream; K_ZZ-bar fp64) for the library named by DSVGP_LIB_PATH.
Geometry from BWD_GEOM="M,B,d,p" (default C4: 500,4096,20,5; C3: 300,512,10,10)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M, B, d, p = (int(v) for v in os.environ.get("BWD_GEOM", "500,4096,20,5").split(","))
hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
Z, V = torch.rand(M, d, device=dev), torch.eye(d, device=dev)[:p].repeat(M, 1)
X, D = torch.rand(B, d, device=dev), torch.eye(d, device=dev)[:p].repeat(B, 1)
pz, px = ops.pack_points(ctx, Z, V, p, hyp), ops.pack_points(ctx, X, D, p, hyp)
q = p + 1
for name, G, p2, n2, sym in (("K_ZX-bar f32", torch.randn(M * q, B * q, device=dev), px, B, False),
                             ("K_ZZ-bar f64", torch.randn(M * q, M * q, device=dev, dtype=torch.float64), pz, M, True)):
    dx, dv, dh = torch.zeros(M, d, device=dev), torch.zeros(M * p, d, device=dev), torch.zeros(4, device=dev)
    ws = torch.empty(int(dsvgp_amd._lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, n2, d, p)), dtype=torch.uint8, device=dev)
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.kernel_bwd(ctx, G, pz, M, p2, n2, d, p, hyp, sym, dx, dv, dh, ws)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s: %s bwd %.1f us  %.2f TB/s" % (os.environ.get("DSVGP_LIB_PATH", "default"), name, ms * 1e3,
                                             G.numel() * G.element_size() / ms / 1e9))
