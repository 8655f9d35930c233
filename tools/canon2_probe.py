#!/usr/bin/env python3
"""tools only (round 6): kernel assembly with one-hot directions on both sides (the full-gradient SVGP) at the C3 geometry (M = 300, B = 512,
d = p = 10): K_ZX forward / backward (float) and K_ZZ forward (double out) / backward (double upstream, symmetric) -- the general kernels against
the canon2 kernels, for the library named by DSVGP_LIB_PATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M, B, d = (int(v) for v in os.environ.get("CANON2_GEOM", "300,512,10").split(","))
p, q = d, d + 1
hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
g = torch.Generator(device=dev).manual_seed(0)
Z, X = torch.rand(M, d, device=dev, generator=g), torch.rand(B, d, device=dev, generator=g)
center = ops.column_mean(ctx, Z)
pz = ops.pack_points(ctx, Z, torch.eye(d, device=dev).repeat(M, 1), p, hyp, center)
px = ops.pack_points(ctx, X, torch.eye(d, device=dev).repeat(B, 1), p, hyp, center)
di = torch.arange(d, dtype=torch.int32, device=dev)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, n2, p2, dt, sym in (("K_ZX float", B, px, torch.float32, False), ("K_ZZ double", M, pz, torch.float64, True)):
    out = torch.empty(M * q, n2 * q, device=dev, dtype=dt)
    G = torch.randn(M * q, n2 * q, device=dev, generator=g).to(dt)
    if sym:
        G = (0.5 * (G + G.t())).contiguous()
    nbytes = out.numel() * out.element_size()
    ws = torch.empty(int(dsvgp_amd._lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, n2, d, p)), dtype=torch.uint8, device=dev)
    dx, dv, dh = torch.zeros(M, d, device=dev), torch.zeros(M * p, d, device=dev), torch.zeros(4, device=dev)
    t_fg = timeit(lambda: ops.kernel_fwd(ctx, pz, M, p2, n2, d, p, hyp, jitter=1e-3 if sym else 0.0, out=out, dtype=dt))
    Kg = out.clone()
    t_f2 = timeit(lambda: ops.kernel_fwd_canon2(ctx, pz, M, p2, n2, d, p, di, 0, hyp, jitter=1e-3 if sym else 0.0, out=out))
    err = ((out - Kg).abs().max() / Kg.abs().max()).item()
    t_bg = timeit(lambda: ops.kernel_bwd(ctx, G, pz, M, p2, n2, d, p, hyp, sym, dx, dv, dh, ws))
    t_b2 = timeit(lambda: ops.kernel_bwd_canon2(ctx, G, pz, M, p2, n2, d, p, di, 0, hyp, sym, dx, dv, dh, ws))
    tb = lambda us: nbytes / us / 1e6
    print("%s %d x %d (%.1f MB): fwd general %.1f us (%.2f TB/s)  canon2 %.1f us (%.2f TB/s = %.2f of 8)  |diff| %.1e ;  bwd general %.1f us (%.2f)  "
          "canon2 %.1f us (%.2f TB/s = %.2f of 8)" % (name, M * q, n2 * q, nbytes / 1e6, t_fg, tb(t_fg), t_f2, tb(t_f2), tb(t_f2) / 8, err,
                                                    t_bg, tb(t_bg), t_b2, tb(t_b2), tb(t_b2) / 8))
