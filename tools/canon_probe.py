#!/usr/bin/env python3
"""tools only (round 6): kernel assembly forward / backward of K_ZX with canonical data directions -- the general kernels (direction matrix) against
the canonical-direction kernels (index list) at a geometry CANON_GEOM="M,B,d,p" (default C4: 500,4096,20,5), for the library named by DSVGP_LIB_PATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
for geom in os.environ.get("CANON_GEOM", "500,4096,20,5;200,512,5,2;500,512,20,5").split(";"):
    M, B, d, p = (int(v) for v in geom.split(","))
    q = p + 1
    hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    Z, V = torch.rand(M, d, device=dev, generator=g), torch.randn(M * p, d, device=dev, generator=g)
    idx = sorted(torch.randperm(d)[:p].tolist())
    X, D = torch.rand(B, d, device=dev, generator=g), torch.eye(d, device=dev)[idx].repeat(B, 1)
    center = ops.column_mean(ctx, Z)
    pz, px = ops.pack_points(ctx, Z, V, p, hyp, center), ops.pack_points(ctx, X, D, p, hyp, center)
    di = (torch.tensor(idx, dtype=torch.int32) + 1).to(dev)
    # CANON_ROT output / upstream buffers used in turn (default 5: 1.5 GB at C4, well past the 256 MB memory-side cache -- every launch then
    # writes / reads memory that is not cached from the launch before, as in the step; CANON_ROT=1: the same buffer again and again)
    ROT = int(os.environ.get("CANON_ROT", "5"))
    outs = [torch.empty(M * q, B * q, device=dev) for _ in range(ROT)]
    Gs = [torch.randn(M * q, B * q, device=dev, generator=g) for _ in range(ROT)]
    out, G = outs[0], Gs[0]
    turn = [0]
    def nxt(lst):
        turn[0] += 1
        return lst[turn[0] % ROT]
    nbytes = out.numel() * 4 + (M + B) * d * q * 4

    def timeit(fn, n=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    dx, dv, dh = torch.zeros(M, d, device=dev), torch.zeros(M * p, d, device=dev), torch.zeros(4, device=dev)
    ws = torch.empty(int(dsvgp_amd._lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, B, d, p)), dtype=torch.uint8, device=dev)
    t_fg = timeit(lambda: ops.kernel_fwd(ctx, pz, M, px, B, d, p, hyp, out=nxt(outs)))
    ops.kernel_fwd(ctx, pz, M, px, B, d, p, hyp, out=out)
    Kg = out.clone()
    t_fc = timeit(lambda: ops.kernel_fwd_canon(ctx, pz, M, px, B, d, p, di, 1, hyp, out=nxt(outs)))
    ops.kernel_fwd_canon(ctx, pz, M, px, B, d, p, di, 1, hyp, out=out)
    err = ((out - Kg).abs().max() / Kg.abs().max()).item()
    t_bg = timeit(lambda: ops.kernel_bwd(ctx, nxt(Gs), pz, M, px, B, d, p, hyp, False, dx, dv, dh, ws))
    t_bc = timeit(lambda: ops.kernel_bwd_canon(ctx, nxt(Gs), pz, M, px, B, d, p, di, 1, hyp, dx, dv, dh, ws))
    tb = lambda ms: nbytes / ms / 1e9
    if os.environ.get("CANON_FILL", "1") == "1":       # what streaming writes of the same buffer cost on this box (the forward's floor)
        t_z = timeit(lambda: nxt(outs).zero_()); t_c = timeit(lambda: nxt(outs).copy_(nxt(Gs)))
        print("  fill of the %d x %d result: %.1f us (%.2f TB/s written)   copy: %.1f us (%.2f TB/s read + written)"
              % (M * q, B * q, t_z * 1e3, out.numel() * 4 / t_z / 1e9, t_c * 1e3, out.numel() * 8 / t_c / 1e9))
    print("M=%d B=%d d=%d p=%d (%.1f MB): fwd general %.1f us (%.2f TB/s = %.2f of 8)  canonical %.1f us (%.2f TB/s = %.2f)  |diff| %.1e ;  "
          "bwd general %.1f us (%.2f = %.2f)  canonical %.1f us (%.2f TB/s = %.2f)"
          % (M, B, d, p, nbytes / 1e6, t_fg * 1e3, tb(t_fg), tb(t_fg) / 8, t_fc * 1e3, tb(t_fc), tb(t_fc) / 8, err, t_bg * 1e3, tb(t_bg), tb(t_bg) / 8,
             t_bc * 1e3, tb(t_bc), tb(t_bc) / 8))
