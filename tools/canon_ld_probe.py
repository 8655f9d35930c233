#!/usr/bin/env python3
"""tools only (round 6): does the leading dimension of the result matter to the canonical forward assembly?  K_ZX at C4 (3000 x 24576 fp32) written
into views of wider buffers (row stride 24576 + pad floats), five buffers in turn (memory that is not cached), against a plain fill of the same view."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0)
ctx = ops.Context.get(dev)
M, B, d, p = 500, 4096, 20, 5
q = p + 1
hyp = torch.tensor([0.69, 0.69, 0.1, 0.0], device=dev)
g = torch.Generator(device=dev).manual_seed(0)
Z, V = torch.rand(M, d, device=dev, generator=g), torch.randn(M * p, d, device=dev, generator=g)
idx = sorted(torch.randperm(d)[:p].tolist())
X, D = torch.rand(B, d, device=dev, generator=g), torch.eye(d, device=dev)[idx].repeat(B, 1)
center = ops.column_mean(ctx, Z)
pz, px = ops.pack_points(ctx, Z, V, p, hyp, center), ops.pack_points(ctx, X, D, p, hyp, center)
di = (torch.tensor(idx, dtype=torch.int32) + 1).to(dev)
nbytes = M * q * B * q * 4


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for pad in [int(v) for v in os.environ.get("PADS", "0,16,32,64,256,1024,4096").split(",")]:
    bufs = [torch.empty(M * q, B * q + pad, device=dev) for _ in range(5)]
    views = [b[:, :B * q] for b in bufs]
    k = [0]
    def nxt():
        k[0] += 1
        return views[k[0] % 5]
    t_c = timeit(lambda: ops.kernel_fwd_canon(ctx, pz, M, px, B, d, p, di, 1, hyp, out=nxt()))
    t_g = timeit(lambda: ops.kernel_fwd(ctx, pz, M, px, B, d, p, hyp, out=nxt()))
    t_z = timeit(lambda: nxt().zero_())
    print("row stride %6d floats (%d B): canonical %.1f us (%.2f TB/s = %.2f of 8)   general %.1f us   zero_ of the view %.1f us"
          % (B * q + pad, (B * q + pad) * 4, t_c, nbytes / t_c / 1e6, nbytes / t_c / 1e6 / 8, t_g, t_z))
