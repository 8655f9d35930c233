"""tools only: the M' x M' x M' fp64 products of the ELBO fast path / Cholesky backward, each timed alone (GPU box).
usage: python tools/m3_probe.py [n] [reps]      (DSVGP_LIB_PATH selects a variant build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
from dsvgp_amd import _lib, _ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
ctx = _ops.Context.get(dev)
f64, f32 = torch.float64, torch.float32
TA, AU, AL, BL, OL = _lib.TRANS_A, _lib.A_UPPER, _lib.A_LOWER, _lib.B_LOWER, _lib.OUT_LOWER
g = torch.Generator(device=dev).manual_seed(0)
L = torch.tril(torch.randn(n, n, dtype=f64, device=dev, generator=g)) / n ** 0.5 + torch.eye(n, dtype=f64, device=dev)
Lbar = torch.tril(torch.randn(n, n, dtype=f64, device=dev, generator=g))
S = torch.randn(n, n, dtype=f64, device=dev, generator=g); S = S + S.t()
Linv = torch.tril(torch.randn(n, n, dtype=f64, device=dev, generator=g))
Qe = torch.randn(n, n + 1, dtype=f64, device=dev, generator=g)
Ge = torch.randn(n + 1, n, dtype=f32, device=dev, generator=g)
out = torch.empty(n, n, dtype=f64, device=dev)
cases = [
    ("tril(L^T Lbar)      A_UPPER B_LOWER OUT_LOWER", TA | AU | BL | OL, L, Lbar, n ** 3 / 6),
    ("tril(S L^-1)        B_LOWER OUT_LOWER        ", TA | BL | OL, S, Linv, n ** 3 / 3),
    ("S L^-1              B_LOWER                  ", TA | BL, S, Linv, n ** 3 / 2),
    ("tril(L^-T Y)        A_UPPER OUT_LOWER        ", TA | AU | OL, Linv, S, n ** 3 / 6),
    ("L^-T X (Q')         A_UPPER                  ", TA | AU, Linv, S, n ** 3 / 2),
    ("tril(Qe Ge) (L-bar) OUT_LOWER, f64 x f32     ", OL, Qe, Ge, n ** 3 / 2),
]
for name, fl, A, B, macs in cases:
    for _ in range(3):
        _ops.gemm(ctx, fl, A, B, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _ops.gemm(ctx, fl, A, B, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%s n=%d  %.1f us  %.1f TFLOP/s (algorithmic 2 x %.2e MAC)" % (name, n, ms * 1e3, 2 * macs / ms / 1e9, macs))
