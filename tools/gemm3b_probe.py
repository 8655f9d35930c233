"""tools only: the fp32-equivalent bf16 x 3 GEMM (csrc/gemm3b.hip) against the fp32 MFMA kernel (gemm32.hip) on the two big fp32
products of the C4 step -- time and error against a float64 product of the same float32 operands (GPU box).
usage: python3 tools/gemm3b_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsvgp_amd
from dsvgp_amd import _lib, _ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
ctx = _ops.Context.get(dev)
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def shrink(C, ref, lower=False):
    """mean signed error along the sign of the exact value, relative to the mean magnitude: a biased (truncating) accumulation
    shows here, an unbiased one averages out"""
    C = C.double()
    if lower:
        C, ref = C.tril(), ref.tril()
    return (((C - ref) * torch.sign(ref)).sum() / ref.abs().sum()).item()


def relerr(C, ref, lower=False):
    C, ref = C.double(), ref
    if lower:
        C, ref = C.tril(), ref.tril()
    return ((C - ref).abs().max() / ref.abs().max()).item()


for name, M, N, K in (("dense K_ZX-bar  [3000 x 3001] x [3001 x 24576]", 3000, 24576, 3001),
                      ("dense, 8-rank share          N = 3072", 3000, 3072, 3001),
                      ("small ragged                 ", 700, 1100, 1300)):
    Kp4 = (K + 3) // 4 * 4
    A = torch.zeros(M, Kp4, device=dev); A[:, :K] = torch.randn(M, K, device=dev, generator=g)
    B = torch.randn(K, N, device=dev, generator=g)
    ref = A[:, :K].double() @ B.double()
    C32 = torch.empty(M, N, device=dev); C3b = torch.empty(M, N, device=dev)
    t32 = timeit(lambda: _ops.gemm(ctx, _lib.K_PADDED, A[:, :K], B, C32))
    pa = _ops.split3_bf16(ctx, A[:, :K]); pb = _ops.split3_bf16(ctx, B, transpose=True)
    ts = timeit(lambda: (_ops.split3_bf16(ctx, A[:, :K], out=pa), _ops.split3_bf16(ctx, B, transpose=True, out=pb)))
    t3b = timeit(lambda: _ops.gemm3b(ctx, 0, M, N, K, pa, M, pb, N, C3b))
    fl = 2.0 * M * N * K
    print("%s  fp32 MFMA %.3f ms %.1f TF err %.2e bias %+.1e | bf16x3 %.3f ms %.1f TF-equivalent err %.2e bias %+.1e | split passes %.3f ms"
          % (name, t32, fl / t32 / 1e9, relerr(C32, ref), shrink(C32, ref), t3b, fl / t3b / 1e9, relerr(C3b, ref), shrink(C3b, ref), ts))
for name, M, K in (("Gram tril([A;mu]A^T)  K = 24576", 3001, 24576), ("Gram, 8-rank share    K = 3072", 3001, 3072)):
    A = torch.randn(M, K, device=dev, generator=g)
    ref = A.double() @ A[:M - 1].double().t()
    C32 = torch.empty(M, M - 1, device=dev); C3b = torch.empty(M, M - 1, device=dev)
    t32 = timeit(lambda: _ops.gemm(ctx, _lib.TRANS_B | _lib.OUT_LOWER, A, A[:M - 1], C32))
    pa = _ops.split3_bf16(ctx, A)
    ts = timeit(lambda: _ops.split3_bf16(ctx, A, out=pa))
    t3b = timeit(lambda: _ops.gemm3b(ctx, _lib.OUT_LOWER, M, M - 1, K, pa, M, pa, M, C3b))
    fl = 1.0 * M * (M - 1) * K
    up = C3b.triu(1).abs().max().item()
    print("%s  fp32 MFMA %.3f ms %.1f TF err %.2e bias %+.1e | bf16x3 %.3f ms %.1f TF-equivalent err %.2e bias %+.1e (upper %s) | split pass %.3f ms"
          % (name, t32, fl / t32 / 1e9, relerr(C32, ref, True), shrink(C32, ref, True), t3b, fl / t3b / 1e9, relerr(C3b, ref, True),
             shrink(C3b, ref, True), "zero" if up == 0 else "NONZERO", ts))
