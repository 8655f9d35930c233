"""GPU parity tests, op level: every C-ABI entry point against the CPU oracle / plain torch fp64.

Tolerances (stated per test) are against fp64 ground truth; the HIP kernels compute in fp32 with the
same quadratic-expansion arithmetic as the reference (rel. error of kernel entries ~1e-6), the
Cholesky / triangular-solve path in fp64."""
import glob
import os

import numpy as np
import pytest
import torch

import dsvgp_oracle as O

pytestmark = pytest.mark.gpu
from _golden import GOLDEN, kernel_error


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _hyp(dev, ell, s=1.0, noise=0.1):
    return torch.tensor([ell, s, noise, 0.0], dtype=torch.float32, device=dev)


def _kernel_gpu(dsvgp, dev, x1, x2, v1, v2, ell, s=1.0, jitter=0.0, dtype=torch.float32):
    ops = dsvgp._ops
    ctx = ops.Context.get(dev)
    n1, d = x1.shape
    n2 = x2.shape[0]
    p = v1.shape[0] // n1
    hyp = _hyp(dev, ell, s)
    x1d = x1.float().to(dev).contiguous()
    center = ops.column_mean(ctx, x1d)
    p1 = ops.pack_points(ctx, x1d, v1.float().to(dev).contiguous(), p, hyp, center)
    p2 = ops.pack_points(ctx, x2.float().to(dev).contiguous(), v2.float().to(dev).contiguous(), p, hyp, center)
    return ops.kernel_fwd(ctx, p1, n1, p2, n2, d, p, hyp, jitter=jitter, dtype=dtype), (ctx, hyp, p1, p2)


# ------------------------------------------------------------------ kernel assembly forward
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_kernel_fwd_matches_reference_golden_vectors(dsvgp, gpu_device, path):
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k])
    K, _ = _kernel_gpu(dsvgp, gpu_device, t("x1"), t("x2"), t("v1"), t("v2"), float(g["lengthscale"]))
    e_sub, e_sum = kernel_error(K, g)
    print("kernel_fwd %s: HIP fp32 error %.2e (row/col sums %s); reference's own fp32 round-off %.2e"
          % (os.path.basename(path), e_sub, "%.2e" % e_sum if e_sum is not None else "-", float(g["ref_fp32_relerr"])))
    assert e_sub < 2e-6 and (e_sum is None or e_sum < 2e-6)        # fp32 arithmetic vs the reference run in fp64 (observed <= 4.1e-7,
                                                                   # the reference's own fp32 run: 1e-7 ... 7.9e-7)
    if "Kdiag" in g:                                               # diag=True branch (:110-119) for every symmetric case
        ops = dsvgp._ops
        ctx = ops.Context.get(gpu_device)
        dg = ops.kernel_diag(ctx, g["x1"].shape[0], int(g["p"]), _hyp(gpu_device, float(g["lengthscale"])))
        assert relmax(dg, t("Kdiag")) < 1e-6
        assert relmax(torch.diagonal(K), t("Kdiag")) < 2e-6
        # fp64 output of the same assembly (what feeds the Cholesky), + jitter on the diagonal only
        K64, _ = _kernel_gpu(dsvgp, gpu_device, t("x1"), t("x2"), t("v1"), t("v2"), float(g["lengthscale"]), jitter=1e-3,
                             dtype=torch.float64)
        assert relmax(K64 - 1e-3 * torch.eye(K64.shape[0], dtype=torch.float64, device=gpu_device), K) < 1e-6


@pytest.mark.parametrize("n1,n2,d,p", [(37, 53, 5, 2), (16, 16, 20, 5), (33, 70, 20, 5), (9, 130, 3, 0),
                                       (20, 11, 10, 10), (7, 40, 45, 1), (130, 7, 2, 1),
                                       # several tiles of the split-row kernel (q = 11) and of the KSM = 16 pair kernels (q = 6 / 3, 28 < d <= 60)
                                       (50, 95, 10, 10), (30, 60, 50, 5), (20, 70, 40, 2)])
def test_kernel_fwd_random_shapes(dsvgp, gpu_device, n1, n2, d, p):
    g = torch.Generator().manual_seed(n1 * 1000 + n2)
    x1, x2 = torch.rand(n1, d, generator=g), torch.rand(n2, d, generator=g)
    v1 = torch.randn(max(n1 * p, 0), d, generator=g)
    v2 = torch.randn(max(n2 * p, 0), d, generator=g)
    ell, s = 0.9, 1.7
    K, _ = _kernel_gpu(dsvgp, gpu_device, x1, x2, v1, v2, ell, s)
    ref = s * O.kernel_matrix(x1.double(), x2.double(), v1.double().reshape(n1 * p, d), v2.double().reshape(n2 * p, d), ell)
    assert relmax(K, ref) < 2e-5


# ------------------------------------------------------------------ canonical (one-hot, shared) directions on side 2: K_ZX as the reference's callers build it
def _canon_case(dsvgp, dev, x1, x2, v1, idx, ell, s, idx_base):
    """packs (x1, v1) and (x2, E[idx] tiled); returns ctx, hyp, packs and the device index list (+ idx_base)"""
    ops = dsvgp._ops
    ctx = ops.Context.get(dev)
    n1, d = x1.shape
    n2 = x2.shape[0]
    p = len(idx)
    hyp = _hyp(dev, ell, s)
    x1d = x1.float().to(dev).contiguous()
    center = ops.column_mean(ctx, x1d)
    v2 = torch.eye(d)[idx].repeat(n2, 1)
    p1 = ops.pack_points(ctx, x1d, v1.float().to(dev).contiguous(), p, hyp, center)
    p2 = ops.pack_points(ctx, x2.float().to(dev).contiguous(), v2.to(dev).contiguous(), p, hyp, center)
    di = (torch.tensor(idx, dtype=torch.int32) + idx_base).to(dev)
    return ops, ctx, hyp, p1, p2, di, v2


@pytest.mark.parametrize("name", ["kernel_c_onehot", "kernel_e_c2shape", "kernel_g_c4tiles", "kernel_h_c2tiles"])
def test_kernel_fwd_canon_matches_reference_golden_vectors(dsvgp, gpu_device, name):
    """the golden vectors of the reference's kernel file whose v2 is one-hot and the same for every point of side 2, through the
    canonical-direction kernel (index list instead of a direction matrix): same 2e-6 as the general kernel"""
    path = [q for q in GOLDEN if os.path.basename(q) == name + ".npz"][0]
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k])
    x2, v2 = t("x2"), t("v2")
    n2, d = x2.shape
    p = v2.shape[0] // n2
    idx = v2.reshape(n2, p, d).argmax(2)
    assert bool((idx == idx[0]).all()) and bool(((v2 == 0) | (v2 == 1)).all())
    ops, ctx, hyp, p1, p2, di, _ = _canon_case(dsvgp, gpu_device, t("x1"), x2, t("v1"), idx[0].tolist(), float(g["lengthscale"]), 1.0, 1)
    K = ops.kernel_fwd_canon(ctx, p1, g["x1"].shape[0], p2, n2, d, p, di, 1, hyp)
    e_sub, e_sum = kernel_error(K, g)
    print("kernel_fwd_canon %s: HIP fp32 error %.2e (row/col sums %s)" % (name, e_sub, "%.2e" % e_sum if e_sum is not None else "-"))
    assert e_sub < 2e-6 and (e_sum is None or e_sum < 2e-6)


@pytest.mark.parametrize("n1,n2,d,p,base", [(37, 53, 5, 2, 0), (16, 16, 20, 5, 1), (33, 70, 20, 5, 0), (130, 7, 5, 2, 1), (9, 300, 28, 5, 1),
                                            (50, 131, 9, 2, 0), (24, 8, 20, 5, 1)])
def test_kernel_fwd_canon_equals_the_general_kernel(dsvgp, gpu_device, n1, n2, d, p, base):
    g = torch.Generator().manual_seed(n1 * 1000 + n2 + d)
    x1, x2 = torch.rand(n1, d, generator=g), torch.rand(n2, d, generator=g)
    v1 = torch.randn(n1 * p, d, generator=g)
    idx = sorted(torch.randperm(d, generator=g)[:p].tolist())
    ell, s = 0.9, 1.7
    ops, ctx, hyp, p1, p2, di, v2 = _canon_case(dsvgp, gpu_device, x1, x2, v1, idx, ell, s, base)
    assert ops.canon_supported(d, p)
    Kc = ops.kernel_fwd_canon(ctx, p1, n1, p2, n2, d, p, di, base, hyp)
    Kg = ops.kernel_fwd(ctx, p1, n1, p2, n2, d, p, hyp)
    ref = s * O.kernel_matrix(x1.double(), x2.double(), v1.double(), v2.double(), ell)
    assert relmax(Kc, ref) < 2e-5 and relmax(Kc, Kg) < 5e-6, (relmax(Kc, ref), relmax(Kc, Kg))
    # a view with a padded leading dimension (rows not 16-byte aligned: the per-lane store path)
    buf = torch.zeros(n1 * (p + 1), n2 * (p + 1) + 3, device=gpu_device)
    Kv = ops.kernel_fwd_canon(ctx, p1, n1, p2, n2, d, p, di, base, hyp, out=buf[:, 1:n2 * (p + 1) + 1])
    assert torch.equal(Kv, Kc) and buf[:, 0].abs().max().item() == 0.0 and buf[:, -2:].abs().max().item() == 0.0


@pytest.mark.parametrize("n1,n2,d,p", [(11, 23, 5, 2), (20, 45, 20, 5), (40, 300, 20, 5), (70, 130, 5, 2), (9, 9, 28, 5), (64, 64, 12, 2)])
def test_kernel_bwd_canon_matches_autograd_and_the_general_kernel(dsvgp, gpu_device, n1, n2, d, p):
    g = torch.Generator().manual_seed(n1 + 31 * n2 + d)
    x1, x2 = torch.rand(n1, d, generator=g), torch.rand(n2, d, generator=g)
    v1 = torch.randn(n1 * p, d, generator=g)
    idx = sorted(torch.randperm(d, generator=g)[:p].tolist())
    ell, s = 0.8, 1.3
    q = p + 1
    G = torch.randn(n1 * q, n2 * q, generator=g, dtype=torch.float64)
    dev = gpu_device
    ops, ctx, hyp, p1, p2, di, v2 = _canon_case(dsvgp, dev, x1, x2, v1, idx, ell, s, 1)
    x1r = x1.double().requires_grad_(True)
    v1r = v1.double().requires_grad_(True)
    ellr = torch.tensor(ell, dtype=torch.float64, requires_grad=True)
    sr = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    (sr * O.kernel_matrix(x1r, x2.double(), v1r, v2.double(), ellr) * G).sum().backward()
    for Gd in (G.to(dev), G.float().to(dev)):                 # double and float upstream gradients
        res = []
        for canon in (True, False):
            dx = torch.zeros(n1, d, device=dev)
            dv = torch.zeros(n1 * p, d, device=dev)
            dh = torch.zeros(4, device=dev)
            if canon:
                ops.kernel_bwd_canon(ctx, Gd.contiguous(), p1, n1, p2, n2, d, p, di, 1, hyp, dx, dv, dh)
            else:
                ops.kernel_bwd(ctx, Gd.contiguous(), p1, n1, p2, n2, d, p, hyp, False, dx, dv, dh)
            res.append((dx, dv, dh))
        (dx, dv, dh), (gx, gv, gh) = res
        assert relmax(dx, x1r.grad) < 2e-4 and relmax(dv, v1r.grad) < 2e-4, (relmax(dx, x1r.grad), relmax(dv, v1r.grad))
        assert abs(dh[0].item() - ellr.grad.item()) < 2e-4 * max(1.0, abs(ellr.grad.item()))
        assert abs(dh[1].item() - sr.grad.item()) < 2e-4 * max(1.0, abs(sr.grad.item()))
        assert relmax(dx, gx) < 1e-4 and relmax(dv, gv) < 1e-4


def test_kernel_canon_rejects_geometries_it_does_not_take(dsvgp, gpu_device):
    ops = dsvgp._ops
    assert not ops.canon_supported(50, 5) and not ops.canon_supported(10, 10) and ops.canon_supported(28, 5) and ops.canon_supported(5, 2)
    g = torch.Generator().manual_seed(1)
    x1, x2, v1 = torch.rand(6, 10, generator=g), torch.rand(7, 10, generator=g), torch.randn(6 * 3, 10, generator=g)
    o, ctx, hyp, p1, p2, di, _ = _canon_case(dsvgp, gpu_device, x1, x2, v1, [0, 4, 7], 0.7, 1.0, 0)
    with pytest.raises(dsvgp._lib.DsvgpError):
        o.kernel_fwd_canon(ctx, p1, 6, p2, 7, 10, 3, di, 0, hyp)       # q = 4


# ------------------------------------------------------------------ one-hot directions on both sides (the full-gradient SVGP, BASELINE config 3)
def _rbf_grad_closed_form(x1, x2, ell, s):
    """gpytorch RBFKernelGrad's closed form in the interleaved layout (reference GradVariationalStrategy.py:89-99 uses it on
    cat([Z, x])): block (i, j) = k [[1, delta^T / ell], [-delta / ell, (I - delta delta^T) / ell^2]], delta = (x1_i - x2_j) / ell"""
    n1, d = x1.shape
    n2 = x2.shape[0]
    dl = (x1[:, None, :] - x2[None, :, :]) / ell
    k = s * torch.exp(-0.5 * (dl ** 2).sum(-1))
    K = torch.zeros(n1, d + 1, n2, d + 1, dtype=x1.dtype)
    K[:, 0, :, 0] = k
    K[:, 0, :, 1:] = k[..., None] * dl / ell
    K[:, 1:, :, 0] = (-k[..., None] * dl / ell).permute(0, 2, 1)
    blk = (torch.eye(d, dtype=x1.dtype)[None, None] - dl[..., :, None] * dl[..., None, :]) * (k / ell ** 2)[..., None, None]
    K[:, 1:, :, 1:] = blk.permute(0, 2, 1, 3)
    return K.reshape(n1 * (d + 1), n2 * (d + 1))


def _canon2_case(dsvgp, dev, x1, x2, ell, s, base):
    ops = dsvgp._ops
    ctx = ops.Context.get(dev)
    d = x1.shape[1]
    hyp = _hyp(dev, ell, s)
    x1d, x2d = x1.float().to(dev).contiguous(), x2.float().to(dev).contiguous()
    center = ops.column_mean(ctx, x1d)
    E1, E2 = torch.eye(d).repeat(x1.shape[0], 1).to(dev), torch.eye(d).repeat(x2.shape[0], 1).to(dev)
    p1, p2 = ops.pack_points(ctx, x1d, E1, d, hyp, center), ops.pack_points(ctx, x2d, E2, d, hyp, center)
    di = (torch.arange(d, dtype=torch.int32) + base).to(dev)
    return ops, ctx, hyp, p1, p2, di


@pytest.mark.parametrize("n1,n2,base", [(4, 16, 0), (40, 52, 1), (37, 131, 0), (3, 5, 1), (300, 64, 0), (9, 17, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_kernel_fwd_canon2_matches_the_closed_form_and_the_general_kernel(dsvgp, gpu_device, n1, n2, base, dtype):
    d = p = 10
    g = torch.Generator().manual_seed(7 * n1 + n2)
    x1, x2 = torch.rand(n1, d, generator=g), torch.rand(n2, d, generator=g)
    ell, s = 0.83, 1.4
    ops, ctx, hyp, p1, p2, di = _canon2_case(dsvgp, gpu_device, x1, x2, ell, s, base)
    assert ops.canon2_supported(d, p) and not ops.canon2_supported(20, 5) and not ops.canon2_supported(13, 10)
    K2 = ops.kernel_fwd_canon2(ctx, p1, n1, p2, n2, d, p, di, base, hyp, dtype=dtype)
    Kg = ops.kernel_fwd(ctx, p1, n1, p2, n2, d, p, hyp, dtype=dtype)
    ref = _rbf_grad_closed_form(x1.double(), x2.double(), ell, s)
    assert K2.dtype == dtype
    assert relmax(K2, ref) < 2e-6 and relmax(K2, Kg) < 2e-6, (relmax(K2, ref), relmax(K2, Kg))
    # padded leading dimension / rows that are not 16-byte aligned: the bounds-checked store path
    buf = torch.zeros(n1 * (p + 1), n2 * (p + 1) + 3, device=gpu_device, dtype=dtype)
    Kv = ops.kernel_fwd_canon2(ctx, p1, n1, p2, n2, d, p, di, base, hyp, out=buf[:, 1:n2 * (p + 1) + 1])
    assert torch.equal(Kv, K2) and buf[:, 0].abs().max().item() == 0.0 and buf[:, -2:].abs().max().item() == 0.0


def test_kernel_fwd_canon2_symmetric_with_jitter(dsvgp, gpu_device):
    d = p = 10
    M = 45
    Z = torch.rand(M, d, generator=torch.Generator().manual_seed(5))
    ops, ctx, hyp, p1, _, di = _canon2_case(dsvgp, gpu_device, Z, Z, 0.7, 0.9, 0)
    K2 = ops.kernel_fwd_canon2(ctx, p1, M, p1, M, d, p, di, 0, hyp, jitter=1e-3, dtype=torch.float64)
    Kg = ops.kernel_fwd(ctx, p1, M, p1, M, d, p, hyp, jitter=1e-3, dtype=torch.float64)
    ref = _rbf_grad_closed_form(Z.double(), Z.double(), 0.7, 0.9) + 1e-3 * torch.eye(M * (p + 1), dtype=torch.float64)
    assert relmax(K2, ref) < 2e-6 and relmax(K2, Kg) < 2e-6
    Kc = K2.cpu()
    assert (Kc - Kc.t()).abs().max().item() < 1e-6 and torch.equal(Kc, Kc.float().double())
    torch.linalg.cholesky(Kc)


@pytest.mark.parametrize("n1,n2,sym", [(4, 16, False), (40, 52, False), (37, 131, False), (3, 5, False), (61, 61, True), (300, 300, True)])
@pytest.mark.parametrize("gdtype", [torch.float32, torch.float64])
def test_kernel_bwd_canon2_matches_autograd_and_the_general_kernel(dsvgp, gpu_device, n1, n2, sym, gdtype):
    d = p = 10
    q = p + 1
    g = torch.Generator().manual_seed(n1 + 13 * n2)
    x1 = torch.rand(n1, d, generator=g)
    x2 = x1 if sym else torch.rand(n2, d, generator=g)
    ell, s = 0.9, 1.2
    G = torch.randn(n1 * q, n2 * q, generator=g, dtype=torch.float64)
    if sym:
        G = 0.5 * (G + G.t())
    dev = gpu_device
    ops, ctx, hyp, p1, p2, di = _canon2_case(dsvgp, dev, x1, x2, ell, s, 1)
    if sym:
        p2 = p1
    x1r = x1.double().requires_grad_(True)
    ellr = torch.tensor(ell, dtype=torch.float64, requires_grad=True)
    sr = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    (_rbf_grad_closed_form_ad(x1r, x1r if sym else x2.double(), ellr, sr) * G).sum().backward()
    ldp = (n2 * q + 3) // 4 * 4                               # (rows of whole 16-byte pieces: what the DMA path asks for)
    buf = torch.zeros(n1 * q, ldp, dtype=gdtype, device=dev)
    buf[:, :n2 * q] = G.to(gdtype).to(dev)
    Gd = buf[:, :n2 * q]
    res = []
    for two in (True, False):
        dx, dv, dh = torch.zeros(n1, d, device=dev), torch.zeros(n1 * p, d, device=dev), torch.zeros(4, device=dev)
        if two:
            ops.kernel_bwd_canon2(ctx, Gd, p1, n1, p2, n2, d, p, di, 1, hyp, sym, dx, dv, dh)
            assert dv.abs().max().item() == 0.0
        else:
            ops.kernel_bwd(ctx, Gd, p1, n1, p2, n2, d, p, hyp, sym, dx, dv, dh)
        res.append((dx, dh))
    (dx, dh), (gx, gh) = res
    assert relmax(dx, x1r.grad) < 2e-4, relmax(dx, x1r.grad)
    assert abs(dh[0].item() - ellr.grad.item()) < 2e-4 * max(1.0, abs(ellr.grad.item())), (dh[0].item(), ellr.grad.item())
    assert abs(dh[1].item() - sr.grad.item()) < 2e-4 * max(1.0, abs(sr.grad.item()))
    assert relmax(dx, gx) < 1e-4 and abs(dh[0].item() - gh[0].item()) < 1e-4 * max(1.0, abs(gh[0].item()))
    # a second call accumulates (+=)
    dx2, dv2, dh2 = dx.clone(), torch.zeros(n1 * p, d, device=dev), dh.clone()
    ops.kernel_bwd_canon2(ctx, Gd, p1, n1, p2, n2, d, p, di, 1, hyp, sym, dx2, dv2, dh2)
    assert relmax(dx2, 2 * dx) < 1e-5


def _rbf_grad_closed_form_ad(x1, x2, ell, s):
    """the same closed form, differentiable in (x1, ell, s) (x2 = x1 passes the same tensor: the symmetric case)"""
    n1, d = x1.shape
    n2 = x2.shape[0]
    dl = (x1[:, None, :] - x2[None, :, :]) / ell
    k = s * torch.exp(-0.5 * (dl ** 2).sum(-1))
    top = torch.cat([k[..., None, None], (k[..., None] * dl / ell)[..., None, :]], -1)                       # [n1, n2, 1, d+1]
    blk = (torch.eye(d, dtype=x1.dtype) - dl[..., :, None] * dl[..., None, :]) * (k / ell ** 2)[..., None, None]
    bot = torch.cat([(-k[..., None] * dl / ell)[..., :, None], blk], -1)                                    # [n1, n2, d, d+1]
    K = torch.cat([top, bot], -2)                                                                             # [n1, n2, d+1, d+1]
    return K.permute(0, 2, 1, 3).reshape(n1 * (d + 1), n2 * (d + 1))


def test_kernel_bwd_canon2_rejects_rows_that_are_not_16_byte_pieces(dsvgp, gpu_device):
    d = p = 10
    x1, x2 = torch.rand(5, d), torch.rand(6, d)
    ops, ctx, hyp, p1, p2, di = _canon2_case(dsvgp, gpu_device, x1, x2, 0.8, 1.0, 0)
    buf = torch.randn(5 * 11, 6 * 11 + 1, device=gpu_device)
    dx, dv, dh = torch.zeros(5, d, device=gpu_device), torch.zeros(50, d, device=gpu_device), torch.zeros(4, device=gpu_device)
    with pytest.raises(dsvgp._lib.DsvgpError):
        ops.kernel_bwd_canon2(ctx, buf[:, :66], p1, 5, p2, 6, d, p, di, 0, hyp, False, dx, dv, dh)


def test_kernel_fwd_symmetric_double_with_jitter_and_exact_diagonal(dsvgp, gpu_device):
    g = torch.Generator().manual_seed(3)
    M, d, p = 50, 20, 5
    Z, V = torch.rand(M, d, generator=g), torch.randn(M * p, d, generator=g)
    ell, s = 0.6931, 0.6931
    K, _ = _kernel_gpu(dsvgp, gpu_device, Z, Z, V, V, ell, s, jitter=1e-3, dtype=torch.float64)
    assert K.dtype == torch.float64
    ref = s * O.kernel_matrix(Z.double(), Z.double(), V.double(), V.double(), ell) + 1e-3 * torch.eye(M * (p + 1), dtype=torch.float64)
    assert relmax(K, ref) < 2e-5
    Kc = K.cpu()
    assert (Kc - Kc.t()).abs().max().item() < 2e-6
    # r == 0 exactly on the diagonal blocks: value s + jitter, cross terms of the diagonal micro-blocks exactly 0
    q = p + 1
    for i in (0, 7, M - 1):
        blk = Kc[i * q:(i + 1) * q, i * q:(i + 1) * q]
        assert blk[0, 0].item() == pytest.approx(np.float32(s) + np.float32(1e-3), rel=1e-6)
        assert blk[0, 1:].abs().max().item() == 0.0 and blk[1:, 0].abs().max().item() == 0.0
    # fp64 output carries fp32 values (the reference's .double() cast)
    assert torch.equal(Kc, Kc.float().double())


def test_kernel_diag_and_errors(dsvgp, gpu_device):
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    hyp = _hyp(gpu_device, 0.5, 2.0)
    dg = ops.kernel_diag(ctx, 4, 2, hyp).cpu()
    assert torch.allclose(dg, torch.tensor([2.0, 8.0, 8.0] * 4))
    k = dsvgp._rbf_mod.RBFKernelDirectionalGrad().to(gpu_device)
    x = torch.rand(3, 2, device=gpu_device)
    v = torch.rand(6, 2, device=gpu_device)
    with pytest.raises(RuntimeError):
        k.forward(x, x + 1, diag=True, v1=v, v2=v)
    with pytest.raises(AssertionError):
        k.forward(x, x, v1=v, v2=v[:3])
    Kd = k.forward(x, x, diag=True, v1=v, v2=v)
    Kf = k.forward(x, x, v1=v, v2=v)
    assert relmax(torch.diagonal(Kf), Kd) < 1e-5 and k.num_outputs_per_input(x, x) == 3


# ------------------------------------------------------------------ kernel assembly backward
@pytest.mark.parametrize("n1,n2,d,p,sym", [(11, 23, 5, 2, False), (20, 45, 20, 5, False), (18, 18, 20, 5, True),
                                           (40, 300, 4, 1, False), (9, 9, 3, 3, True), (6, 10, 6, 0, False),
                                           (4, 7, 80, 40, False),        # (q DP = 41 x 84 > 3072: the one-wave tail of the backward)
                                           # kernel_bwd_split_kernel<., 11, 4> (q = 11) and kernel_bwd_pair_kernel with KSM = 16 over several tiles
                                           (50, 95, 10, 10, False), (45, 45, 10, 10, True), (30, 60, 50, 5, False), (20, 70, 40, 2, False)])
def test_kernel_bwd_matches_autograd(dsvgp, gpu_device, n1, n2, d, p, sym):
    ops = dsvgp._ops
    g = torch.Generator().manual_seed(n1 + 31 * n2 + d)
    x1 = torch.rand(n1, d, generator=g)
    v1 = torch.randn(n1 * p, d, generator=g)
    if sym:
        x2, v2 = x1, v1
    else:
        x2, v2 = torch.rand(n2, d, generator=g), torch.randn(n2 * p, d, generator=g)
    ell, s = 0.8, 1.3
    q = p + 1
    G = torch.randn(n1 * q, n2 * q, generator=g, dtype=torch.float64)
    if sym:
        G = G + G.t()
    # fp64 autograd truth
    x1r = x1.double().requires_grad_(True)
    v1r = v1.double().requires_grad_(True)
    ellr = torch.tensor(ell, dtype=torch.float64, requires_grad=True)
    sr = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    if sym:
        K = sr * O.kernel_matrix(x1r, x1r, v1r, v1r, ellr)
    else:
        K = sr * O.kernel_matrix(x1r, x2.double(), v1r, v2.double(), ellr)
    (K * G).sum().backward()
    # HIP
    dev = gpu_device
    _, (ctx, hyp, p1, p2) = _kernel_gpu(dsvgp, dev, x1, x2, v1, v2, ell, s)
    dx = torch.zeros(n1, d, device=dev)
    dv = torch.zeros(max(n1 * p, 1), d, device=dev)
    dh = torch.zeros(4, device=dev)
    for Gd in (G.to(dev), G.float().to(dev)):                 # double and float upstream gradients
        dx.zero_(); dv.zero_(); dh.zero_()
        ops.kernel_bwd(ctx, Gd.contiguous(), p1, n1, p2, n2, d, p, hyp, sym, dx, dv, dh)
        assert relmax(dx, x1r.grad) < 2e-4, "d_x1"
        if p > 0:
            assert relmax(dv[:n1 * p], v1r.grad) < 2e-4, "d_v1"
        assert abs(dh[0].item() - ellr.grad.item()) < 2e-4 * max(1.0, abs(ellr.grad.item())), "d_lengthscale"
        assert abs(dh[1].item() - sr.grad.item()) < 2e-4 * max(1.0, abs(sr.grad.item())), "d_outputscale"


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-4), (torch.float64, 1e-9)])
@pytest.mark.parametrize("n1,n2,d,p,same", [(13, 21, 5, 2, False), (16, 16, 20, 5, True), (7, 40, 4, 1, False), (6, 9, 6, 0, False)])
def test_kernel_plugin_is_differentiable_like_the_reference(dsvgp, gpu_device, dt, tol, n1, n2, d, p, same):
    """``RBFKernelDirectionalGrad.forward`` (reference RBFKernelDirectionalGrad.py:41-108) is an autograd participant: gradients
    w.r.t. x1, x2, v1, v2 and raw_lengthscale through the mirrored plugin (dsvgp_kernel_fwd / dsvgp_kernel_bwd behind a
    torch.autograd.Function) against fp64 autograd through the oracle's kernel.  x1 is x2 (``same``): the K_ZZ call pattern,
    where autograd sums both roles."""
    g = torch.Generator().manual_seed(5 * n1 + n2 + d)
    x1 = torch.rand(n1, d, generator=g, dtype=torch.float64)
    v1 = torch.randn(n1 * p, d, generator=g, dtype=torch.float64)
    x2 = x1 if same else torch.rand(n2, d, generator=g, dtype=torch.float64)
    v2 = v1 if same else torch.randn(n2 * p, d, generator=g, dtype=torch.float64)
    q = p + 1
    G = torch.randn(n1 * q, (n1 if same else n2) * q, generator=g, dtype=torch.float64)
    raw = 0.4
    # fp64 truth: autograd through the oracle kernel
    leaves = [t.clone().requires_grad_(True) for t in ((x1, v1) if same else (x1, x2, v1, v2))]
    rawr = torch.tensor([[raw]], dtype=torch.float64, requires_grad=True)
    ell = torch.nn.functional.softplus(rawr).reshape(())
    if same:
        Kref = O.kernel_matrix(leaves[0], leaves[0], leaves[1], leaves[1], ell)
    else:
        Kref = O.kernel_matrix(leaves[0], leaves[1], leaves[2], leaves[3], ell)
    (Kref * G).sum().backward()
    # the plugin
    k = dsvgp._rbf_mod.RBFKernelDirectionalGrad().to(gpu_device)
    if dt == torch.float64:
        k = k.double()
    with torch.no_grad():
        k.raw_lengthscale.fill_(raw)
    gl = [t.to(dt).to(gpu_device).requires_grad_(True) for t in ((x1, v1) if same else (x1, x2, v1, v2))]
    if same:
        K = k.forward(gl[0], gl[0], v1=gl[1], v2=gl[1])
    else:
        K = k.forward(gl[0], gl[1], v1=gl[2], v2=gl[3])
    assert K.requires_grad and relmax(K, Kref) < (2e-6 if dt == torch.float32 else 1e-12)
    (K * G.to(dt).to(gpu_device)).sum().backward()
    errs = {}
    for i, (a, b) in enumerate(zip(gl, leaves)):
        if a.numel():
            errs["arg%d" % i] = relmax(a.grad, b.grad)
    errs["raw_lengthscale"] = relmax(k.raw_lengthscale.grad, rawr.grad)
    print("[parity] differentiable kernel plugin %s %s: %s" % (str(dt).split(".")[-1], (n1, n2, d, p, same),
                                                               ", ".join("%s %.1e" % kv for kv in errs.items())))
    assert max(errs.values()) < tol, errs


# ------------------------------------------------------------------ GEMM / trsm / potrf
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_plain(dsvgp, gpu_device, dt, ta, tb):
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(7)
    M, N, K = 150, 263, 77
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=torch.float64)
    B = torch.randn((N, K) if tb else (K, N), generator=g, dtype=torch.float64)
    Cin = torch.randn(M, N, generator=g, dtype=torch.float64)
    ref = 0.7 * (A.t() if ta else A) @ (B.t() if tb else B) - 0.3 * Cin
    C = torch.empty(M, N, dtype=dt, device=gpu_device)
    flags = (L.TRANS_A if ta else 0) | (L.TRANS_B if tb else 0)
    ops.gemm(ctx, flags, A.to(dt).to(gpu_device), B.to(dt).to(gpu_device), C, alpha=0.7, beta=-0.3,
             Cin=Cin.to(dt).to(gpu_device))
    assert relmax(C, ref) < (1e-12 if dt == torch.float64 else 2e-5)


def test_gemm_mfma_layout_asymmetric_identity(dsvgp, gpu_device):
    """A = I with an asymmetric integer B catches a transposed C write (exact arithmetic)."""
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    n = 200
    B = (torch.arange(n)[:, None] * 3 + torch.arange(n)[None, :] * 7 % 11).double()
    for dt in (torch.float32, torch.float64):
        C = torch.empty(n, n, dtype=dt, device=gpu_device)
        ops.gemm(ctx, 0, torch.eye(n, dtype=dt, device=gpu_device), B.to(dt).to(gpu_device), C)
        assert torch.equal(C.cpu().double(), B)


def test_gemm_triangular_flags_mask_garbage(dsvgp, gpu_device):
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(11)
    n, N = 300, 170
    Afull = torch.randn(n, n, generator=g, dtype=torch.float64)
    B = torch.randn(n, N, generator=g, dtype=torch.float64)
    dev = gpu_device
    for dt, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        Ad, Bd = Afull.to(dt).to(dev), B.to(dt).to(dev)
        C = torch.empty(n, N, dtype=dt, device=dev)
        ops.gemm(ctx, L.A_LOWER, Ad, Bd, C)                              # tril(A) B
        assert relmax(C, Afull.tril() @ B) < tol
        ops.gemm(ctx, L.TRANS_A | L.A_UPPER, Ad, Bd, C)                  # tril(A)^T B
        assert relmax(C, Afull.tril().t() @ B) < tol
        ops.gemm(ctx, L.A_UPPER, Ad, Bd, C)                              # triu(A) B
        assert relmax(C, Afull.triu() @ B) < tol
        S = torch.empty(n, n, dtype=dt, device=dev)
        ops.gemm(ctx, L.TRANS_A | L.A_UPPER | L.B_LOWER, Ad, Ad, S)      # tril(A)^T tril(A)
        assert relmax(S, Afull.tril().t() @ Afull.tril()) < tol
        C2 = torch.empty(N, n, dtype=dt, device=dev)
        ops.gemm(ctx, L.B_LOWER, Bd.t().contiguous(), Ad, C2)           # B^T tril(A)
        assert relmax(C2, B.t() @ Afull.tril()) < tol
        O_ = torch.full((n, n), 7.0, dtype=dt, device=dev)
        ops.gemm(ctx, L.TRANS_B | L.OUT_LOWER, Bd, Bd, O_, alpha=2.0)     # tril(2 B B^T), zeros above
        assert relmax(O_, (2 * B @ B.t()).tril()) < tol


def test_gemm_splitk_kscale_long_k(dsvgp, gpu_device):
    """K = minibatch axis: split-K + atomics path, column scaling (A diag(v) W^T), tril output."""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(13)
    Mp, Bp = 190, 6000
    A = torch.randn(Mp, Bp, generator=g)
    W = torch.randn(Mp, Bp, generator=g)
    v = torch.rand(Bp, generator=g)
    dev = gpu_device
    out = torch.full((Mp, Mp), 3.0, device=dev)
    ops.gemm(ctx, L.TRANS_B | L.OUT_LOWER, A.to(dev), W.to(dev), out, alpha=2.0, kscale=v.to(dev))
    ref = (2 * (A.double() * v.double()) @ W.double().t()).tril()
    assert relmax(out, ref) < 3e-5
    out64 = torch.empty(Mp, Mp, dtype=torch.float64, device=dev)
    ops.gemm(ctx, L.TRANS_B | L.OUT_LOWER, A.double().to(dev), W.double().to(dev), out64, alpha=-1.0)
    assert relmax(out64, -(A.double() @ W.double().t()).tril()) < 1e-12


def _spd(n, g, cond_jitter=1e-3):
    Z = torch.rand(n // 3 + 1, 4, generator=g, dtype=torch.float64)
    V = torch.randn((n // 3 + 1) * 2, 4, generator=g, dtype=torch.float64)
    K = 0.7 * O.kernel_matrix(Z, Z, V, V, 0.7)[:n, :n]
    return K + cond_jitter * torch.eye(n, dtype=torch.float64)


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("n", [64, 100, 333, 700, 1500, 2048, 2100])       # (>= 1700: launches of the pipelined-strip kernel)
def test_potrf_row_major_lower(dsvgp, gpu_device, n, algo):
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    K = _spd(n, torch.Generator().manual_seed(n))
    A = K.tril().to(gpu_device).contiguous()                    # only the lower triangle may be read
    info = torch.full((1,), -7, dtype=torch.int32, device=gpu_device)
    ops.potrf_(ctx, A, info, algo)
    assert int(info.item()) == 0
    Lref = torch.linalg.cholesky(K)
    assert relmax(A.tril(), Lref) < 1e-10
    bad = torch.eye(n, dtype=torch.float64, device=gpu_device)
    bad[n // 2, n // 2] = -1.0
    ops.potrf_(ctx, bad, info, algo)
    assert int(info.item()) == n // 2 + 1                       # leading minor index, as LAPACK


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_lib_f32_row_major_convention(dsvgp, gpu_device, ta, tb):
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(5 + ta + 2 * tb)
    M, N, K = 70, 131, 53
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    C = torch.randn(M, N, generator=g)
    ref = 0.7 * (A.t() if ta else A).double() @ (B.t() if tb else B).double() + 0.3 * C.double()
    Cg = C.to(gpu_device)
    ops.gemm_lib_f32(ctx, (L.TRANS_A if ta else 0) | (L.TRANS_B if tb else 0), A.to(gpu_device), B.to(gpu_device), Cg,
                     alpha=0.7, beta=0.3)
    assert relmax(Cg, ref) < 1e-5


@pytest.mark.parametrize("n", [40, 64, 100, 333, 704, 1500, 2048, 2100, 2001])
def test_potrf_inverse_fused(dsvgp, gpu_device, n):
    """Blocked Cholesky with the fused forward elimination: L and L^-1 from the same launches (ragged last block, one
    block, several blocks); later solves reuse the inverse.  n >= 1300: launches with more than 300 tiles run the
    software-pipelined strip kernel (one workgroup per CU, LDS-DMA operands) -- whole blocks (2048), a ragged last block
    (2100), and an odd leading dimension (2001), which keeps the two-per-CU kernel (16-byte DMA granules need an even one)."""
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(n)
    K = _spd(n, g)
    A = K.clone().to(gpu_device)
    info = torch.zeros(1, dtype=torch.int32, device=gpu_device)
    nb = 4096
    ws = ops.trsm_workspace(n, max(n, 600), nb, gpu_device)
    ops.potrf_inverse_(ctx, A, info, nb, ws)
    assert int(info.item()) == 0
    L = torch.tril(A).cpu()
    assert (L @ L.t() - K).abs().max() < 1e-10 * K.abs().max()
    Linv = torch.tril(ws[:n * n * 8].view(torch.float64).view(n, n)).cpu()
    assert (Linv @ L - torch.eye(n, dtype=torch.float64)).abs().max() < 1e-9
    # the transposed image comes out of the same launches (second slot of the workspace), bit-identical
    LinvT = torch.triu(ws[n * n * 8:2 * n * n * 8].view(torch.float64).view(n, n)).cpu()
    assert torch.equal(LinvT, Linv.t())
    for nrhs in (17, 600):                   # >= 512 right-hand sides: the forward solve streams the transposed image
        B = torch.randn(n, nrhs, generator=g, dtype=torch.float64)
        X = torch.empty(n, nrhs, dtype=torch.float64, device=gpu_device)
        for trans in (0, 1):
            ops.trsm(ctx, A, B.to(gpu_device), trans, X, None, nb, ws, reuse_inverse=True)
            ref = torch.linalg.solve_triangular(L.t() if trans else L, B, upper=bool(trans))
            assert relmax(X, ref) < 1e-9
    bad = K.clone()
    bad[n // 2, n // 2] = -1.0
    bad = bad.to(gpu_device)
    ops.potrf_inverse_(ctx, bad, info, nb, ws)
    assert int(info.item()) == n // 2 + 1


def test_potrf_inverse_base_not_16_byte_aligned(dsvgp, gpu_device):
    """dsvgp_potrf_inverse takes any double*: a matrix that starts 8 bytes into an allocation (a torch view at an odd element
    offset) with an even leading dimension must not reach the pipelined-strip kernel, whose B tiles arrive by 16-byte LDS-DMA
    pieces -- the launcher's gate sends it to the two-per-CU kernel; n = 1800: launches with more than 300 tiles."""
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    n, ld = 1800, 1802
    K = _spd(n, torch.Generator().manual_seed(77))
    buf = torch.zeros(n * ld + 1, dtype=torch.float64, device=gpu_device)
    A = buf[1:].view(n, ld)[:, :n]
    assert A.data_ptr() % 16 == 8
    A.copy_(K.to(gpu_device))
    info = torch.zeros(1, dtype=torch.int32, device=gpu_device)
    ws = ops.trsm_workspace(n, n, 4096, gpu_device)
    ops.potrf_inverse_(ctx, A, info, 4096, ws)
    assert int(info.item()) == 0
    L = torch.tril(A).cpu()
    assert (L @ L.t() - K).abs().max() < 1e-10 * K.abs().max()
    Linv = torch.tril(ws[:n * n * 8].view(torch.float64).view(n, n)).cpu()
    assert (Linv @ L - torch.eye(n, dtype=torch.float64)).abs().max() < 1e-9


@pytest.mark.parametrize("n,nrhs,nb", [(100, 37, 64), (333, 500, 128), (700, 260, 256), (700, 260, 1024), (520, 129, 512)])
@pytest.mark.parametrize("trans", [0, 1])
def test_trsm_panel_vs_solve_triangular(dsvgp, gpu_device, n, nrhs, nb, trans):
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(n + nrhs)
    Lc = torch.linalg.cholesky(_spd(n, g))
    Lg = (Lc + torch.randn(n, n, generator=g, dtype=torch.float64).triu(1)).contiguous().to(gpu_device)   # garbage above the diagonal
    B = torch.randn(n, nrhs, generator=g, dtype=torch.float64)
    ref = torch.linalg.solve_triangular(Lc.t() if trans else Lc, B, upper=bool(trans))
    ws = ops.trsm_workspace(n, max(nrhs, n), nb, gpu_device)
    X64 = torch.empty(n, nrhs, dtype=torch.float64, device=gpu_device)
    X32 = torch.empty(n, nrhs, dtype=torch.float32, device=gpu_device)
    ops.trsm(ctx, Lg, B.float().to(gpu_device), trans, X64, X32, nb, ws)                 # float RHS
    ref32 = torch.linalg.solve_triangular(Lc.t() if trans else Lc, B.float().double(), upper=bool(trans))
    assert relmax(X64, ref32) < 1e-9
    assert relmax(X32, ref32) < 1e-6
    Bd = B.to(gpu_device)
    ops.trsm(ctx, Lg, Bd, trans, Bd, None, nb, ws, reuse_inverse=True)                   # double RHS, in place
    assert relmax(Bd, ref) < 1e-9


# ------------------------------------------------------------------ ELBO term kernels
def test_predictive_stats_likelihood_abar_rowdot(dsvgp, gpu_device):
    ops = dsvgp._ops
    dev = gpu_device
    ctx = ops.Context.get(dev)
    g = torch.Generator().manual_seed(17)
    Mp, B, p = 211, 173, 2
    Bp = B * (p + 1)
    A, W = torch.randn(Mp, Bp, generator=g) * 0.1, torch.randn(Mp, Bp, generator=g) * 0.1
    m, c = torch.randn(Mp, generator=g), torch.tensor([0.3])
    y = torch.randn(Bp, generator=g)
    hyp = _hyp(dev, 0.7, 1.2, 0.05)
    mu, var = torch.empty(Bp, device=dev), torch.empty(Bp, device=dev)
    ops.predictive_stats(ctx, A.to(dev), W.to(dev), p, m.to(dev), c.to(dev), hyp, mu, var)
    A6, W6 = A.double(), W.double()
    dg = 1.2 * O.kernel_diag(B, p, torch.tensor(0.7, dtype=torch.float64))
    mu_ref = A6.t() @ m.double() + 0.3
    var_ref = dg + 1e-4 + (W6 * W6 - A6 * A6).sum(0)
    assert relmax(mu, mu_ref) < 1e-5 and relmax(var, var_ref) < 1e-5
    for mll, name in ((0, "ELBO"), (1, "PLL")):
        mur = mu_ref.clone().requires_grad_(True)
        varr = var_ref.clone().requires_grad_(True)
        noise = torch.tensor(0.05, dtype=torch.float64, requires_grad=True)
        varn = varr + noise
        if mll == 0:
            ll = -0.5 * (((y.double() - mur) ** 2 + varn) / noise + torch.log(noise) + np.log(2 * np.pi))
        else:
            tot = varn + noise
            ll = -0.5 * ((y.double() - mur) ** 2 / tot + torch.log(tot) + np.log(2 * np.pi))
        rows = 2.0 * Bp
        (-(ll.sum()) / rows).backward()
        mb, vb, vn = (torch.empty(Bp, device=dev) for _ in range(3))
        sc = torch.empty(8, device=dev)
        ops.likelihood_terms(ctx, mu_ref.float().to(dev), var_ref.float().to(dev), y.to(dev), p, hyp, mll, rows, mb, vb, vn, sc)
        assert relmax(mb, mur.grad) < 1e-5 and relmax(vb, varr.grad) < 1e-5, name
        assert relmax(vn, varn.detach()) < 1e-6
        assert abs(sc[0].item() - ll.sum().item()) < 1e-4 * abs(ll.sum().item())
        assert abs(sc[1].item() - noise.grad.item()) < 1e-4 * abs(noise.grad.item())
        assert abs(sc[2].item() - mur.grad.sum().item()) < 1e-4 * max(abs(mur.grad.sum().item()), 1e-3)
    U = torch.randn(Mp, Bp, generator=g)
    out = torch.empty(Mp, Bp, device=dev)
    ops.abar(ctx, A.to(dev), U.to(dev), m.to(dev), mb, vb, out)
    ref = m.double()[:, None] * mb.cpu().double()[None] + 2 * vb.cpu().double()[None] * (U.double() - A6)
    assert relmax(out, ref) < 1e-5
    acc = torch.ones(Mp, device=dev)
    ops.rowdot_accum(ctx, A.to(dev), mb, acc)
    assert relmax(acc, 1 + A6 @ mb.cpu().double()) < 1e-5


def test_kl_terms_phi_transpose_adddiag(dsvgp, gpu_device):
    ops = dsvgp._ops
    dev = gpu_device
    ctx = ops.Context.get(dev)
    g = torch.Generator().manual_seed(19)
    Mp, nd = 301, 1234.0
    m = torch.randn(Mp, generator=g)
    LS = torch.eye(Mp) + 0.05 * torch.randn(Mp, Mp, generator=g)          # upper part is garbage that must be ignored
    mr = m.double().requires_grad_(True)
    Lr = LS.double().requires_grad_(True)
    kl = O.kl_whitened(mr, torch.tril(Lr))
    (kl / nd).backward()
    buf = torch.zeros(Mp + 1, device=dev)
    dm = torch.ones(Mp, device=dev)
    dL = torch.full((Mp, Mp), 5.0, device=dev).tril()                      # pre-existing lower gradient is accumulated
    ops.kl_terms(ctx, m.to(dev), LS.to(dev), nd, buf, dm, dL)
    assert abs(buf[0].item() - kl.item()) < 1e-4 * abs(kl.item())
    assert relmax(dm, 1 + mr.grad) < 1e-5
    assert relmax(dL, 5.0 * torch.ones(Mp, Mp, dtype=torch.float64).tril() + Lr.grad) < 1e-5
    G = torch.randn(Mp, Mp, generator=g, dtype=torch.float64)
    Gd = G.to(dev)
    ops.phi_symmetrize_(ctx, Gd)
    Phi = G.tril()
    Phi.diagonal().mul_(0.5)
    assert relmax(Gd, Phi + Phi.t()) < 1e-15
    R = torch.randn(130, 77, generator=g, dtype=torch.float64).to(dev)
    Rt = torch.empty(77, 130, dtype=torch.float64, device=dev)
    ops.transpose_f64(ctx, R, Rt)
    assert torch.equal(Rt, R.t())
    ops.add_diag_(ctx, Gd, 0.25)
    assert relmax(Gd, Phi + Phi.t() + 0.25 * torch.eye(Mp, dtype=torch.float64)) < 1e-15


@pytest.mark.parametrize("n", [2, 37, 600, 1031])
def test_fused_adam_walks_only_the_lower_triangle_of_a_flagged_square_parameter(dsvgp, gpu_device, n):
    """a parameter flagged ``_dsvgp_tril`` (chol_variational_covar: gradient exactly zero above the diagonal) is updated below and on
    the diagonal exactly as by the dense launch / torch.optim.Adam; what lies above the diagonal -- parameter, moments -- is not touched"""
    dev = gpu_device
    g = torch.Generator().manual_seed(n)
    w0, v0 = torch.randn(n, n, generator=g), torch.randn(n, generator=g)         # (garbage above the diagonal stays where it is)
    pt = [w0.clone().requires_grad_(True), v0.clone().requires_grad_(True)]
    pa = [torch.nn.Parameter(w0.clone().to(dev)), torch.nn.Parameter(v0.clone().to(dev))]
    pb = [torch.nn.Parameter(w0.clone().to(dev)), torch.nn.Parameter(v0.clone().to(dev))]
    pa[0]._dsvgp_tril = True
    ot, oa, ob = torch.optim.Adam(pt, lr=0.03), dsvgp.FusedAdam(pa, lr=0.03), dsvgp.FusedAdam(pb, lr=0.03)
    for k in range(4):
        gw, gv = torch.randn(n, n, generator=g).tril(), torch.randn(n, generator=g)
        for ps in (pt, pa, pb):
            ps[0].grad = gw.clone().to(ps[0].device)
            ps[1].grad = gv.clone().to(ps[1].device)
        ot.step(); oa.step(); ob.step()
    assert torch.equal(pa[0].detach(), pb[0].detach()) and torch.equal(pa[1].detach(), pb[1].detach())       # bit-identical to the dense walk
    assert relmax(pa[0].detach(), pt[0].detach()) < 2e-6
    assert torch.equal(pa[0].detach().cpu().triu(1), w0.triu(1))
    st = oa.state[pa[0]]
    assert st["exp_avg"].triu(1).abs().max().item() == 0.0 and st["exp_avg_sq"].triu(1).abs().max().item() == 0.0
    # the flag alone on the single-tensor path
    pc = torch.nn.Parameter(w0.clone().to(dev))
    pc._dsvgp_tril = True
    oc, od = dsvgp.FusedAdam([pc], lr=0.03), dsvgp.FusedAdam([torch.nn.Parameter(w0.clone().to(dev))], lr=0.03)
    gw = torch.randn(n, n, generator=g).tril().to(dev)
    pc.grad = gw.clone(); od.param_groups[0]["params"][0].grad = gw.clone()
    oc.step(); od.step()
    assert torch.equal(pc.detach(), od.param_groups[0]["params"][0].detach())


def test_gather_batch_and_fused_adam(dsvgp, gpu_device):
    ops = dsvgp._ops
    dev = gpu_device
    ctx = ops.Context.get(dev)
    g = torch.Generator().manual_seed(23)
    N, d, p = 500, 7, 3
    X, Y = torch.rand(N, d, generator=g), torch.rand(N, d + 1, generator=g)
    idx = torch.randperm(N, generator=g)[:64]
    cols = [0, 2, 5, 6]
    xb = torch.empty(64, d, device=dev)
    yb = torch.empty(64 * (p + 1), device=dev)
    ops.gather_batch(ctx, X.to(dev), Y.to(dev), idx.to(dev), torch.tensor(cols, dtype=torch.int32, device=dev), p, xb, yb)
    assert torch.equal(xb.cpu(), X[idx]) and torch.equal(yb.cpu(), Y[idx][:, cols].reshape(-1))
    # ... and the batch's derivative directions in the same launch: rows cols[1:] - 1 of the direction table, repeated per point
    E = torch.rand(d, d, generator=g)
    Db = torch.empty(64 * p, d, device=dev)
    xb.zero_(); yb.zero_()
    ops.gather_batch(ctx, X.to(dev), Y.to(dev), idx.to(dev), torch.tensor(cols, dtype=torch.int32, device=dev), p, xb, yb,
                     E.to(dev), Db)
    assert torch.equal(xb.cpu(), X[idx]) and torch.equal(yb.cpu(), Y[idx][:, cols].reshape(-1))
    assert torch.equal(Db.cpu(), E[torch.tensor(cols[1:]) - 1].repeat(64, 1))
    # [S | .] -> [S - I | m noise rows] (the right-hand side of the [Q' | a] solve) in one launch
    n = 37
    Se = torch.rand(n, n + 3, generator=g)
    mvec, hyp = torch.rand(n, generator=g), torch.tensor([0.7, 1.3, 0.02, 0.0])
    Sd = Se.to(dev)
    ops.sminus_i_col_(ctx, Sd[:, :n + 1], n, mvec.to(dev), hyp.to(dev), 1536.0)
    ref = Se.clone()
    ref[:, :n] -= torch.eye(n)
    ref[:, n] = mvec * (hyp[2] * 1536.0)
    assert torch.equal(Sd.cpu(), ref)
    # Adam: 5 steps against torch.optim.Adam
    w0 = torch.randn(1000, generator=g)
    wt = w0.clone().requires_grad_(True)
    wh = torch.nn.Parameter(w0.clone().to(dev))
    ot = torch.optim.Adam([wt], lr=0.01)
    oh = dsvgp.FusedAdam([wh], lr=0.01)
    for k in range(5):
        gr = torch.randn(1000, generator=g) * (k + 1)
        wt.grad = gr.clone()
        wh.grad = gr.clone().to(dev)
        ot.step()
        oh.step()
    assert relmax(wh.detach(), wt.detach()) < 2e-6
    # several tensors of one optimizer in ONE launch (dsvgp_adam_step_multi): ragged sizes, a [1,1] and a 0-d parameter,
    # one parameter without gradient, more tensors than one launch takes -- bit-identical to the per-tensor launches
    shapes = [(300, 7), (5,), (1, 1), (), (70000,), (3, 3)] + [(11,)] * 14
    init = [torch.randn(*sh, generator=g) if len(sh) else torch.randn((), generator=g) for sh in shapes]
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pt = [t.clone().requires_grad_(True) for t in init]
    oa, ot = dsvgp.FusedAdam(pa, lr=0.02), torch.optim.Adam(pt, lr=0.02)
    ob = [dsvgp.FusedAdam([q], lr=0.02) for q in pb]
    for k in range(4):
        for i, (a, b, t) in enumerate(zip(pa, pb, pt)):
            if i == 1 and k < 2:
                a.grad = b.grad = t.grad = None          # joins later: its step count differs from the others'
                continue
            gr = torch.randn(t.shape, generator=g)
            a.grad, b.grad, t.grad = gr.to(dev), gr.to(dev), gr.clone()
        oa.step()
        ot.step()
        for o in ob:
            o.step()
    for a, b, t in zip(pa, pb, pt):
        assert torch.equal(a.detach(), b.detach())
        assert relmax(a.detach().reshape(-1), t.detach().reshape(-1)) < 5e-6


def test_no_cpu_fallback(dsvgp):
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    with pytest.raises(Exception):
        dsvgp._ops.Context.get(torch.device("cpu"))
    eng = dsvgp.ElboEngine(torch.device("cpu"))
    P = O.init_params(torch.rand(4, 2), torch.eye(2)[:1].repeat(4, 1))
    with pytest.raises(Exception):
        eng.predict(P, torch.rand(3, 2), torch.eye(2)[:1].repeat(3, 1))


@pytest.mark.parametrize("n,nextra,ld_pad", [(1, 0, 0), (7, 7, 0), (333, 334, 5), (3000, 3000, 0)])
def test_tril_pack_unpack_roundtrip(dsvgp, gpu_device, n, nextra, ld_pad):
    """The data-parallel wire format [packed tril | extra]: n(n+1)/2 + k floats, exact round trip, upper part untouched."""
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(n, n + ld_pad, generator=g).to(gpu_device)[:, :n]
    extra = torch.randn(nextra, generator=g).to(gpu_device)
    pk = torch.full((ops.tril_packed_numel(n, nextra) + 3,), -7.0, device=gpu_device)
    ops.tril_pack_f32(ctx, A, extra, pk)
    i, j = torch.tril_indices(n, n)
    assert torch.equal(pk[:i.numel()].cpu(), A.cpu()[i, j])
    assert torch.equal(pk[i.numel():i.numel() + nextra], extra) and (pk[-3:] == -7.0).all()
    B = torch.full((n, n + ld_pad), 5.0, device=gpu_device)[:, :n]
    e2 = torch.zeros(nextra, device=gpu_device)
    ops.tril_unpack_f32(ctx, pk, B, e2)
    assert torch.equal(B.tril(), A.tril()) and (B.triu(1) == torch.full_like(B, 5.0).triu(1)).all()
    assert torch.equal(e2, extra)


@pytest.mark.parametrize("M,N,K,ta,tb,lower,pad", [
    (700, 1100, 1300, 0, 0, 0, 0),        # ragged in M, N and K; A k-contiguous, B n-contiguous (the dense K_ZX-bar layout)
    (700, 1100, 1300, 1, 1, 0, 0),        # A m-contiguous, B k-contiguous
    (640, 1000, 1301, 0, 0, 0, 1),        # K % 4 != 0 with caller-zeroed padding (DSVGP_GEMM_K_PADDED): LDS-DMA kernel
    (640, 1000, 1301, 0, 0, 0, 0),        # ... and without the promise: register-staged kernel
    (1001, 1000, 6000, 0, 1, 1, 0),       # [G ; b^T] = tril([A ; mu^T] A^T): both k-contiguous, split-K atomics, square tile grid
    (1025, 1024, 2048, 0, 1, 1, 0),       # the extra row opens a tile row of its own (tiles_m = tiles_n + 1)
    (900, 1000, 2048, 1, 0, 1, 0),        # lower output, fewer tile rows than columns
])
def test_gemm32_mfma_32x32x2_kernels(dsvgp, gpu_device, M, N, K, ta, tb, lower, pad):
    """csrc/gemm32.hip (fp32 products >= 512^3 without triangular operands) against fp64 torch; 3e-5 of the max magnitude
    (fp32 accumulation over K <= 6000, split-K partial sums met in atomics)."""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(M + N + K)
    K4 = (K + 3) // 4 * 4
    ka = K4 if (not ta) else K            # leading dimension of a k-contiguous operand: rows padded to a multiple of 4
    A = torch.randn((K, M) if ta else (M, ka), generator=g)
    B = torch.randn((N, K4) if tb else (K, N), generator=g)
    if pad:
        if not ta:
            A[:, K:] = 0
        if tb:
            B[:, K:] = 0
    Ad, Bd = A.to(gpu_device), B.to(gpu_device)
    Av = Ad if ta else Ad[:, :K]
    Bv = Bd[:, :K] if tb else Bd
    C = torch.full((M, N), 5.0, device=gpu_device)
    flags = (L.TRANS_A if ta else 0) | (L.TRANS_B if tb else 0) | (L.OUT_LOWER if lower else 0) | (L.K_PADDED if pad else 0)
    ops.gemm(ctx, flags, Av, Bv, C, alpha=0.75)
    opA = (A.t() if ta else A[:, :K]).double()
    opB = (B[:, :K].t() if tb else B).double()
    ref = 0.75 * opA @ opB
    if lower:
        ref = ref.tril()
    err = relmax(C, ref)
    print("gemm32 M=%d N=%d K=%d ta=%d tb=%d lower=%d pad=%d: rel. error %.2e" % (M, N, K, ta, tb, lower, pad, err))
    assert err < 3e-5
    if lower:
        assert C.triu(1).abs().max().item() == 0.0


@pytest.mark.gpu
def test_mfma_rate_probe_reports_a_plausible_roof(dsvgp, gpu_device):
    """bench.py's `roofline.sustained`: back-to-back MFMAs from registers must land between half the data-sheet peak and the peak"""
    ctx = dsvgp._ops.Context.get(gpu_device)
    r64 = dsvgp._ops.mfma_rate(ctx, True, 10)
    r32 = dsvgp._ops.mfma_rate(ctx, False, 10)
    assert 39.0 < r64 <= 78.6 * 1.02, r64
    assert 78.0 < r32 <= 157.3 * 1.02, r32


@pytest.mark.parametrize("M,N,pad", [(300, 1537, 0), (37, 64, 1), (3000, 2048, 0), (1, 5, 0)])
def test_gemv_f64_both_orientations(dsvgp, gpu_device, M, N, pad):
    """dsvgp_gemv_f64 (the float64 model mode's A^T m and A mu-bar): against torch fp64, 1e-13; odd row strides and sizes take
    the scalar path"""
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, N + pad, generator=g, dtype=torch.float64).to(gpu_device)[:, :N]
    for trans in (False, True):
        x = torch.randn(M if trans else N, generator=g, dtype=torch.float64).to(gpu_device)
        y = torch.full((N if trans else M,), float("nan"), dtype=torch.float64, device=gpu_device)
        ops.gemv_f64(ctx, A, x, y, trans=trans)
        ref = (A.t() if trans else A) @ x
        assert relmax(y, ref) < 1e-13, (M, N, pad, trans, relmax(y, ref))


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("bfloat", [False, True])
def test_gemm_f64_k_contiguous_operands_on_the_lean_kernel(dsvgp, gpu_device, ta, tb, bfloat):
    """fp64 products with >= 1024 tiles of 64 x 64 whose operands are K-contiguous (A stored [M, K] and / or B stored [N, K]) run on
    gemm64.hip's A_KC / B_KC staging (L-bar of the fp32 step; Gram and dense products of the float64 model mode): ragged M, N and K,
    a float right operand, OUT_LOWER -- against torch fp64"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(7 + ta + 2 * tb)
    M, N, K = 2101, 2050, 301
    A = torch.randn((K, M) if ta else (M, K + 1), generator=g, dtype=torch.float64)          # (even leading dimension for the row-major case)
    B = torch.randn((N, K + 3) if tb else (K, N), generator=g, dtype=torch.float64)
    if not ta:
        A = A[:, :K]
    if tb:
        B = B[:, :K]
    Bq = B.float().double() if bfloat else B
    ref = (A.t() if ta else A) @ (Bq.t() if tb else Bq)
    Ad = A.to(gpu_device) if ta else torch.randn(M, K + 1, dtype=torch.float64).to(gpu_device)[:, :K].copy_(A.to(gpu_device))
    if tb:
        Bd = torch.empty(N, K + 3, dtype=torch.float32 if bfloat else torch.float64, device=gpu_device)[:, :K]
        Bd.copy_(B.to(gpu_device))
    else:
        Bd = (B.float() if bfloat else B).to(gpu_device)
    for lower in (0, L.OUT_LOWER):
        C = torch.full((M, N), float("nan"), dtype=torch.float64, device=gpu_device)
        ops.gemm(ctx, (L.TRANS_A if ta else 0) | (L.TRANS_B if tb else 0) | lower, Ad, Bd, C, alpha=-0.5)
        want = -0.5 * (torch.tril(ref) if lower else ref)
        assert relmax(C, want) < 1e-13, (ta, tb, bfloat, lower, relmax(C, want))


@pytest.mark.parametrize("tri", [0, 1, 2])
@pytest.mark.parametrize("bfloat", [False, True])
def test_gemm_f64_wide_tile_kernel(dsvgp, gpu_device, tri, bfloat):
    """>= 8192 tiles of 64 x 64 with a dense mn-contiguous right operand: gemm64.hip's 64 x 192 (float B) / 64 x 128 (double B) kernel,
    the forward panel solve's.  Ragged M, N (not a multiple of the tile width) and K, lower / upper triangular left operand, fp64 and
    fp32 outputs -- against torch fp64"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(11 + tri)
    M, N, K = 601, 64 * 830 + 37, 601
    A = torch.randn(K, M + 1, generator=g, dtype=torch.float64)[:, :M]            # op(A)[m][k] = A[k][m]
    if tri == 1:
        A = torch.triu(A)                          # op(A) lower triangular (A_LOWER)
    elif tri == 2:
        A = torch.tril(A)                          # op(A) upper triangular (A_UPPER)
    B = torch.randn(K, N + (3 if bfloat else 1), generator=g, dtype=torch.float32 if bfloat else torch.float64)[:, :N]
    Ad = torch.empty(K, M + 1, dtype=torch.float64, device=gpu_device)[:, :M]
    Ad.copy_(A if tri == 0 else torch.randn(K, M, generator=g, dtype=torch.float64))      # (the masked half must not be read: garbage there)
    if tri:
        keep = torch.triu(torch.ones(K, M, dtype=torch.bool)) if tri == 1 else torch.tril(torch.ones(K, M, dtype=torch.bool))
        Ad.copy_(torch.where(keep.to(gpu_device), A.to(gpu_device), Ad))
    Bd = torch.empty(K, B.shape[1] + (3 if bfloat else 1), dtype=B.dtype, device=gpu_device)[:, :N]
    Bd.copy_(B)
    flags = L.TRANS_A | (L.A_LOWER if tri == 1 else (L.A_UPPER if tri == 2 else 0))
    C = torch.full((M, N), float("nan"), dtype=torch.float64, device=gpu_device)
    C32 = torch.full((M, N), float("nan"), dtype=torch.float32, device=gpu_device)
    ops.gemm(ctx, flags, Ad, Bd, C, alpha=0.75, C32=C32)
    ref = 0.75 * (A.t().to(gpu_device) @ Bd.double())
    assert relmax(C, ref) < 1e-13, (tri, bfloat, relmax(C, ref))
    assert relmax(C32, ref) < 2e-7


@pytest.mark.parametrize("tri", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,padA,padB", [(1601, 2947, 1601, 1, 1), (1600, 2944, 1600, 0, 0), (1985, 2050, 1985, 3, 2)])      # (26 x 47, 25 x 46, 32 x 33 tiles)
def test_gemm_f64_lean_pipelined_kernel(dsvgp, gpu_device, tri, M, N, K, padA, padB):
    """1024 .. 8191 tiles of 64 x 64 with a FLOAT mn-contiguous right operand: gemm64.hip's lean kernel in its software-pipelined
    form (LDS-DMA stages; the mid-size panel solves and the [Q' | a] solve): dense / lower / upper triangular left operand with
    garbage in the masked half, odd and even sizes (16-byte DMA pieces past the last column; the very last k row on the masked path),
    whole and ragged tiles, OUT_LOWER, fp64 and fp32 outputs -- against torch fp64"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(23 + tri + M)
    lda = M + padA + ((M + padA) & 1)                              # (even: 16-byte vector loads)
    ldb = (N + padB + 3) // 4 * 4
    A = torch.randn(K, M, generator=g, dtype=torch.float64)        # op(A)[m][k] = A[k][m]
    keep = None
    if tri == 1:
        keep = torch.triu(torch.ones(K, M, dtype=torch.bool))      # op(A) lower triangular (A_LOWER)
    elif tri == 2:
        keep = torch.tril(torch.ones(K, M, dtype=torch.bool))      # op(A) upper triangular (A_UPPER)
    B = torch.randn(K, N, generator=g)
    Ad = torch.randn(K, lda, generator=g, dtype=torch.float64).to(gpu_device)[:, :M]        # (garbage in the padding too)
    if keep is None:
        Ad.copy_(A)
    else:
        Ad.copy_(torch.where(keep.to(gpu_device), A.to(gpu_device), Ad))                    # (the masked half must not be read)
        A = A * keep
    Bd = torch.randn(K, ldb, generator=g).to(gpu_device)[:, :N]
    Bd.copy_(B)
    flags = L.TRANS_A | (L.A_LOWER if tri == 1 else (L.A_UPPER if tri == 2 else 0))
    ref = 0.75 * (A.t().to(gpu_device) @ Bd.double())
    for lower in (0, L.OUT_LOWER):
        C = torch.full((M, N), float("nan"), dtype=torch.float64, device=gpu_device)
        C32 = torch.full((M, N), float("nan"), dtype=torch.float32, device=gpu_device)
        ops.gemm(ctx, flags | lower, Ad, Bd, C, alpha=0.75, C32=C32)
        want = torch.tril(ref) if lower else ref
        assert relmax(C, want) < 1e-13, (tri, lower, relmax(C, want))
        assert relmax(C32, want) < 2e-7


@pytest.mark.parametrize("tri_a,tri_b", [(0, 2), (2, 0), (0, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("M,K", [(1985, 1985), (2048, 1536), (1985, 3201)])
def test_gemm_f64_lean_pipelined_kernel_double_operands(dsvgp, gpu_device, tri_a, tri_b, M, K):
    """the same kernel with a DOUBLE right operand, a triangular right operand (K range trimmed by the tile column) and split-K (K >= 1536 with
    an fp64 target: chunks of 768 with fp64 atomics) -- the Cholesky backward's products  G1^T L^-1 (B_LOWER)  and  L^-T Y^T (A_UPPER),
    both OUT_LOWER -- against torch fp64"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(41 + 3 * tri_a + tri_b + M + K)
    N = M if (tri_a or tri_b) else M + 130
    def masked(rows, cols, tri):             # tri 1: keep k <= col (op upper of B / "A_LOWER" of op(A)); 2: keep k >= col
        if tri == 0:
            return None
        ones = torch.ones(rows, cols, dtype=torch.bool)
        return torch.triu(ones) if tri == 1 else torch.tril(ones)
    A = torch.randn(K, M, generator=g, dtype=torch.float64)
    B = torch.randn(K, N, generator=g, dtype=torch.float64)
    ka, kb = masked(K, M, tri_a), masked(K, N, tri_b)
    Ad = torch.randn(K, M + (M & 1), generator=g, dtype=torch.float64).to(gpu_device)[:, :M]
    Bd = torch.randn(K, N + (N & 1), generator=g, dtype=torch.float64).to(gpu_device)[:, :N]
    Ad.copy_(A if ka is None else torch.where(ka.to(gpu_device), A.to(gpu_device), Ad))
    Bd.copy_(B if kb is None else torch.where(kb.to(gpu_device), B.to(gpu_device), Bd))
    if ka is not None:
        A = A * ka
    if kb is not None:
        B = B * kb
    flags = L.TRANS_A | (L.A_LOWER if tri_a == 1 else (L.A_UPPER if tri_a == 2 else 0)) | (L.B_UPPER if tri_b == 1 else (L.B_LOWER if tri_b == 2 else 0))
    ref = -0.5 * (A.t().to(gpu_device) @ B.to(gpu_device))
    for lower in ((L.OUT_LOWER,) if N == M else (0,)):
        C = torch.full((M, N), float("nan"), dtype=torch.float64, device=gpu_device)
        ops.gemm(ctx, flags | lower, Ad, Bd, C, alpha=-0.5)
        want = torch.tril(ref) if lower else ref
        assert relmax(C, want) < 1e-13, (tri_a, tri_b, M, K, lower, relmax(C, want))


# ------------------------------------------------------------------ round 4: bf16 x 3 split products (opt-in), widening copy
@pytest.mark.parametrize("R,C,transpose", [(700, 1300, False), (257, 33, False), (1300, 700, True), (33, 257, True)])
def test_split3_planes_reconstruct_the_operand(dsvgp, gpu_device, R, C, transpose):
    """dsvgp_split3_bf16: x = h + m + l with three bf16 planes, remainder below 2^-24 |x| (the two subtractions are exact), planes
    K-blocked ((k / 16) rows_out + row) 16 + k % 16, K padded with zeros to a multiple of 16; plain and transposing form"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(R + C)
    X = (torch.randn(R, C, generator=g) * torch.exp(3 * torch.randn(R, C, generator=g))).to(gpu_device)      # wide dynamic range
    buf = ops.split3_bf16(ctx, X, transpose=transpose)
    torch.cuda.synchronize()
    rows_out, K = (C, R) if transpose else (R, C)
    Kp = int(L.lib.dsvgp_split3_kpad(K))
    assert Kp % 16 == 0 and Kp >= K and Kp - K < 16
    planes = buf[:3 * rows_out * Kp * 2].view(torch.bfloat16).view(3, Kp // 16, rows_out, 16).float()
    rec = planes.sum(0).permute(1, 0, 2).reshape(rows_out, Kp)                      # [rows_out, Kp]
    ref = (X.t() if transpose else X)
    err = ((rec[:, :K].double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-30)).max().item()
    assert err < 2.0 ** -24, err
    assert rec[:, K:].abs().max().item() == 0.0 if Kp > K else True
    # the first plane is the correctly rounded bf16 of the operand
    assert torch.equal(planes[0].permute(1, 0, 2).reshape(rows_out, Kp)[:, :K], ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K,lower", [(700, 1100, 1300, False), (300, 260, 4097, False), (1001, 1000, 6000, True), (513, 512, 700, True), (300, 600, 520, True)])
def test_gemm3b_is_an_fp32_grade_product(dsvgp, gpu_device, M, N, K, lower):
    """csrc/gemm3b.hip: C = alpha A B^T from bf16 plane triples (six bf16 MFMA products per fp32 product, fp32 accumulation) against a
    float64 product of the same float32 operands: error of the largest entry <= 4e-6 (the fp32 MFMA kernel: 3e-5 bound, ~2e-6
    measured), ragged edges, K padding, lower-triangular output with split K"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(gpu_device)
    B = torch.randn(N, K, generator=g).to(gpu_device)
    pa, pb = ops.split3_bf16(ctx, A), ops.split3_bf16(ctx, B)
    C = torch.full((M, N), 5.0, device=gpu_device)
    ops.gemm3b(ctx, L.OUT_LOWER if lower else 0, M, N, K, pa, M, pb, N, C, alpha=0.75)
    ref = 0.75 * A.double() @ B.double().t()
    if lower:
        ref = ref.tril()
        assert C.triu(1).abs().max().item() == 0.0
    err = relmax(C, ref)
    print("gemm3b M=%d N=%d K=%d lower=%d: rel. error %.2e" % (M, N, K, lower, err))
    assert err < 4e-6


def test_widen_f32_f64(dsvgp, gpu_device):
    ops = dsvgp._ops
    ctx = ops.Context.get(gpu_device)
    X = torch.randn(301, 303, device=gpu_device)
    out = torch.full((301, 304), 7.0, dtype=torch.float64, device=gpu_device)
    ops.widen_f32_f64(ctx, X, out[:, :303])
    assert torch.equal(out[:, :303], X.double()) and (out[:, 303] == 7.0).all()
