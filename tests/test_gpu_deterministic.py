"""Deterministic mode (``ElboEngine.deterministic`` / ``dsvgp_set_deterministic``): the CPU reference is deterministic for a
fixed seed; the default HIP step meets its split-K partial sums in floating-point atomics, whose rounding depends on the
order in which workgroups retire.  With the mode on, K slices go to scratch slabs and are added in a fixed order, scalar
sums go through per-workgroup partials, and the step runs on one stream: two runs are BITWISE equal, and the data-parallel
replicas need no re-broadcast (``DataParallel.resync`` is skipped)."""
import pytest
import torch

import dsvgp_oracle as O

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _five_steps(dsvgp, dev, N, d, M, p, B, deterministic, mll="ELBO", fast=None):
    from test_gpu_step import make_problem
    P, _, _, _, nd = make_problem(N, d, M, p, B, seed=11)
    g = torch.Generator().manual_seed(12)
    X = torch.rand(N, d, generator=g)
    Y = O.testfun(X)
    names = list(O.PARAM_NAMES)
    var_names = ("variational_mean", "chol_variational_covar")
    Pd = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in P.items()}
    od = [dsvgp.FusedAdam([Pd[k] for k in var_names], lr=0.01),
          dsvgp.FusedAdam([Pd[k] for k in names if k not in var_names], lr=0.01)]
    eng = dsvgp.ElboEngine(dev)
    eng.deterministic = deterministic
    losses = []
    for step in range(5):
        idx = torch.randperm(N, generator=g)[:B]
        cols = sorted([0] + (torch.randperm(d, generator=g)[:p] + 1).tolist())
        x, y = X[idx].contiguous(), Y[idx][:, cols].reshape(-1).contiguous()
        D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1)
        loss, grads, _, _ = eng.loss_and_grads({k: v.detach() for k, v in Pd.items()}, x.to(dev), y.to(dev), D.to(dev), nd, mll,
                                               fast=fast)
        losses.append(loss.clone())
        for k in names:
            Pd[k].grad = grads[k].clone()
        for o in od:
            o.step()
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in Pd.items()}, {k: v.cpu().clone() for k, v in grads.items()}


# C2 size (small-problem split-K policy of gemm.hip), a C4-geometry step whose Gram product and fp64 M'^3 products split K
# (gemm32.hip / gemm64.hip), and the per-output path (likelihood partials)
@pytest.mark.parametrize("N,d,M,p,B,mll,fast", [(3000, 5, 200, 2, 512, "ELBO", None), (6000, 20, 260, 5, 1024, "ELBO", None),
                                                 (3000, 5, 200, 2, 512, "PLL", False)])
def test_five_steps_are_bitwise_reproducible(dsvgp, gpu_device, N, d, M, p, B, mll, fast):
    l1, P1, g1 = _five_steps(dsvgp, gpu_device, N, d, M, p, B, True, mll, fast)
    l2, P2, g2 = _five_steps(dsvgp, gpu_device, N, d, M, p, B, True, mll, fast)
    assert torch.equal(l1, l2), (l1, l2)
    for k in P1:
        assert torch.equal(P1[k], P2[k]), k
        assert torch.equal(g1[k], g2[k]), k
    # ... and the deterministic step is the same step: against the default (atomics) engine to fp32 reduction noise
    l0, P0, g0 = _five_steps(dsvgp, gpu_device, N, d, M, p, B, False, mll, fast)
    assert relmax(l1, l0) < 1e-5
    for k in g1:
        if g0[k].numel():
            assert relmax(g1[k], g0[k]) < 2e-3, (k, relmax(g1[k], g0[k]))
    print("[parity] deterministic vs default engine after 5 steps (M'=%d, B'=%d, %s): loss %.1e, worst gradient %.1e" % (
        M * (p + 1), B * (p + 1), mll, relmax(l1, l0), max(relmax(g1[k], g0[k]) for k in g1 if g0[k].numel())))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_split_k_slabs_match_the_atomic_products(dsvgp, gpu_device, dt):
    """the products that split K (OUT_LOWER Gram shape, a long-K dense product, an in-place accumulation) through the slab path:
    equal to the float64 reference to rounding, and bitwise equal between two calls"""
    ops, L = dsvgp._ops, dsvgp._lib
    ctx = ops.Context.get(gpu_device)
    g = torch.Generator().manual_seed(5)
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=gpu_device)
    tol = 2e-5 if dt == torch.float32 else 1e-12
    cases = [(300, 300, 6000, L.TRANS_B | L.OUT_LOWER), (257, 190, 4097, L.TRANS_B), (640, 640, 2048, L.TRANS_A)]
    for (M, N, K, flags) in cases:
        ta, tb = bool(flags & L.TRANS_A), bool(flags & L.TRANS_B)
        A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=torch.float64)
        B = torch.randn((N, K) if tb else (K, N), generator=g, dtype=torch.float64)
        ref = (A.t() if ta else A) @ (B.t() if tb else B)
        if flags & L.OUT_LOWER:
            ref = torch.tril(ref)
        Ad, Bd = A.to(dt).to(gpu_device), B.to(dt).to(gpu_device)
        outs = []
        for rep in range(2):
            C = torch.full((M, N), float("nan"), dtype=dt, device=gpu_device)
            ctx.set_deterministic(scratch)
            try:
                ops.gemm(ctx, flags, Ad, Bd, C)
            finally:
                ctx.set_deterministic(None)
            outs.append(C.clone())
        assert torch.equal(outs[0], outs[1])
        assert relmax(outs[0], ref) < tol, (M, N, K, flags, relmax(outs[0], ref))
    # a scratch too small for two slices: the product runs unsplit, still correct
    small = torch.empty(4096, dtype=torch.uint8, device=gpu_device)
    C = torch.empty(300, 300, dtype=dt, device=gpu_device)
    A = torch.randn(300, 6000, generator=g, dtype=torch.float64)
    ctx.set_deterministic(small)
    try:
        ops.gemm(ctx, L.TRANS_B | L.OUT_LOWER, A.to(dt).to(gpu_device), A.to(dt).to(gpu_device), C)
    finally:
        ctx.set_deterministic(None)
    assert relmax(C, torch.tril(A @ A.t())) < tol
