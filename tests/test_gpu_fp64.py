"""GPU parity tests of the fp64 model mode (``torch.set_default_dtype(torch.float64)`` in the reference's experiment
scripts, experiments/synthetic/exp_script.py:56): the fp64 kernel assembly against the reference-generated golden
vectors and the oracle, the full step (loss, every gradient, predictive moments) against the fp64 oracle, and the
``train_gp`` / ``eval_gp`` drop-in run under a float64 default dtype.

Stated tolerance (everything is double precision on both sides; what remains is summation order and the cond(K_ZZ)
amplification of ~1e3 through the Cholesky backward): kernel entries 1e-12, loss 1e-9 relative, predictive moments 1e-9,
gradients 1e-7 relative in max-norm per parameter."""
import os
import sys

import numpy as np
import pytest
import torch

import dsvgp_oracle as O
from _golden import GOLDEN, GRADIENT, PARAM_KEYS, STRATEGY, gradient_problem, kernel_error, strategy_problem

pytestmark = pytest.mark.gpu
f64 = torch.float64


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def _kernel64(dsvgp, dev, x1, x2, v1, v2, ell, s=1.0, jitter=0.0):
    ops = dsvgp._ops
    ctx = ops.Context.get(dev)
    n1, d = x1.shape
    n2 = x2.shape[0]
    p = v1.shape[0] // n1 if n1 else 0
    hyp = torch.tensor([ell, s, 0.1, 0.0], dtype=f64, device=dev)
    x1d = x1.double().to(dev).contiguous()
    center = x1d.mean(0).contiguous()
    p1 = ops.pack_points_f64(ctx, x1d, v1.double().to(dev).contiguous(), p, hyp, center)
    p2 = ops.pack_points_f64(ctx, x2.double().to(dev).contiguous(), v2.double().to(dev).contiguous(), p, hyp, center)
    return ops.kernel_fwd_f64(ctx, p1, n1, p2, n2, d, p, hyp, jitter=jitter), (ctx, hyp, p1, p2)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_kernel_fwd_f64_matches_reference_golden_vectors(dsvgp, gpu_device, path):
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k])
    if int(g["p"]) > 16:
        pytest.skip("fp64 assembly keeps p <= 16 directions in registers")
    K, _ = _kernel64(dsvgp, gpu_device, t("x1"), t("x2"), t("v1"), t("v2"), float(g["lengthscale"]))
    e_sub, e_sum = kernel_error(K, g)
    print("kernel_fwd_f64 %s: error %.2e (row/col sums %s)" % (os.path.basename(path), e_sub,
                                                                "%.2e" % e_sum if e_sum is not None else "-"))
    assert e_sub < 1e-12 and (e_sum is None or e_sum < 1e-12)


@pytest.mark.parametrize("n1,n2,d,p", [(37, 53, 5, 2), (16, 16, 20, 5), (9, 130, 3, 0), (20, 11, 10, 10), (7, 40, 45, 1),
                                       (130, 7, 2, 1), (5, 6, 17, 16)])
def test_kernel_f64_forward_backward_random_shapes(dsvgp, gpu_device, n1, n2, d, p):
    """forward vs the oracle's pair-wise kernel; backward (x1, v1, lengthscale, outputscale) vs autograd through it"""
    g = torch.Generator().manual_seed(n1 * 1000 + n2)
    x1, x2 = torch.rand(n1, d, generator=g, dtype=f64), torch.rand(n2, d, generator=g, dtype=f64)
    v1 = torch.randn(n1 * p, d, generator=g, dtype=f64)
    v2 = torch.randn(n2 * p, d, generator=g, dtype=f64)
    ell, s = 0.9, 1.7
    K, (ctx, hyp, p1, p2) = _kernel64(dsvgp, gpu_device, x1, x2, v1, v2, ell, s, jitter=0.0)
    xr, vr = x1.clone().requires_grad_(True), v1.clone().requires_grad_(True)
    er, sr = torch.tensor(ell, dtype=f64, requires_grad=True), torch.tensor(s, dtype=f64, requires_grad=True)
    Kref = sr * O.kernel_matrix(xr, x2, vr, v2, er)
    assert relmax(K, Kref.detach()) < 1e-12
    G = torch.randn(K.shape, generator=g, dtype=f64)
    (Kref * G).sum().backward()
    ops = dsvgp._ops
    dx = torch.zeros(n1, d, dtype=f64, device=gpu_device)
    dv = torch.zeros(max(n1 * p, 1), d, dtype=f64, device=gpu_device)[:n1 * p]
    d_hyp = torch.zeros(4, dtype=f64, device=gpu_device)
    ops.kernel_bwd_f64(ctx, G.to(gpu_device), p1, n1, p2, n2, d, p, hyp, False, dx, dv, d_hyp)
    errs = dict(dx=relmax(dx, xr.grad), dell=abs(d_hyp[0].item() - er.grad.item()) / abs(er.grad.item()),
                ds=abs(d_hyp[1].item() - sr.grad.item()) / abs(sr.grad.item()))
    if p:
        errs["dv"] = relmax(dv, vr.grad)
    print("[parity] kernel_bwd_f64 %s: %s" % ((n1, n2, d, p), errs))
    assert max(errs.values()) < 1e-10


def test_kernel_f64_symmetric_backward_and_jitter(dsvgp, gpu_device):
    """K_ZZ: the same points on both sides (gradient flows through both arguments), jitter on the diagonal only"""
    g = torch.Generator().manual_seed(3)
    n, d, p = 23, 6, 3
    x = torch.rand(n, d, generator=g, dtype=f64)
    v = torch.randn(n * p, d, generator=g, dtype=f64)
    K, (ctx, hyp, p1, _) = _kernel64(dsvgp, gpu_device, x, x, v, v, 0.7, 1.3, jitter=1e-3)
    xr, vr = x.clone().requires_grad_(True), v.clone().requires_grad_(True)
    Kref = 1.3 * O.kernel_matrix(xr, xr, vr, vr, torch.tensor(0.7, dtype=f64))
    assert relmax(K, Kref.detach() + 1e-3 * torch.eye(K.shape[0], dtype=f64)) < 1e-12
    G = torch.randn(K.shape, generator=g, dtype=f64)
    G = G + G.t()                                      # the engine's K_ZZ-bar is symmetric
    (Kref * G).sum().backward()
    ops = dsvgp._ops
    dx = torch.zeros(n, d, dtype=f64, device=gpu_device)
    dv = torch.zeros(n * p, d, dtype=f64, device=gpu_device)
    d_hyp = torch.zeros(4, dtype=f64, device=gpu_device)
    ops.kernel_bwd_f64(ctx, G.to(gpu_device), p1, n, p1, n, d, p, hyp, True, dx, dv, d_hyp)
    assert relmax(dx, xr.grad) < 1e-10 and relmax(dv, vr.grad) < 1e-10


def make_problem64(N, d, M, p, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(N, d, generator=g, dtype=f64)
    Y = O.testfun(X)
    Z = X[:M].clone()
    V = torch.eye(d, dtype=f64)[:p].repeat(M, 1) + 0.1 * torch.randn(M * p, d, generator=g, dtype=f64)
    P = O.init_params(Z, V, f64, mean_init_std=0.2, generator=g)
    Mp = M * (p + 1)
    P["chol_variational_covar"] = torch.eye(Mp, dtype=f64) + 0.05 * torch.randn(Mp, Mp, generator=g, dtype=f64)
    P["constant"] = torch.tensor([0.1], dtype=f64)
    P["raw_outputscale"] = torch.tensor(0.2, dtype=f64)
    P["raw_lengthscale"] = torch.tensor([[0.3]], dtype=f64)
    P["raw_noise"] = torch.tensor([-0.5], dtype=f64)
    cols = sorted([0] + (torch.randperm(d, generator=g)[:p] + 1).tolist())
    x = X[M:M + B].contiguous()
    y = Y[M:M + B][:, cols].reshape(-1).contiguous()
    D = torch.eye(d, dtype=f64)[[c - 1 for c in cols[1:]]].repeat(B, 1)
    return P, x, y, D, (d + 1) * N


CASES = [
    # N, d, M, p, B
    (400, 2, 20, 2, 200),      # reference tests/test_dsvgp.py sizes
    (600, 5, 40, 2, 128),      # scaled-down C2
    (500, 20, 30, 5, 96),      # C4 geometry
    (300, 4, 25, 4, 50),       # p == d
    (300, 6, 70, 0, 64),       # p = 0
    (300, 7, 18, 3, 33),       # odd sizes
]


@pytest.mark.parametrize("N,d,M,p,B", CASES)
@pytest.mark.parametrize("mll", ["ELBO", "PLL", "ELBO-gram"])
def test_fp64_step_matches_fp64_oracle(dsvgp, gpu_device, N, d, M, p, B, mll):
    """both formulations of the fp64 step: per-output (ELBO, PLL) and the Gram-matrix ELBO (no per-output variances)"""
    from dsvgp_amd._step64 import ElboEngine64
    fast = mll == "ELBO-gram"
    mll = "ELBO" if fast else mll
    P, x, y, D, nd = make_problem64(N, d, M, p, B, seed=N + d)
    assert all(v.dtype == f64 for v in P.values())
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, nd, mll)
    eng = ElboEngine64(gpu_device)
    eng.fast_min_work = 0                  # (the Gram formulation at test sizes too)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll, fast=fast)
    torch.cuda.synchronize()
    assert loss.dtype == f64 and all(v.dtype == f64 for v in grads.values())
    if fast:
        assert varn.numel() == 0
        varn = var_ref.to(gpu_device)
    errs = {"loss": abs(loss.item() - l_ref.item()) / abs(l_ref.item()), "mu": relmax(mu, mu_ref), "var": relmax(varn, var_ref)}
    assert errs["loss"] < 1e-9 and errs["mu"] < 1e-9 and errs["var"] < 1e-9, errs
    for k in O.PARAM_NAMES:
        if k == "inducing_directions" and p == 0:
            continue
        gk, rk = grads[k], g_ref[k]
        if k == "chol_variational_covar":
            assert torch.triu(gk, 1).abs().max().item() == 0.0
        errs[k] = relmax(gk, rk)
    print("[parity] fp64 step %s %s: %s" % ((N, d, M, p, B), mll, ", ".join("%s %.1e" % kv for kv in errs.items())))
    assert max(errs[k] for k in O.PARAM_NAMES if k in errs) < 1e-7, errs
    # prediction entry point
    mu_p, var_p = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu_p, mu_ref) < 1e-9 and relmax(var_p, var_ref) < 1e-9


def test_fp64_dfree_data_outputs(dsvgp, gpu_device):
    """derivative-free data (DFreeDirectionalGradVariationalStrategy.py:113-136) in the fp64 mode"""
    from dsvgp_amd._step64 import ElboEngine64
    P, x, y, D, nd = make_problem64(300, 5, 24, 2, 60, seed=9)
    y = y.reshape(60, 3)[:, 0].contiguous()
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, nd, "ELBO", data_outputs="values")
    eng = ElboEngine64(gpu_device)
    eng.data_outputs, eng.fast_min_work = "values", 0
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    for fast in (False, True):
        loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO", fast=fast)
        assert abs(loss.item() - l_ref.item()) / abs(l_ref.item()) < 1e-9
        assert relmax(mu, mu_ref) < 1e-9 and (fast or relmax(varn, var_ref) < 1e-9)
        for k in O.PARAM_NAMES:
            assert relmax(grads[k], g_ref[k]) < 1e-7, (fast, k)


def test_fp64_row_shards_add_up_and_joint_covariance(dsvgp, gpu_device):
    """what DataParallel relies on: shard losses / gradients (global row count, KL on one shard) sum to the full-batch step;
    and the joint predictive covariance against the oracle"""
    from dsvgp_amd._step64 import ElboEngine64
    P, x, y, D, nd = make_problem64(300, 5, 24, 2, 60, seed=4)
    eng = ElboEngine64(gpu_device)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xg, yg, Dg = x.to(gpu_device), y.to(gpu_device), D.to(gpu_device)
    for min_work in (eng.fast_min_work, 0):                     # per-output formulation (small problem), then the Gram one
        eng.fast_min_work = min_work
        l_full, g_full, _, _ = eng.loss_and_grads(Pg, xg, yg, Dg, nd)
        g_full = {k: v.clone() for k, v in g_full.items()}
        rows = float(y.shape[0])
        tot_l, tot_g = 0.0, None
        for r, (lo, hi) in enumerate(((0, 25), (25, 60))):
            l, g, _, _ = eng.loss_and_grads(Pg, xg[lo:hi].contiguous(), yg[lo * 3:hi * 3].contiguous(),
                                            Dg[lo * 2:hi * 2].contiguous(), nd, global_rows=rows, include_kl=(r == 0))
            tot_l += l.item()
            tot_g = {k: v.clone() for k, v in g.items()} if tot_g is None else {k: tot_g[k] + g[k] for k in g}
        assert abs(tot_l - l_full.item()) < 1e-12 * abs(l_full.item())
        for k in g_full:
            assert relmax(tot_g[k], g_full[k]) < 1e-10, (min_work, k)
    mu_ref, Sigma_ref = O.predictive_joint(P, x, D)
    _, _, noise = O.constrained(P)
    mu, Sigma = eng.predict_joint(Pg, xg, Dg)
    assert relmax(mu, mu_ref) < 1e-9
    assert relmax(Sigma, Sigma_ref + noise * torch.eye(Sigma_ref.shape[0], dtype=f64)) < 1e-9


@pytest.mark.parametrize("path", [p for p in STRATEGY if "shared" not in p], ids=lambda p: os.path.basename(p))
def test_fp64_predictive_matches_reference_strategy_vectors(dsvgp, gpu_device, path):
    """fp64 engine against the fp64 run of the reference's own strategy forward text: 1e-10"""
    from dsvgp_amd._step64 import ElboEngine64
    P, x, D, fl, mean_ref, cov_ref = strategy_problem(path)
    eng = ElboEngine64(gpu_device)
    eng.data_outputs = fl["outputs"]
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    mu, Sigma = eng.predict_joint(Pg, x.to(gpu_device), D.to(gpu_device))
    noise = torch.nn.functional.softplus(torch.zeros((), dtype=f64)) + 1e-4
    Sigma = Sigma.cpu() - noise * torch.eye(Sigma.shape[0], dtype=f64)
    assert relmax(mu, mean_ref) < 1e-10 and relmax(Sigma, cov_ref) < 1e-10


@pytest.mark.parametrize("path", [p for p in GRADIENT if "shared" not in p], ids=lambda p: os.path.basename(p))
def test_fp64_step_matches_autograd_through_the_reference_forward(dsvgp, gpu_device, path):
    """fp64 engine: loss 1e-10, gradients 1e-7 against autograd through the reference's own strategy forward + kernel file"""
    from dsvgp_amd._step64 import ElboEngine64
    P, x, y, D, nd, fl, loss_ref, g_ref = gradient_problem(path)
    eng = ElboEngine64(gpu_device)
    eng.data_outputs = fl["outputs"]
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    assert abs(loss.item() - loss_ref) < 1e-10 * abs(loss_ref)
    for k in PARAM_KEYS:
        gk = torch.tril(grads[k]) if k == "chol_variational_covar" else grads[k]
        assert relmax(gk, g_ref[k]) < 1e-7, (k, relmax(gk, g_ref[k]))


def test_fp64_mode_refuses_what_it_does_not_cover(dsvgp, gpu_device):
    from dsvgp_amd._step64 import ElboEngine64
    P, x, y, D, nd = make_problem64(200, 3, 10, 1, 20, seed=1)
    eng = ElboEngine64(gpu_device)
    with pytest.raises(dsvgp._lib.DsvgpError):                     # host tensors: no CPU fallback
        eng.loss_and_grads(P, x, y, D, nd)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    with pytest.raises(TypeError):                                 # fp32 data into the fp64 model
        eng.loss_and_grads(Pg, x.float().to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    eng.whitening = "ciq"
    with pytest.raises(NotImplementedError):
        eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)


def test_operator_forward_fp64(dsvgp, gpu_device):
    """RBFKernelDirectionalGrad.forward on float64 inputs (the reference's tests/test_dsvgp.py builds it under a fp64 default)"""
    g = torch.Generator().manual_seed(5)
    n1, n2, d, p = 12, 9, 4, 2
    x1, x2 = torch.rand(n1, d, generator=g, dtype=f64), torch.rand(n2, d, generator=g, dtype=f64)
    v1, v2 = torch.randn(n1 * p, d, generator=g, dtype=f64), torch.randn(n2 * p, d, generator=g, dtype=f64)
    k = dsvgp._rbf_mod.RBFKernelDirectionalGrad().to(device=gpu_device, dtype=f64)
    K = k(x1.to(gpu_device), x2.to(gpu_device), v1=v1.to(gpu_device), v2=v2.to(gpu_device))
    ell = torch.nn.functional.softplus(torch.zeros((), dtype=f64))
    assert K.dtype == f64 and relmax(K, O.kernel_matrix(x1, x2, v1, v2, ell)) < 1e-12
    dg = k(x1.to(gpu_device), x1.to(gpu_device), diag=True, v1=v1.to(gpu_device), v2=v1.to(gpu_device))
    assert relmax(dg, O.kernel_diag(n1, p, ell)) < 1e-14


def test_train_gp_eval_gp_under_float64_default(dsvgp, gpu_device):
    """the reference's experiment setting: torch.set_default_dtype(torch.float64) before building data and model"""
    from torch.utils.data import TensorDataset
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        n, d, M, p, B = 400, 2, 20, 2, 200
        X = torch.rand(n, d)
        Y = O.testfun(X)
        assert X.dtype == f64
        model, lik = dsvgp.train_gp(TensorDataset(X, Y), num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p,
                                    num_epochs=3, seed=0, verbose=False)
        prm = model._param_dict(lik)
        assert all(v.dtype == f64 for v in prm.values())
        from dsvgp_amd._step64 import ElboEngine64
        assert isinstance(model.engine, ElboEngine64)
        # one more step from the trained state: engine (through the harness objects) vs the fp64 oracle
        P = {k: v.detach().cpu().clone() for k, v in prm.items()}
        xb, yb = X[:B], Y[:B][:, [0, 1, 2]].reshape(-1)
        Db = torch.eye(d)[[0, 1]].repeat(B, 1)
        l_ref, g_ref, _, _ = O.elbo_loss_and_grads(P, xb, yb, Db, (d + 1) * n, "ELBO")
        out = lik(model(xb.to(gpu_device), derivative_directions=Db.to(gpu_device)))
        mll = dsvgp.gp_shim.VariationalELBO(lik, model, num_data=(d + 1) * n)
        model.zero_grad(); lik.zero_grad()
        loss = -mll(out, yb.to(gpu_device))
        loss.backward()
        assert abs(loss.item() - l_ref.item()) / abs(l_ref.item()) < 1e-9
        vs = model.variational_strategy
        assert relmax(vs.inducing_points.grad, g_ref["inducing_points"]) < 1e-7
        assert relmax(vs._variational_distribution.chol_variational_covar.grad, g_ref["chol_variational_covar"]) < 1e-7
        means, variances = dsvgp.eval_gp(TensorDataset(X[:50], Y[:50]), model, lik, num_directions=p, minibatch_size=25,
                                         minibatch_dim=p)
        assert means.dtype == f64 and means.shape == (50 * (p + 1),) and bool((variances > 0).all())
        # joint distribution protocol of the BO drivers in double precision
        model.eval(); lik.eval()
        xs, Ds = X[:12].to(gpu_device), torch.eye(d)[:p].repeat(12, 1).to(gpu_device)
        preds = lik(model(xs, derivative_directions=Ds))
        Sig = preds.covariance_matrix
        smp = preds.sample(torch.Size([5]))
        assert Sig.dtype == f64 and smp.dtype == f64 and smp.shape == (5, 12 * (p + 1))
        assert relmax(torch.diagonal(Sig), preds.variance) < 1e-10
        assert relmax(preds.covariance_matrix, Sig) == 0.0            # (sampling did not touch the covariance)
        mu_ref, var_ref = O.predictive(P, X[:50], torch.eye(d)[:p].repeat(50, 1))
        _, _, noise = O.constrained(P)
        assert relmax(means, mu_ref) < 1e-9 and relmax(variances, var_ref + noise) < 1e-9
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("mll,fast", [("ELBO", False), ("ELBO", True), ("PLL", False)])
def test_fp64_natural_parameters_match_oracle(dsvgp, gpu_device, mll, fast):
    """train_gp(use_ngd=True) under a float64 default (the bunny / GNN experiment drivers offer both): q(u) as a
    NaturalVariationalDistribution in the fp64 engine -- loss, predictive moments, ordinary gradients and the
    expectation-parameter gradients of (theta_1, theta_2) against the fp64 oracle"""
    from dsvgp_amd._step64 import ElboEngine64
    from test_ngd import make_ngd_problem
    P, x, y, D, nd = make_ngd_problem(300, 4, 14, 2, 40, seed=7, dtype=f64)
    P["natural_mat"] = 0.5 * (P["natural_mat"] + P["natural_mat"].t())     # (built in fp32: symmetric only to 1e-8; the factorisation
    l_ref, g_ref, mu_ref, var_ref = O.ngd_loss_and_grads(P, x, y, D, nd, mll)       # reads the lower triangle, the oracle the mean)
    eng = ElboEngine64(gpu_device)
    eng.fast_min_work = 0
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll, fast=fast)
    assert set(grads) == set(O.NGD_PARAM_NAMES) and all(v.dtype == f64 for v in grads.values())
    errs = {"loss": abs(loss.item() - l_ref.item()) / abs(l_ref.item()), "mu": relmax(mu, mu_ref)}
    if not fast:
        errs["var"] = relmax(varn, var_ref)
    for k in O.NGD_PARAM_NAMES:
        errs[k] = relmax(grads[k], g_ref[k])
    print("[parity] fp64 natural parameters %s%s: %s" % (mll, " (Gram)" if fast else "", ", ".join("%s %.1e" % kv for kv in errs.items())))
    assert errs["loss"] < 1e-9 and errs["mu"] < 1e-9 and errs.get("var", 0.0) < 1e-9, errs
    assert max(errs[k] for k in O.NGD_PARAM_NAMES) < 1e-7, errs
    mu_p, var_p = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu_p, mu_ref) < 1e-9 and relmax(var_p, var_ref) < 1e-9


def test_train_gp_with_ngd_under_float64_default(dsvgp, gpu_device):
    from torch.utils.data import TensorDataset
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        n, d, M, p, B = 400, 2, 16, 2, 200
        X = torch.rand(n, d)
        Y = O.testfun(X)
        model, lik = dsvgp.train_gp(TensorDataset(X, Y), num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p,
                                    num_epochs=3, seed=0, use_ngd=True, learning_rate_ngd=0.05, verbose=False)
        prm = model._param_dict(lik)
        assert "natural_vec" in prm and all(v.dtype == f64 for v in prm.values())
        from dsvgp_amd._step64 import ElboEngine64
        assert isinstance(model.engine, ElboEngine64)
        P = {k: v.detach().cpu().clone() for k, v in prm.items()}
        xb, yb = X[:B], Y[:B][:, [0, 1, 2]].reshape(-1)
        Db = torch.eye(d)[[0, 1]].repeat(B, 1)
        l_ref, g_ref, _, _ = O.ngd_loss_and_grads(P, xb, yb, Db, (d + 1) * n, "ELBO")
        loss, grads, _, _ = model.engine.loss_and_grads({k: v.to(gpu_device) for k, v in P.items()}, xb.to(gpu_device),
                                                        yb.to(gpu_device), Db.to(gpu_device), (d + 1) * n)
        assert abs(loss.item() - l_ref.item()) / abs(l_ref.item()) < 1e-9
        for k in ("natural_vec", "natural_mat", "inducing_points"):
            assert relmax(grads[k], g_ref[k]) < 1e-7, k
        means, variances = dsvgp.eval_gp(TensorDataset(X[:50], Y[:50]), model, lik, num_directions=p, minibatch_size=25,
                                         minibatch_dim=p)
        assert means.dtype == f64 and bool(torch.isfinite(means).all()) and bool((variances > 0).all())
    finally:
        torch.set_default_dtype(prev)


def test_fp64_model_with_ciq_strategy(dsvgp, gpu_device):
    """train_gp(use_ciq=True) under a float64 default (reference experiments/bunny/exp_bunny.py:66,78): the CIQ step of a float64
    model runs the float64 msMINRES (csrc/ciq.hip ``*_f64``) on fp64 kernel matrices -- the same iteration count as the float64
    oracle and its numbers to 1e-8 / 1e-6 (the fp32 CIQ step is held to 1e-3 / 2e-2 against the same oracle)"""
    from torch.utils.data import TensorDataset
    from dsvgp_amd._step64 import ElboEngine64
    from test_ngd import make_ngd_problem
    P, x, y, D, nd = make_ngd_problem(400, 2, 20, 2, 100, seed=402, dtype=f64)
    st = {}
    l_ref, g_ref, mu_ref, var_ref = O.ciq_loss_and_grads(P, x, y, D, nd, stats=st)
    eng = ElboEngine64(gpu_device)
    eng.whitening = "ciq"
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    assert loss.dtype == f64 and all(v.dtype == f64 for v in grads.values()) and mu.dtype == f64

    def errors(loss, grads, mu, varn):
        e = {"loss": abs(loss.item() - l_ref.item()) / abs(l_ref.item()), "mu": relmax(mu, mu_ref), "var": relmax(varn, var_ref)}
        for k in O.NGD_PARAM_NAMES:
            if g_ref[k].numel() and g_ref[k].abs().max() > 0:
                e["g_" + k] = relmax(grads[k], g_ref[k])
        return e
    # (a) as shipped: the spectrum interval from the engine's own 20 Lanczos steps.  The largest Ritz value is converged
    # (1e-9), the smallest is not (K_ZZ has a cluster of tiny eigenvalues; the 20-step estimate moves by tens of percent with
    # the rounding of the recurrence), so the two quadratures differ at the quadrature's own accuracy
    assert abs(eng.ciq_stats["lmax"] - st["lmax"]) < 1e-9 * st["lmax"] and 0.3 * st["lmin"] < eng.ciq_stats["lmin"] < 3 * st["lmin"]
    assert abs(eng.ciq_stats["iterations"] - st["iterations"]) <= 10
    errs = errors(loss, grads, mu, varn)
    print("[parity] float64 model, float64 msMINRES vs the float64 oracle (own Lanczos bounds): %s" % ", ".join("%s %.2e" % kv for kv in errs.items()))
    assert errs["loss"] < 1e-3 and errs["mu"] < 5e-3 and errs["var"] < 5e-3 and all(v < 2e-2 for k, v in errs.items() if k.startswith("g_")), errs
    # (b) the same step on the oracle's interval: one quadrature, the same iteration count
    eng.ciq_eig_bounds = (st["lmin"], st["lmax"])
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    errs = errors(loss, grads, mu, varn)
    print("[parity] float64 model, float64 msMINRES vs the float64 oracle (oracle's bounds): iterations %d (oracle %d), %s" % (
        eng.ciq_stats["iterations"], st["iterations"], ", ".join("%s %.2e" % kv for kv in errs.items())))
    # measured: loss 1.4e-8, mean 1.5e-5, variance 1.1e-6, gradients <= 1.5e-5 -- not round-off: 30 Lanczos steps on a 60 x 60
    # matrix lose orthogonality, and past that point two float64 recurrences agree to what the iteration has converged to
    # (tests/test_ciq.py::test_ciq_f64_lanczos_solve_mix_cross_match_oracle shows both regimes: 4e-15 at 20 steps on a
    # well-conditioned matrix).  The fp32 CIQ step sits at 1e-3 / 2e-2 against the same oracle.
    assert eng.ciq_stats["iterations"] == st["iterations"]
    assert errs["loss"] < 1e-6 and errs["mu"] < 1e-4 and errs["var"] < 1e-4, errs
    assert all(v < 1e-4 for k, v in errs.items() if k.startswith("g_")), errs
    mu_p, var_p = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert mu_p.dtype == f64 and relmax(mu_p, mu_ref) < 1e-4 and relmax(var_p, var_ref) < 1e-4
    mu_j, Sig_j = eng.predict_joint(Pg, x.to(gpu_device), D.to(gpu_device))
    assert Sig_j.dtype == f64 and relmax(torch.diagonal(Sig_j), var_ref) < 1e-4 and float((Sig_j - torch.diag(torch.diagonal(Sig_j))).abs().max()) == 0.0
    for form in ("forward", "shifts"):                       # the other stackings of the backward's sum over shifts: the same matrix
        eng.ciq_backward_form = form
        _, g2, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
        for k in ("inducing_points", "inducing_directions", "raw_lengthscale", "raw_outputscale"):
            assert relmax(g2[k], grads[k]) < 1e-9, (form, k, relmax(g2[k], grads[k]))
    eng.ciq_backward_form = None
    eng.ciq_eig_bounds = None
    # the harness: a float64 model with the CIQ strategy trains and evaluates
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        n, d, M, p, B = 400, 2, 16, 2, 100
        X = torch.rand(n, d)
        Y = O.testfun(X)
        model, lik = dsvgp.train_gp(TensorDataset(X, Y), num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p,
                                    num_epochs=2, seed=0, use_ngd=True, use_ciq=True, learning_rate_ngd=0.01, verbose=False)
        assert isinstance(model.engine, ElboEngine64) and model.engine.whitening == "ciq"
        assert all(v.dtype == f64 for v in model._param_dict(lik).values())
        means, variances = dsvgp.eval_gp(TensorDataset(X[:50], Y[:50]), model, lik, num_directions=p, minibatch_size=25,
                                         minibatch_dim=p)
        assert means.dtype == f64 and bool(torch.isfinite(means).all()) and bool((variances > 0).all())
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("mll", ["ELBO", "PLL"])
def test_fp64_shared_directions_match_oracle(dsvgp, gpu_device, mll):
    """SharedDirectionalGradVariationalStrategy (the GNN BO driver imports it under its float64 default,
    experiments/GNN_bo/gcn_turbo.py:24,120) in the fp64 engine: one shared direction set, q(u) over M + p values, zero middle term"""
    from dsvgp_amd._step64 import ElboEngine64
    N, d, M, p, B = 400, 4, 14, 2, 80
    P, x, y, D, nd = make_problem64(N, d, M, p, B, seed=21)
    g = torch.Generator().manual_seed(4)
    P["inducing_directions"] = torch.eye(d, dtype=f64)[:p] + 0.2 * torch.randn(p, d, generator=g, dtype=f64)
    P["variational_mean"] = 0.3 * torch.randn(M + p, generator=g, dtype=f64)
    P["chol_variational_covar"] = torch.eye(M + p, dtype=f64) + 0.05 * torch.randn(M + p, M + p, generator=g, dtype=f64)
    l_ref, g_ref, mu_ref, var_ref = O.shared_loss_and_grads(P, x, y, D, nd, mll)
    eng = ElboEngine64(gpu_device)
    eng.shared_directions = True
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll)
    errs = {"loss": abs(loss.item() - l_ref.item()) / abs(l_ref.item()), "mu": relmax(mu, mu_ref), "var": relmax(varn, var_ref)}
    for k in O.PARAM_NAMES:
        assert grads[k].shape == g_ref[k].shape and grads[k].dtype == f64, k
        errs[k] = relmax(grads[k], g_ref[k])
    print("[parity] fp64 shared directions %s: %s" % (mll, ", ".join("%s %.1e" % kv for kv in errs.items())))
    assert errs["loss"] < 1e-9 and errs["mu"] < 1e-9 and errs["var"] < 1e-9, errs
    assert max(errs[k] for k in O.PARAM_NAMES) < 1e-7, errs
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 1e-9 and relmax(varn2, var_ref) < 1e-9


def test_shared_train_gp_under_float64_default(dsvgp, gpu_device):
    from torch.utils.data import TensorDataset
    from dsvgp_amd._step64 import ElboEngine64
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        n, dim, p = 400, 2, 2
        X = torch.rand(n, dim)
        Y = O.testfun(X)
        S = dsvgp.shared_directional_vi
        model, lik = S.train_gp(TensorDataset(X, Y), num_inducing=16, num_directions=p, minibatch_size=200, minibatch_dim=p,
                                num_epochs=3, inducing_data_initialization=False, tqdm=False, seed=2, verbose=False)
        assert isinstance(model.engine, ElboEngine64) and model.engine.shared_directions
        sd = model.state_dict()
        assert sd["variational_strategy.inducing_directions"].shape == (p, dim)
        assert sd["variational_strategy._variational_distribution.variational_mean"].dtype == f64
        means, variances = S.eval_gp(TensorDataset(X[:50], Y[:50]), model, lik, num_directions=p, minibatch_size=25, minibatch_dim=p)
        assert means.dtype == f64 and means.shape == (150,) and bool((variances > 0).all())
    finally:
        torch.set_default_dtype(prev)


def test_fp64_shared_directions_with_natural_parameters(dsvgp, gpu_device):
    """shared directions + NaturalVariationalDistribution over the M + p shared values, all float64 (shared_directional_vi.py:37-39)"""
    from dsvgp_amd._step64 import ElboEngine64
    from test_ngd import make_ngd_problem
    N, d, M, p, B = 400, 4, 14, 2, 80
    P, x, y, D, nd = make_ngd_problem(N, d, M, p, B, seed=33, dtype=f64)
    g = torch.Generator().manual_seed(7)
    P["inducing_directions"] = torch.eye(d, dtype=f64)[:p] + 0.2 * torch.randn(p, d, generator=g, dtype=f64)
    P["natural_vec"] = 0.3 * torch.randn(M + p, generator=g, dtype=f64)
    R = 0.15 * torch.randn(M + p, M + p, generator=g, dtype=f64)
    P["natural_mat"] = -0.5 * (torch.eye(M + p, dtype=f64) + R @ R.t())
    P["natural_mat"] = 0.5 * (P["natural_mat"] + P["natural_mat"].t())
    l_ref, g_ref, mu_ref, var_ref = O.ngd_loss_and_grads(P, x, y, D, nd, "ELBO", forward=O.shared_forward)
    eng = ElboEngine64(gpu_device)
    eng.shared_directions = True
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO")
    assert set(grads) == set(O.NGD_PARAM_NAMES)
    assert abs(loss.item() - l_ref.item()) < 1e-9 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 1e-9 and relmax(varn, var_ref) < 1e-9
    for k in O.NGD_PARAM_NAMES:
        assert grads[k].shape == g_ref[k].shape, k
        if g_ref[k].abs().max() > 0:
            assert relmax(grads[k], g_ref[k]) < 1e-7, (k, relmax(grads[k], g_ref[k]))
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 1e-9 and relmax(varn2, var_ref) < 1e-9


def test_other_harnesses_under_float64_default(dsvgp, gpu_device):
    """grad_svgp (tests/test_grad_svgp.py sizes), dfree_directional_vi and traditional_vi built under a float64 default: fp64 engine,
    float64 parameters and predictions, the loss goes down"""
    import math as _m
    from torch.utils.data import TensorDataset
    from dsvgp_amd._step64 import ElboEngine64
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        n, dim, p = 400, 2, 2
        X, Xt = torch.rand(n, dim), torch.rand(60, dim)
        Y, Yt = O.testfun(X), O.testfun(Xt)

        def first_last_loss(loop_fn):
            import io, contextlib
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                out = loop_fn()
            ls = [float(l.split("loss: ")[1].split(",")[0]) for l in buf.getvalue().splitlines() if l.startswith("Epoch")]
            return out, ls

        G = dsvgp.grad_svgp
        (model, lik), ls = first_last_loss(lambda: G.train_gp(TensorDataset(X, Y), dim, num_inducing=16, minibatch_size=200,
                                                              num_epochs=30, mll_type="PLL", tqdm=False, seed=1))
        assert isinstance(model.engine, ElboEngine64) and len(ls) >= 2 and ls[-1] < ls[0]
        means, variances = G.eval_gp(TensorDataset(Xt, Yt), model, lik, minibatch_size=30)
        assert means.dtype == f64 and means.shape == (60 * 3,) and bool((variances > 0).all())

        F = dsvgp.dfree_directional_vi
        (model, lik), ls = first_last_loss(lambda: F.train_gp(TensorDataset(X, Y[:, 0].contiguous()), num_inducing=16,
                                                              num_directions=p, minibatch_size=200, minibatch_dim=p, num_epochs=30,
                                                              inducing_data_initialization=False, tqdm=False, seed=5))
        assert isinstance(model.engine, ElboEngine64) and len(ls) >= 2 and ls[-1] < ls[0]
        means, variances = F.eval_gp(TensorDataset(Xt, Yt[:, 0].contiguous()), model, lik, num_directions=p, minibatch_size=30,
                                     minibatch_dim=p)
        assert means.dtype == f64 and means.shape == (60,) and bool((variances > 0).all())

        T = dsvgp.traditional_vi
        x1 = torch.rand(300, 1)
        y1 = torch.sin(2 * _m.pi * x1[:, 0]) + 0.05 * torch.randn(300)
        (model, lik), ls = first_last_loss(lambda: T.train_gp(TensorDataset(x1, y1), 1, num_inducing=30, minibatch_size=100,
                                                              num_epochs=40, learning_rate_hypers=0.02, tqdm=False, seed=4))
        assert isinstance(model.engine, ElboEngine64) and len(ls) >= 2 and ls[-1] < ls[0]
        means, variances = T.eval_gp(TensorDataset(x1[:40], y1[:40]), model, lik, minibatch_size=20)
        assert means.dtype == f64 and means.shape == (40,) and bool((variances > 0).all())
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("state", ["init", "mid"])
def test_fp64_model_ciq_step_at_c5_size_measured_against_float64_oracle(dsvgp, gpu_device, state):
    """The reference's bunny experiment runs ``use_ciq=True`` under ``torch.set_default_dtype(torch.float64)``
    (experiments/bunny/exp_bunny.py:66,78).  Here that is the float64 msMINRES of csrc/ciq.hip (``*_f64``) on fp64 kernel matrices,
    held at BASELINE config 5 size (M' = 6144, B' = 3072, Q = 15) to the oracle evaluated in float64
    (tests/golden/c5_step_{init,mid}64.npz, oracle/make_c5_fixture.py init64 / mid64; 190 s and 1530 s of 8 CPU cores), on the
    oracle's spectrum interval (one quadrature for both).
    ``init`` (10 iterations): the same count, everything at round-off -- 1e-9 asserted, measured loss / moments 4e-16, dZ 1.5e-11,
    dV 1.8e-12 (on the fp32 CIQ kernels, through round 3: dZ 7e-3, dV 5e-4, the rest 2e-7).
    ``mid`` (80-90 iterations): the Lanczos vectors have lost orthogonality long before the stopping test passes, so two float64
    recurrences agree to what the iteration has converged to (tolerance 1e-4 on the mean relative update; HIP stops at 80, the
    oracle at 90): measured loss 4e-8, mean 4.6e-4, variance 1.1e-6, gradients <= 1.0e-3 -- asserted 1e-6 / 2e-3 / 1e-5 / 4e-3
    (fp32 kernels: mean 1.2e-3, gradients 2.2e-3)."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from make_c5_fixture import make_inputs, Q
    from dsvgp_amd._step64 import ElboEngine64
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_step_%s64.npz" % state))
    P, x, y, D, nd = make_inputs(state)
    eng = ElboEngine64(gpu_device)
    eng.whitening = "ciq"
    eng.ciq_num_quadrature = Q
    eng.ciq_eig_bounds = (float(g["lmin"]), float(g["lmax"]))        # the oracle's spectrum interval: one quadrature for both
    Pg = {k: v.double().to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.double().to(gpu_device), y.double().to(gpu_device), D.double().to(gpu_device), nd)
    torch.cuda.synchronize()
    assert loss.dtype == f64 and mu.dtype == f64 and all(v.dtype == f64 for v in grads.values())
    t = lambda k: torch.from_numpy(g[k])
    errs = {"loss": abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), "mu": relmax(mu, t("mu")), "var": relmax(varn, t("varn"))}
    gm = grads["natural_mat"]
    errs["g_nm_norm"] = abs(gm.norm().item() - float(g["g_nm_norm"])) / float(g["g_nm_norm"])
    errs["g_nm_block"] = relmax(gm[:96, :96], t("g_nm_block"))
    errs["g_nm_lastrows"] = relmax(gm[-8:, :], t("g_nm_lastrows"))
    for k in O.NGD_PARAM_NAMES:
        if k != "natural_mat" and t("g_" + k).numel() and t("g_" + k).abs().max() > 0:
            errs["g_" + k] = relmax(grads[k], t("g_" + k))
    print("[parity] float64 model, float64 msMINRES, C5 %s vs the float64 oracle: iterations %d (oracle %d), %s" % (
        state, eng.ciq_stats["iterations"], int(g["iterations"]), ", ".join("%s %.2e" % kv for kv in errs.items())))
    if state == "init":
        assert eng.ciq_stats["iterations"] == int(g["iterations"])
        tol_loss, tol_mu, tol_var, tol_g = 1e-9, 1e-9, 1e-9, 1e-9
    else:
        assert abs(eng.ciq_stats["iterations"] - int(g["iterations"])) <= 10
        tol_loss, tol_mu, tol_var, tol_g = 1e-6, 2e-3, 1e-5, 4e-3
    assert errs["loss"] < tol_loss and errs["mu"] < tol_mu and errs["var"] < tol_var, errs
    for k, v in errs.items():
        if k.startswith("g_"):
            assert v < tol_g, (k, v)


def test_fp64_fused_adam_matches_torch_adam(dsvgp, gpu_device):
    """``optim.make_adam`` hands float64 parameters to the hand-written multi-tensor update (dsvgp_adam_step_multi_f64) since round
    4: five steps on three tensors against torch.optim.Adam in float64 (the reference's optimizer, directional_vi.py:189-198)."""
    g = torch.Generator().manual_seed(5)
    shapes = [(37, 5), (130,), (64, 64)]
    p0 = [torch.randn(*s, generator=g, dtype=torch.float64) for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(gpu_device)) for t in p0]
    ref = [torch.nn.Parameter(t.clone().to(gpu_device)) for t in p0]
    o1 = dsvgp.optim.make_adam([{"params": mine[:2]}, {"params": mine[2:]}], lr=0.01)
    assert isinstance(o1, dsvgp.optim.FusedAdam)
    o2 = torch.optim.Adam([{"params": ref[:2]}, {"params": ref[2:]}], lr=0.01)
    for step in range(5):
        for a, b in zip(mine, ref):
            gr = torch.randn(a.shape, generator=g, dtype=torch.float64).to(gpu_device)
            a.grad, b.grad = gr.clone(), gr.clone()
        o1.step(); o2.step()
    torch.cuda.synchronize()
    for a, b in zip(mine, ref):
        assert (a - b).abs().max().item() < 1e-14, (a - b).abs().max().item()
