"""CIQ whitening (reference CiqDirectionalGradVariationalStrategy.py:197-295 with the NGD terms of :19-123;
``train_gp(use_ciq=True)``, BASELINE config 5).

The quadrature, msMINRES and the custom backward live in gpytorch 1.4.0 (un-vendored, parity unpinned): the oracle
restates them; CPU tests hold the restatement to what it approximates (the exact K^{-1/2} through an eigendecomposition
and plain autograd), GPU tests hold the HIP path (csrc/ciq.hip + _step.ElboEngine._ciq_step) to the oracle.
Tolerances: the solver stops at a mean relative update of 1e-4 tested every 10 iterations, and both sides estimate the
spectrum from 20 Lanczos steps, so HIP (fp32) and oracle agree to ~1e-3 of the max magnitude, CIQ and the exact
whitening to ~1e-2 (the Lanczos estimate of lambda_min is an overestimate by construction)."""
import math

import pytest
import torch

import dsvgp_oracle as O
from test_ngd import make_ngd_problem, relmax


def spd(n, cond=1e3, seed=0):
    g = torch.Generator().manual_seed(seed)
    U, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    lam = torch.logspace(0, math.log10(cond), n, dtype=torch.float64) * 1e-2
    return (U * lam) @ U.t(), lam


# ------------------------------------------------------------------ oracle (CPU)
def test_quadrature_approximates_inverse_square_root():
    lmin, lmax = 1e-3, 50.0
    sigma, omega = O.ciq_quadrature(lmin, lmax, 15)
    assert (sigma > 0).all() and (omega > 0).all()
    lam = torch.logspace(math.log10(lmin), math.log10(lmax), 200, dtype=torch.float64)
    approx = (omega.unsqueeze(1) / (lam.unsqueeze(0) + sigma.unsqueeze(1))).sum(0)
    assert ((approx - lam.rsqrt()).abs() * lam.sqrt()).max() < 1e-5          # relative error of the rational approximation


def test_msminres_solves_every_shift():
    K, _ = spd(60, 1e3)
    g = torch.Generator().manual_seed(1)
    R = torch.randn(60, 7, generator=g, dtype=torch.float64)
    sigma = torch.tensor([0.0, 0.01, 0.5, 3.0], dtype=torch.float64)
    X, its = O.msminres(K, R, sigma, tol=1e-10, max_iter=200)
    for q in range(4):
        ref = torch.linalg.solve(K + sigma[q] * torch.eye(60, dtype=torch.float64), R)
        assert relmax(X[q], ref) < 1e-6
    assert its <= 200


def test_sqrt_inv_matmul_and_its_backward_against_exact():
    K, lam = spd(50, 1e3, seed=3)
    g = torch.Generator().manual_seed(2)
    R = torch.randn(50, 9, generator=g, dtype=torch.float64)
    Kc, Rc = K.clone().requires_grad_(True), R.clone().requires_grad_(True)
    Ke, Re = K.clone().requires_grad_(True), R.clone().requires_grad_(True)
    st = {}
    T = O.sqrt_inv_matmul(Kc, Rc, 15, st)
    Te = O.sqrt_inv_matmul_exact(Ke, Re)
    assert st["lmax"] <= lam.max() * 1.0001 and st["lmin"] >= lam.min() * 0.999   # Ritz values lie inside the spectrum
    assert relmax(T, Te) < 2e-2
    W = torch.randn(50, 9, generator=g, dtype=torch.float64)
    (T * W).sum().backward()
    (Te * W).sum().backward()
    assert relmax(Rc.grad, Re.grad) < 2e-2
    assert relmax(Kc.grad, 0.5 * (Ke.grad + Ke.grad.t())) < 5e-2


def test_ciq_step_close_to_exact_whitening_and_plain_autograd():
    P, x, y, D, nd = make_ngd_problem(300, 3, 10, 2, 24, dtype=torch.float64)
    st = {}
    l, g, mu, var = O.ciq_loss_and_grads(P, x, y, D, nd, stats=st)
    l2, g2, mu2, var2 = O.ciq_loss_and_grads(P, x, y, D, nd, exact=True)
    assert st["iterations"] <= 100
    assert abs(l.item() - l2.item()) < 1e-3 * abs(l2.item())
    assert relmax(mu, mu2) < 1e-2 and relmax(var, var2) < 1e-2
    for k in g:
        if g[k].numel():
            assert relmax(g[k], g2[k]) < 3e-2, k


def test_ngd_interp_terms_match_the_reference_function():
    """the oracle's restatement against vectors produced by the reference file's OWN autograd function ``_NgdInterpTerms``
    (CiqDirectionalGradVariationalStrategy.py:19-123; forward and hand-written backward), oracle/make_strategy_fixtures.py"""
    import glob, os
    import numpy as np
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_ngd_interp_*.npz")))
    assert len(paths) >= 2
    for path in paths:
        g = np.load(path)
        t = lambda k: torch.from_numpy(g[k])
        T, nv, nm = (t(k).clone().requires_grad_(True) for k in ("interp_term", "natural_vec", "natural_mat"))
        im, iv, kl = O._NgdInterpTermsFn.apply(T, nv, nm)
        assert relmax(im, t("interp_mean")) < 1e-12 and relmax(iv, t("interp_var")) < 1e-12 and kl.item() == float(g["kl_div"]) == 0.0
        ((im * t("g_mean")).sum() + (iv * t("g_var")).sum() + kl * t("g_kl")).backward()
        assert relmax(T.grad, t("d_interp_term")) < 1e-12
        assert relmax(nv.grad, t("d_natural_vec")) < 1e-12
        assert relmax(nm.grad, t("d_natural_mat")) < 1e-12


def test_ciq_predictive_matches_the_reference_strategy_forward():
    """``ciq_predictive`` (exact K^-1/2: the limit of the quadrature) against the reference's CIQ strategy ``forward`` text run
    with the exact inverse square root in place of gpytorch's ``sqrt_inv_matmul`` (CiqDGVS.py:197-295)"""
    import glob, os
    import numpy as np
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_ciq_[0-9].npz")))
    assert len(paths) >= 2
    for path in paths:
        g = np.load(path)
        t = lambda k: torch.from_numpy(g[k])
        inv_softplus = lambda v: float(np.log(np.expm1(float(v))))
        P = dict(inducing_points=t("Z"), inducing_directions=t("V"), natural_vec=t("natural_vec"), natural_mat=t("natural_mat"),
                 constant=torch.tensor([float(g["constant"])], dtype=torch.float64),
                 raw_outputscale=torch.tensor(inv_softplus(g["outputscale"]), dtype=torch.float64),
                 raw_lengthscale=torch.tensor([[inv_softplus(g["lengthscale"])]], dtype=torch.float64),
                 raw_noise=torch.tensor([0.0], dtype=torch.float64))
        mu, var, kl = O.ciq_predictive(P, t("x"), t("D"), exact=True)
        assert relmax(mu, t("mean")) < 1e-9 and relmax(var, t("variance")) < 1e-9, path
        # ... and the quadrature itself stays within its tolerance of that limit
        mu_q, var_q, _ = O.ciq_predictive(P, t("x"), t("D"))
        assert relmax(mu_q, t("mean")) < 2e-3 and relmax(var_q, t("variance")) < 2e-3


# ------------------------------------------------------------------ HIP kernels vs oracle
def _ciq_grad_problem(path, dtype=torch.float64):
    import numpy as np
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k]).to(dtype)
    P = dict(inducing_points=t("Z"), inducing_directions=t("V"), natural_vec=t("natural_vec"), natural_mat=t("natural_mat"),
             constant=t("constant"), raw_outputscale=t("raw_outputscale"), raw_lengthscale=t("raw_lengthscale"), raw_noise=t("raw_noise"))
    ref = {k: torch.from_numpy(g["d_" + k]) for k in O.NGD_PARAM_NAMES}
    return P, t("x"), t("y"), t("D"), float(g["num_data"]), float(g["loss"]), ref


def test_ciq_gradients_match_autograd_through_the_reference_forward():
    """the oracle's CIQ step (exact K^-1/2) against torch autograd run THROUGH the reference's CIQ strategy forward, its own
    ``_NgdInterpTerms`` backward and the reference kernel file (tests/golden/strategy_ciq_grad_*.npz)"""
    import glob, os
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_ciq_grad_*.npz")))
    assert paths
    for path in paths:
        P, x, y, D, nd, loss_ref, g_ref = _ciq_grad_problem(path)
        loss, grads, _, _ = O.ciq_loss_and_grads(P, x, y, D, nd, exact=True)
        assert abs(loss.item() - loss_ref) < 1e-10 * abs(loss_ref)
        for k in O.NGD_PARAM_NAMES:
            assert relmax(grads[k], g_ref[k]) < 1e-7, (k, relmax(grads[k], g_ref[k]))


@pytest.mark.gpu
def test_ciq_lanczos_and_solve_match_oracle(dsvgp, gpu_device):
    ops = dsvgp._ops
    dev = gpu_device
    ctx = ops.Context.get(dev)
    n, t, Q = 96, 40, 15
    K, lam = spd(n, 1e3, seed=5)
    g = torch.Generator().manual_seed(6)
    R = torch.randn(t, n, generator=g, dtype=torch.float64)          # one right-hand side per row
    K32 = K.float().to(dev).contiguous()
    R32 = R.float().to(dev).contiguous()
    # Lanczos coefficients -> the same Ritz bounds as the oracle
    alpha, beta = ops.ciq_lanczos(ctx, K32, R32[0].contiguous(), 20)
    a, b = alpha.double().cpu(), beta.double().cpu()
    Tm = torch.diag(a) + torch.diag(b[:19], 1) + torch.diag(b[:19], -1)
    eigs = torch.linalg.eigvalsh(Tm)
    lmin, lmax = O.lanczos_eig_bounds(K.float().double(), R.float().double()[0])
    assert abs(eigs.max().item() - lmax) < 1e-4 * lmax and abs(eigs.min().item() - lmin) < 2e-2 * lmin
    sigma, omega = O.ciq_quadrature(lmin, lmax, Q)
    out = torch.empty(t, n, device=dev)
    sig32, om32 = sigma.float().to(dev), omega.float().to(dev)

    def solve(cap):
        basis = torch.empty(cap + 1, t, n, device=dev)
        ycoef = torch.empty(t, cap, ops.ciq_qp(Q), device=dev)
        rnorm = torch.empty(t, device=dev)
        ws = torch.empty(int(dsvgp._lib.lib.dsvgp_ciq_workspace_bytes(Q, t, n, cap)), dtype=torch.uint8, device=dev)
        return ops.ciq_solve(ctx, K32, R32, sig32, om32, basis, ycoef, rnorm, out, ws), basis, ycoef, rnorm

    assert solve(10)[0] is None                                      # DSVGP_ENOSPACE: more than 10 Lanczos rows are needed
    its, basis, ycoef, rnorm = solve(200)
    # the per-shift solves are not stored: materialise them from the basis and the coefficient table
    X = ops.ciq_mix(ctx, basis, its, ycoef, Q, rnorm, torch.empty(Q, t, n, device=dev))
    Xr, its_r = O.msminres(K, R.t().contiguous(), sigma)             # oracle: columns
    assert abs(its - its_r) <= 10
    for q in (0, 7, 14):                                              # fp32 vs fp64, stopped within 10 iterations of each other
        assert relmax(X[q].t(), Xr[q]) < 1e-2
    T_ref = (omega.reshape(-1, 1, 1) * Xr).sum(0)
    assert relmax(out.t(), T_ref) < 5e-3
    # and against the exact inverse square root (quadrature + solver error)
    assert relmax(out.t(), O.sqrt_inv_matmul_exact(K, R.t().contiguous())) < 2e-2


@pytest.mark.gpu
def test_ciq_f64_lanczos_solve_mix_cross_match_oracle(dsvgp, gpu_device):
    """The double-precision entry points (dsvgp_ciq_*_f64: the CIQ strategy of a float64 model): Ritz bounds, iteration count,
    every shifted solve and the quadrature sum against the float64 oracle at round-off level; mix / cross against tensor
    expressions (vector width 2 and 1)."""
    ops = dsvgp._ops
    dev = gpu_device
    ctx = ops.Context.get(dev)
    f64 = torch.float64
    # (n, cond): 70 iterations on a 96 x 96 matrix lose the orthogonality of the Lanczos vectors, after which two float64
    # implementations agree only at the level the iteration has converged to; 20 iterations on a well-conditioned 256 x 256
    # matrix stay at round-off
    for n, cond, tol_x in ((96, 1e3, 2e-3), (256, 30.0, 1e-10)):
        t, Q = 40, 15
        K, lam = spd(n, cond, seed=5)
        g = torch.Generator().manual_seed(6)
        R = torch.randn(t, n, generator=g, dtype=f64)
        Kd, Rd = K.to(dev).contiguous(), R.to(dev).contiguous()
        alpha, beta = ops.ciq_lanczos(ctx, Kd, Rd[0].contiguous(), 20)
        assert alpha.dtype == f64
        a, b = alpha.cpu(), beta.cpu()
        eigs = torch.linalg.eigvalsh(torch.diag(a) + torch.diag(b[:19], 1) + torch.diag(b[:19], -1))
        lmin, lmax = O.lanczos_eig_bounds(K, R[0])
        assert abs(eigs.max().item() - lmax) < 1e-10 * lmax and abs(eigs.min().item() - lmin) < 1e-8 * lmin
        sigma, omega = O.ciq_quadrature(lmin, lmax, Q)
        out = torch.empty(t, n, device=dev, dtype=f64)
        cap = 200
        basis = torch.empty(cap + 1, t, n, device=dev, dtype=f64)
        ycoef = torch.empty(t, cap, ops.ciq_qp(Q), device=dev, dtype=f64)
        rnorm = torch.empty(t, device=dev, dtype=f64)
        ws = torch.empty(ops.ciq_workspace_bytes(Q, t, n, cap, f64), dtype=torch.uint8, device=dev)
        its = ops.ciq_solve(ctx, Kd, Rd, sigma.to(dev), omega.to(dev), basis, ycoef, rnorm, out, ws)
        Xr, its_r = O.msminres(K, R.t().contiguous(), sigma)
        assert its == its_r
        X = ops.ciq_mix(ctx, basis, its, ycoef, Q, rnorm, torch.empty(Q, t, n, device=dev, dtype=f64))
        ex = max(relmax(X[q].t(), Xr[q]) for q in (0, 7, 14))
        eo = relmax(out.t(), (omega.reshape(-1, 1, 1) * Xr).sum(0))
        print("[parity] float64 msMINRES n=%d cond=%.0e: %d iterations (oracle %d), shifted solves %.1e, quadrature sum %.1e" % (
            n, cond, its, its_r, ex, eo))
        assert ex < tol_x and eo < tol_x
    with pytest.raises(TypeError):
        ops.ciq_solve(ctx, Kd, Rd, sigma.float().to(dev), omega.to(dev), basis, ycoef, rnorm, out, ws)
    for (t2, n2, J, Kout) in ((9, 37, 21, 19), (6, 64, 3, 1)):
        g = torch.Generator().manual_seed(t2 * 100 + n2)
        ld, KP = J + 2, ops.ciq_qp(Kout)
        bs = torch.randn(J + 1, t2, n2, generator=g, dtype=f64)
        Cc = torch.randn(t2, ld, KP, generator=g, dtype=f64)
        scale = torch.rand(t2, generator=g, dtype=f64) + 0.5
        o2 = ops.ciq_mix(ctx, bs.to(dev), J, Cc.to(dev), Kout, scale.to(dev), torch.empty(Kout, t2, n2, device=dev, dtype=f64))
        ref = torch.einsum("rjk,jrn->krn", Cc[:, :J, :Kout], bs[:J]) * scale[None, :, None]
        assert relmax(o2, ref) < 1e-13
        Q2, Ja, Jb = 5, min(J, 6), J
        QP = ops.ciq_qp(Q2)
        ya, yb = torch.randn(t2, Ja + 1, QP, generator=g, dtype=f64), torch.randn(t2, Jb + 3, QP, generator=g, dtype=f64)
        om = torch.rand(Q2, generator=g, dtype=f64)
        rn_a, rn_b = torch.rand(t2, generator=g, dtype=f64) + 0.5, torch.rand(t2, generator=g, dtype=f64) + 0.5
        ctab = ops.ciq_cross(ctx, ya.to(dev), Ja, yb.to(dev), Jb, om.to(dev), rn_a.to(dev), rn_b.to(dev))
        ref = torch.einsum("q,riq,rjq->rji", om, ya[:, :Ja, :Q2], yb[:, :Jb, :Q2]) * (rn_a * rn_b)[:, None, None]
        assert ctab.dtype == f64 and relmax(ctab[:, :, :Ja], ref) < 1e-13
    A = torch.randn(70, 70, dtype=f64, generator=g).to(dev)
    o = torch.empty_like(A)
    ops.sym_average_f64(ctx, A, o)
    assert torch.equal(o, 0.5 * (A + A.t()))


@pytest.mark.gpu
@pytest.mark.parametrize("t,n,J,Kout", [(13, 40, 7, 5), (9, 37, 21, 19), (6, 64, 3, 1)])
def test_ciq_mix_and_cross_against_plain_tensor_expressions(dsvgp, gpu_device, t, n, J, Kout):
    """dsvgp_ciq_mix: out[k, row] = scale_row sum_j C[row, j, k] basis[j, row]; dsvgp_ciq_cross: the per-row coefficients that turn
    sum_q omega_q A_q^T B_q into one stacked product -- both against torch expressions (vector widths 4 and 1, > 16 outputs)"""
    ops = dsvgp._ops
    dev = gpu_device
    ctx = ops.Context.get(dev)
    g = torch.Generator().manual_seed(t * 100 + n)
    ld = J + 2
    KP = ops.ciq_qp(Kout)
    basis = torch.randn(J + 1, t, n, generator=g)
    C = torch.randn(t, ld, KP, generator=g)
    scale = torch.rand(t, generator=g) + 0.5
    out = ops.ciq_mix(ctx, basis.to(dev), J, C.to(dev), Kout, scale.to(dev), torch.empty(Kout, t, n, device=dev))
    ref = torch.einsum("rjk,jrn->krn", C[:, :J, :Kout].double(), basis[:J].double()) * scale.double()[None, :, None]
    assert relmax(out, ref) < 2e-6
    # cross coefficients: A_q = rn_a mix(basisA, ya)_q, B_q = rn_b mix(basisB, yb)_q, Q shifts
    Q, Ja, Jb = 5, min(J, 6), J
    QP = ops.ciq_qp(Q)
    ya, yb = torch.randn(t, Ja + 1, QP, generator=g), torch.randn(t, Jb + 3, QP, generator=g)
    omega = torch.rand(Q, generator=g)
    rn_a, rn_b = torch.rand(t, generator=g) + 0.5, torch.rand(t, generator=g) + 0.5
    basisA = torch.randn(Ja + 1, t, n, generator=g)
    ctab = ops.ciq_cross(ctx, ya.to(dev), Ja, yb.to(dev), Jb, omega.to(dev), rn_a.to(dev), rn_b.to(dev))
    assert ctab.shape == (t, Jb, ops.ciq_qp(Ja))
    Z = ops.ciq_mix(ctx, basis.to(dev), Jb, ctab, Ja, None, torch.empty(Ja, t, n, device=dev))
    stacked = basisA[:Ja].reshape(Ja * t, n).double().t() @ Z.double().cpu().reshape(Ja * t, n)
    Aq = torch.einsum("rjq,jrn->qrn", ya[:, :Ja, :Q].double(), basisA[:Ja].double()) * rn_a.double()[None, :, None]
    Bq = torch.einsum("rjq,jrn->qrn", yb[:, :Jb, :Q].double(), basis[:Jb].double()) * rn_b.double()[None, :, None]
    direct = sum(omega[q].double() * Aq[q].t() @ Bq[q] for q in range(Q))
    assert relmax(stacked, direct) < 1e-5


@pytest.mark.gpu
def test_ciq_backward_forms_and_basis_growth_agree(dsvgp, gpu_device):
    """the backward's sum over shifts stacked over the backward basis, the forward basis or the materialised solves is the
    same matrix; a basis sized too small is grown (ENOSPACE -> twice the rows) without changing the result"""
    P, x, y, D, nd = make_ngd_problem(400, 2, 20, 2, 100, seed=402)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    args = (x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    res = {}
    for form, cap in (("backward", 20), ("forward", 20), ("shifts", 20), (None, 4)):
        eng = dsvgp.ElboEngine(gpu_device, trsm_nb=4096)
        eng.whitening, eng.ciq_backward_form, eng.ciq_capacity = "ciq", form, cap
        loss, grads, mu, varn = eng.loss_and_grads(Pg, *args)
        res[(form, cap)] = (loss.item(), {k: grads[k].clone() for k in O.NGD_PARAM_NAMES}, eng.ciq_stats["iterations"])
        assert eng.ciq_capacity >= eng.ciq_stats["iterations"] and eng.ciq_capacity >= eng.ciq_stats["iterations_backward"]
    l0, g0, it0 = res[("backward", 20)]
    for key, (l, g, it) in res.items():
        assert it == it0 and abs(l - l0) <= 1e-6 * abs(l0), key
        for k in O.NGD_PARAM_NAMES:
            if g0[k].numel() and g0[k].abs().max() > 0:
                assert relmax(g[k], g0[k]) < 2e-4, (key, k, relmax(g[k], g0[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,M,p,B", [(400, 2, 20, 2, 100), (500, 20, 24, 5, 48)])
def test_ciq_step_matches_oracle(dsvgp, gpu_device, N, d, M, p, B):
    P, x, y, D, nd = make_ngd_problem(N, d, M, p, B, seed=N + d)
    st = {}
    l_ref, g_ref, mu_ref, var_ref = O.ciq_loss_and_grads(P, x, y, D, nd, stats=st)
    eng = dsvgp.ElboEngine(gpu_device, trsm_nb=4096)
    eng.whitening = "ciq"
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    assert abs(eng.ciq_stats["lmax"] - st["lmax"]) < 1e-3 * st["lmax"]
    assert abs(eng.ciq_stats["iterations"] - st["iterations"]) <= 10
    assert abs(loss.item() - l_ref.item()) < 1e-3 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 5e-3 and relmax(varn, var_ref) < 5e-3
    for k in O.NGD_PARAM_NAMES:
        if g_ref[k].numel() and g_ref[k].abs().max() > 0:
            assert relmax(grads[k], g_ref[k]) < 2e-2, k
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 5e-3 and relmax(varn2, var_ref) < 5e-3
    # Cholesky variational parameters are rejected like the unreachable branch of the reference harness
    Pc = {k: v for k, v in Pg.items() if not k.startswith("natural_")}
    Pc["variational_mean"] = torch.zeros(M * (p + 1), device=gpu_device)
    Pc["chol_variational_covar"] = torch.eye(M * (p + 1), device=gpu_device)
    with pytest.raises(NotImplementedError):
        eng.loss_and_grads(Pc, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)


@pytest.mark.gpu
def test_train_gp_use_ciq_drop_in(dsvgp, gpu_device, capsys):
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, p = 400, 2, 2
    train_x = torch.rand(n, dim)
    train_y = O.testfun(train_x)
    model, likelihood = dsvgp.train_gp(TensorDataset(train_x, train_y), num_inducing=16, num_directions=p,
                                       minibatch_size=100, minibatch_dim=p, num_epochs=15, use_ciq=True,
                                       learning_rate_ngd=0.1, num_contour_quadrature=15, tqdm=False, seed=3)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 2 and all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
    sd = model.state_dict()
    assert "variational_strategy._variational_distribution.natural_mat" in sd
    assert "variational_strategy.updated_strategy" not in sd
    assert model.engine.whitening == "ciq" and model.engine.ciq_stats["iterations"] >= 10
    means, variances = dsvgp.eval_gp(TensorDataset(train_x[:60], train_y[:60]), model, likelihood,
                                     num_directions=p, minibatch_size=30, minibatch_dim=p)
    assert means.shape == (180,) and (variances > 0).all() and torch.isfinite(means).all()
    # joint distribution under NGD-CIQ: the reference's q(f) is MultivariateNormal(mean, DiagLazyTensor(var)) (CiqDGVS.py:264-267)
    model.eval(); likelihood.eval()
    x = train_x[:20].to(gpu_device)
    D = torch.eye(dim, device=gpu_device)[:p].repeat(20, 1)
    preds = likelihood(model(x, derivative_directions=D))
    Sigma = preds.covariance_matrix
    assert Sigma.shape == (60, 60) and relmax(torch.diagonal(Sigma), preds.variance) < 1e-6
    assert (Sigma - torch.diag(torch.diagonal(Sigma))).abs().max().item() == 0.0
    torch.manual_seed(1)
    smp = preds.sample(torch.Size([3000]))
    assert smp.shape == (3000, 60)
    assert relmax(smp.mean(0), preds.mean) < 0.1 and relmax(smp.var(0), preds.variance) < 0.15


@pytest.mark.gpu
@pytest.mark.parametrize("state", ["init", "mid"])
def test_c5_full_size_step_against_committed_oracle_vector(dsvgp, gpu_device, state):
    """BASELINE config 5 at FULL size: CIQ-whitened DSVGP d=50, M=1024, p=5 -> M'=6144, B=512 -> B'=3072, Q=15
    (reference CiqDirectionalGradVariationalStrategy.py:19-123,197-295) against the oracle runs committed as
    tests/golden/c5_step_{init,mid}.npz (oracle/make_c5_fixture.py; inputs regenerated from the seed).
    Tolerances as in test_ciq_step_matches_oracle: both sides stop msMINRES at a mean relative update of 1e-4 tested every
    10 iterations and take the spectrum from 20 Lanczos steps in fp32; stated (round 3): loss 1e-5, mean / variance 3e-3,
    gradients 6e-3 of the max magnitude per parameter (the small-size CIQ tests keep loss 1e-3, moments 5e-3, gradients 2e-2)."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from make_c5_fixture import make_inputs, Q
    path = os.path.join(os.path.dirname(__file__), "golden", "c5_step_%s.npz" % state)
    g = np.load(path)
    P, x, y, D, nd = make_inputs(state)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.whitening = "ciq"
    eng.ciq_num_quadrature = Q
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    torch.cuda.synchronize()
    t = lambda k: torch.from_numpy(g[k])
    errs = {"lmax": abs(eng.ciq_stats["lmax"] - float(g["lmax"])) / float(g["lmax"]),
            "lmin": abs(eng.ciq_stats["lmin"] - float(g["lmin"])) / float(g["lmin"]),
            "loss": abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])),
            "mu": relmax(mu, t("mu")), "var": relmax(varn, t("varn"))}
    gm = grads["natural_mat"]
    errs["g_nm_norm"] = abs(gm.double().norm().item() - float(g["g_nm_norm"])) / float(g["g_nm_norm"])
    errs["g_nm_block"] = relmax(gm[:96, :96], t("g_nm_block"))
    errs["g_nm_diag"] = relmax(torch.diagonal(gm), t("g_nm_diag"))
    errs["g_nm_lastrows"] = relmax(gm[-8:, :], t("g_nm_lastrows"))
    for k in O.NGD_PARAM_NAMES:
        if k != "natural_mat" and t("g_" + k).numel() and t("g_" + k).abs().max() > 0:
            errs["g_" + k] = relmax(grads[k], t("g_" + k))
    print("[parity] C5 %s: iterations %d (oracle %d), %s" % (state, eng.ciq_stats["iterations"], int(g["iterations"]),
                                                            ", ".join("%s %.2e" % kv for kv in errs.items())))
    assert abs(eng.ciq_stats["iterations"] - int(g["iterations"])) <= 10
    # lambda_min is the smallest Ritz value of 20 fp32 Lanczos steps at n = 6144 (orthogonality already lost): it moves by
    # ~10 % between two fp32 implementations while the quadrature built on it reproduces every output below to ~1e-3
    assert errs["lmax"] < 1e-3 and errs["lmin"] < 0.25
    # (tightened in round 3 to ~3x the measured errors: loss 1.8e-7, mu 7.5e-4, var 1.8e-6, worst gradient 1.9e-3 -- both sides stop
    #  msMINRES at a mean relative update of 1e-4 after the same 90 iterations)
    assert errs["loss"] < 1e-5 and errs["mu"] < 3e-3 and errs["var"] < 3e-3, errs
    assert errs["g_nm_norm"] < 6e-3
    for k, v in errs.items():
        if k.startswith("g_"):
            # init state only: at lengthscale = 1/1024 the x / lengthscale coordinates are O(500) and the r . v inner products
            # of coincident points (the 8 inducing rows inside the batch) cancel in fp32 to ~1e-4 instead of 0, which the
            # 1 / lengthscale^2 = 1e6 factors of the kernel backward amplify.  The committed vector comes from the oracle's
            # pair-wise DIFFERENCE form (exact zeros); the reference's own matmul op sequence run in fp32 on the same state
            # (M = 64 replica, oracle kernel_matrix_refseq vs fp64) is off by 0.32 (dZ) and 0.28 (dV); the HIP kernels measure
            # 0.15 and 0.01.  So these two gradients are held to the reference's own fp32 round-off here, and to 2e-2 in the
            # well-scaled "mid" state.
            loose = state == "init" and k in ("g_inducing_points", "g_inducing_directions")
            if not loose:
                assert v < 6e-3, (k, v)
    if state == "init":
        # dZ / dV at the init state, numerically: against the FLOAT64 oracle (tests/golden/c5_init_refseq.npz,
        # oracle/make_c5_refseq_fixture.py) the HIP step must be no worse than the reference's OWN kernel op sequence evaluated in
        # fp32 on the CPU (kernel_matrix_refseq: 0.343 / 0.0011 off float64), and within 5e-2 / 5e-3 of float64 outright.
        # Measured on MI355X (round 3): 7.0e-3 / 4.7e-4 -- the distance of 0.15 / 0.01 to the committed fp32 vector
        # (c5_step_init.npz, pair-wise form) seen in the loop above is that vector's own round-off (0.148 / 0.0104 off float64).
        r = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_init_refseq.npz"))
        for k in ("inducing_points", "inducing_directions"):
            e_hip = relmax(grads[k], torch.from_numpy(r["g64_" + k]))
            e_ref, e_pw = float(r["err_refseq32_" + k]), float(r["err_pairwise32_" + k])
            print("[parity] C5 init %s vs float64: HIP %.3e, reference op sequence in fp32 %.3e, pair-wise fp32 %.3e" % (k, e_hip, e_ref, e_pw))
            assert e_hip <= e_ref and e_hip < (5e-2 if k == "inducing_points" else 5e-3), (k, e_hip, e_ref, e_pw)


@pytest.mark.gpu
def test_ciq_engine_matches_the_reference_strategy_forward(dsvgp, gpu_device):
    """HIP CIQ path (Lanczos bounds, quadrature, msMINRES, NGD interpolation terms) against the reference's CIQ strategy forward
    text (tests/golden/strategy_ciq_*.npz, exact K^-1/2): within the quadrature / msMINRES tolerance of the reference (1e-4
    relative residual; 5e-3 on the moments)"""
    import glob, os
    import numpy as np
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_ciq_[0-9].npz"))):
        g = np.load(path)
        t = lambda k: torch.from_numpy(g[k]).float().to(gpu_device)
        inv_softplus = lambda v: float(np.log(np.expm1(float(v))))
        P = dict(inducing_points=t("Z"), inducing_directions=t("V"), natural_vec=t("natural_vec"), natural_mat=t("natural_mat"),
                 constant=torch.tensor([float(g["constant"])], device=gpu_device),
                 raw_outputscale=torch.tensor(inv_softplus(g["outputscale"]), device=gpu_device),
                 raw_lengthscale=torch.tensor([[inv_softplus(g["lengthscale"])]], device=gpu_device),
                 raw_noise=torch.tensor([0.0], device=gpu_device))
        eng = dsvgp.ElboEngine(gpu_device)
        eng.whitening = "ciq"
        mu, varn = eng.predict(P, t("x"), t("D"))
        noise = float(torch.nn.functional.softplus(torch.zeros(())) + 1e-4)
        e_mu, e_var = relmax(mu, torch.from_numpy(g["mean"])), relmax(varn.double().cpu() - noise, torch.from_numpy(g["variance"]))
        print("[parity] CIQ strategy vector %s: mean %.2e, variance %.2e" % (os.path.basename(path), e_mu, e_var))
        assert e_mu < 5e-3 and e_var < 5e-3


@pytest.mark.gpu
def test_ciq_engine_matches_the_reference_forward_run_with_msminres(dsvgp, gpu_device):
    """The same reference forward text with ``sqrt_inv_matmul`` = the oracle's quadrature + msMINRES in float64
    (tests/golden/strategy_ciq_minres_*.npz; within 6e-6 of the exact-root vectors): the HIP CIQ path (fp32 Lanczos bounds,
    fp32 msMINRES stopped at the same 1e-4 mean relative update) is held to 1e-3 of the max moment."""
    import glob, os
    import numpy as np
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_ciq_minres_[0-9].npz")))
    assert len(paths) >= 2
    for path in paths:
        g = np.load(path)
        t = lambda k: torch.from_numpy(g[k]).float().to(gpu_device)
        inv_softplus = lambda v: float(np.log(np.expm1(float(v))))
        P = dict(inducing_points=t("Z"), inducing_directions=t("V"), natural_vec=t("natural_vec"), natural_mat=t("natural_mat"),
                 constant=torch.tensor([float(g["constant"])], device=gpu_device),
                 raw_outputscale=torch.tensor(inv_softplus(g["outputscale"]), device=gpu_device),
                 raw_lengthscale=torch.tensor([[inv_softplus(g["lengthscale"])]], device=gpu_device),
                 raw_noise=torch.tensor([0.0], device=gpu_device))
        eng = dsvgp.ElboEngine(gpu_device)
        eng.whitening = "ciq"
        mu, varn = eng.predict(P, t("x"), t("D"))
        noise = float(torch.nn.functional.softplus(torch.zeros(())) + 1e-4)
        e_mu, e_var = relmax(mu, torch.from_numpy(g["mean"])), relmax(varn.double().cpu() - noise, torch.from_numpy(g["variance"]))
        print("[parity] CIQ reference forward + oracle msMINRES %s: mean %.2e, variance %.2e" % (os.path.basename(path), e_mu, e_var))
        assert e_mu < 1e-3 and e_var < 1e-3


@pytest.mark.gpu
def test_ciq_step_matches_autograd_through_the_reference_forward(dsvgp, gpu_device):
    """HIP CIQ step (quadrature + msMINRES forward and backward, NGD interpolation terms) against autograd through the reference's
    CIQ forward with the EXACT inverse square root.  Two bounds per vector: (i) against the oracle's own quadrature step in
    the same fp32 arithmetic, 5e-3 -- what the kernels add; (ii) against the exact-root reference, loss 1e-3 and gradients
    max(2e-2, 1.5 x the error the float64 oracle's quadrature has against the same reference) -- the method's own error:
    at M'=300 (strategy_ciq_grad_1) the Lanczos lower Ritz bound overestimates lambda_min and Q=15 quadrature is 2.6e-2 off
    the exact root in float64 already"""
    import glob, os
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_ciq_grad_*.npz"))):
        P, x, y, D, nd, loss_ref, g_ref = _ciq_grad_problem(path, torch.float32)
        eng = dsvgp.ElboEngine(gpu_device)
        eng.whitening = "ciq"
        Pg = {k: v.to(gpu_device) for k, v in P.items()}
        loss, grads, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
        errs = {"loss": abs(loss.item() - loss_ref) / abs(loss_ref)}
        errs.update({k: relmax(grads[k], g_ref[k]) for k in O.NGD_PARAM_NAMES})
        print("[parity] CIQ reference-forward gradient vector %s: %s" % (os.path.basename(path), ", ".join("%s %.1e" % kv for kv in errs.items())))
        loss32, g32, _, _ = O.ciq_loss_and_grads(P, x, y, D, nd)
        P64, x64, y64, D64, _, _, _ = _ciq_grad_problem(path, torch.float64)
        _, g64, _, _ = O.ciq_loss_and_grads(P64, x64, y64, D64, nd)
        e32 = {"loss": abs(loss.item() - loss32.item()) / abs(loss32.item())}
        e32.update({k: relmax(grads[k], g32[k]) for k in O.NGD_PARAM_NAMES})
        method = max(relmax(g64[k], g_ref[k]) for k in O.NGD_PARAM_NAMES)
        print("[parity]   ... against the fp32 oracle's quadrature step: %s; float64 quadrature vs exact root %.1e" % (
            ", ".join("%s %.1e" % kv for kv in e32.items()), method))
        assert e32["loss"] < 1e-5 and max(e32[k] for k in O.NGD_PARAM_NAMES) < 5e-3, e32
        assert errs["loss"] < 1e-3 and max(errs[k] for k in O.NGD_PARAM_NAMES) < max(2e-2, 1.5 * method), errs
