"""GPU parity tests, step level: the fused HIP ELBO forward/backward, prediction and the train_gp /
eval_gp drop-in against the CPU oracle (reference-mixed precision: fp32 model, fp64 Cholesky/solves).

Stated tolerance (fp32 path vs oracle): loss 2e-5 relative, predictive mean/variance 2e-4 of the max
magnitude, gradients 2e-3 relative in max-norm per parameter (the K_ZZ path amplifies fp32 kernel
rounding by cond(L) ~ 1e2-1e3; the oracle's own fp32-vs-fp64 spread is of the same size)."""
import math

import pytest
import torch

import dsvgp_oracle as O
from _golden import GRADIENT, PARAM_KEYS, STRATEGY, gradient_problem, strategy_problem

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _report(tag, errs):
    """measured GPU-vs-oracle errors go to the test output (``pytest -rP`` / the captured log): the margin under each
    stated tolerance stays visible and regressions show before they fail"""
    print("[parity] %s: %s" % (tag, ", ".join("%s %.2e" % (k, v) for k, v in errs.items())))


def make_problem(N, d, M, p, B, seed=0, perturb=True):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(N, d, generator=g)
    Y = O.testfun(X)
    Z = X[:M].clone()
    V = torch.eye(d)[:p].repeat(M, 1)
    if perturb:
        V = V + 0.1 * torch.randn(M * p, d, generator=g)
    P = O.init_params(Z, V, torch.float32, mean_init_std=0.2 if perturb else 1e-3, generator=g)
    Mp = M * (p + 1)
    if perturb:
        P["chol_variational_covar"] = torch.eye(Mp) + 0.05 * torch.randn(Mp, Mp, generator=g)   # garbage above diag is masked
        P["constant"] = torch.tensor([0.1])
        P["raw_outputscale"] = torch.tensor(0.2)
        P["raw_lengthscale"] = torch.tensor([[0.3]])
        P["raw_noise"] = torch.tensor([-0.5])
    cols = sorted([0] + (torch.randperm(d, generator=g)[:p] + 1).tolist())
    x = X[M:M + B].contiguous()
    y = Y[M:M + B][:, cols].reshape(-1).contiguous()
    D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1)
    return P, x, y, D, (d + 1) * N


def run_gpu(dsvgp, dev, P, x, y, D, nd, mll="ELBO", **kw):
    eng = dsvgp.ElboEngine(dev, **{k: v for k, v in kw.items() if k == "trsm_nb"})
    if "c_step" in kw:                  # (False: the Python-orchestrated path, whose intermediates live in eng._buf)
        eng.c_step = kw["c_step"]
    Pg = {k: v.to(dev) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(dev), y.to(dev), D.to(dev), nd, mll,
                                               **{k: v for k, v in kw.items() if k not in ("trsm_nb", "c_step")})
    torch.cuda.synchronize()
    return loss, grads, mu, varn, eng, Pg


CASES = [
    # N, d, M, p, B
    (400, 2, 20, 2, 200),      # reference tests/test_dsvgp.py sizes
    (600, 5, 40, 2, 128),      # scaled-down C2
    (500, 20, 30, 5, 96),      # C4 geometry (d=20, p=5), small M/B
    (300, 4, 25, 4, 50),       # p == d (full gradient)
    (300, 6, 70, 0, 64),       # p = 0: plain SVGP special case
    (400, 50, 20, 5, 40),      # C5 geometry (d = 50): packed rows of 56 floats, the KSM = 16 instances of the pair kernels
    (300, 7, 18, 3, 33),       # q = 4, odd sizes: the generic tiled kernels
]


GRAD_TOL_FP64 = 3e-4      # every gradient of the fp32 HIP step against the FLOAT64 oracle, max-norm relative per parameter
                          # (observed on MI355X, round 3: <= 1.1e-4 over the 7 geometries x 3 modes; the reference-precision
                          # oracle's own distance to float64 on the same states: <= 1.1e-4)


@pytest.mark.parametrize("N,d,M,p,B", CASES)
@pytest.mark.parametrize("mll", ["ELBO", "ELBO-general", "PLL"])
def test_step_matches_oracle(dsvgp, gpu_device, N, d, M, p, B, mll):
    """ELBO = Gram-matrix fast path (default), ELBO-general / PLL = per-output path."""
    fast = mll == "ELBO"
    mll = mll.split("-")[0]
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=N + d)
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, nd, mll)
    loss, grads, mu, varn, _, _ = run_gpu(dsvgp, gpu_device, P, x, y, D, nd, mll, fast=fast)
    assert abs(loss.item() - l_ref.item()) < 2e-5 * abs(l_ref.item()), (loss.item(), l_ref.item())
    assert relmax(mu, mu_ref) < 2e-4
    if fast:
        assert varn.numel() == 0
    else:
        assert relmax(varn, var_ref) < 2e-4
    P64 = {k: v.double() for k, v in P.items()}
    _, g64, _, _ = O.elbo_loss_and_grads(P64, x.double(), y.double(), D.double(), nd, mll)
    errs = {"loss": abs(loss.item() - l_ref.item()) / abs(l_ref.item()), "mu": relmax(mu, mu_ref)}
    for k in O.PARAM_NAMES:
        if p == 0 and k == "inducing_directions":
            continue
        gr = g_ref[k]
        if k == "chol_variational_covar":
            assert grads[k].triu(1).abs().max().item() == 0.0
        # ONE named oracle: the float64 oracle (the reference-precision oracle's own distance to it is printed beside it as the
        # yardstick of what an fp32 model can deliver at this state)
        e_64 = relmax(grads[k], g64[k])
        errs[k] = e_64
        errs[k + "(ref-precision oracle)"] = relmax(gr, g64[k])
        assert e_64 < GRAD_TOL_FP64, (k, e_64)
    _report("step N=%d d=%d M=%d p=%d B=%d %s%s" % (N, d, M, p, B, mll, "" if not fast else " fast"), errs)


def test_step_at_c2_sizes_against_oracle(dsvgp, gpu_device):
    """BASELINE config 2: d=5, M=200, p=2 (M'=600), B=512."""
    P, x, y, D, nd = make_problem(10000, 5, 200, 2, 512, seed=42)
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, nd)
    for nb, fast in ((128, False), (512, True), (1024, False), (1024, True)):
        loss, grads, mu, varn, _, _ = run_gpu(dsvgp, gpu_device, P, x, y, D, nd, trsm_nb=nb, fast=fast)
        assert abs(loss.item() - l_ref.item()) < 2e-5 * abs(l_ref.item())
        assert relmax(mu, mu_ref) < 2e-4 and (fast or relmax(varn, var_ref) < 2e-4)
        errs = {k: relmax(grads[k], g_ref[k]) for k in O.PARAM_NAMES}
        _report("C2 full size nb=%d%s" % (nb, " fast" if fast else ""), errs)
        for k in O.PARAM_NAMES:
            assert errs[k] < 3e-4, (nb, k, errs[k])               # observed <= 4.6e-5


def test_predict_matches_oracle_and_initial_variance(dsvgp, gpu_device):
    P, x, y, D, nd = make_problem(500, 5, 40, 2, 100, seed=9)
    mu_ref, var_ref = O.predictive(P, x, D)
    _, _, noise = O.constrained(P)
    eng = dsvgp.ElboEngine(gpu_device)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    mu, varn = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu, mu_ref) < 2e-4 and relmax(varn, var_ref + noise) < 2e-4
    # q(u) = N(0, I): variance is exactly prior diag + 1e-4 + noise (size-independent identity)
    P0, x0, _, D0, _ = make_problem(500, 5, 40, 2, 100, seed=9, perturb=False)
    P0["variational_mean"].zero_()
    Pg0 = {k: v.to(gpu_device) for k, v in P0.items()}
    mu0, varn0 = eng.predict(Pg0, x0.to(gpu_device), D0.to(gpu_device))
    ell, s, noise0 = O.constrained(P0)
    expect = s * O.kernel_diag(100, 2, ell) + 1e-4 + noise0
    assert relmax(varn0, expect) < 1e-5 and mu0.abs().max().item() < 1e-6


def test_not_psd_raises_like_reference(dsvgp, gpu_device):
    P, x, y, D, nd = make_problem(200, 3, 10, 1, 20, seed=1)
    P["raw_outputscale"] = torch.tensor(float("nan"))
    with pytest.raises(dsvgp.NotPSDError):
        run_gpu(dsvgp, gpu_device, P, x, y, D, nd)


def test_full_size_properties_c4(dsvgp, gpu_device):
    """BASELINE config 4 geometry at full size on one GPU (d=20, M=500, p=5 -> M'=3000; B=4096 -> B'=24576):
    size-independent identities instead of an oracle run."""
    torch.manual_seed(0)
    dev = gpu_device
    N, d, M, p, B = 20000, 20, 500, 5, 4096
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=4, perturb=False)
    loss_f, grads_f, _, _, _, _ = run_gpu(dsvgp, dev, P, x, y, D, nd, fast=True)
    loss_c, grads_c, mu_c, varn_c, eng_c, _ = run_gpu(dsvgp, dev, P, x, y, D, nd, fast=False)      # one C call (dsvgp_elbo_step_po_f32)
    assert eng_c.c_step_used
    del eng_c
    torch.cuda.empty_cache()
    # (the Python-orchestrated per-output path: its intermediates L, A64, K_ZX are engine buffers the identities below read)
    loss, grads, mu, varn, eng, Pg = run_gpu(dsvgp, dev, P, x, y, D, nd, fast=False, c_step=False)
    assert math.isfinite(loss.item()) and not eng.c_step_used
    assert abs(loss_c.item() - loss.item()) < 4e-6 * abs(loss.item()) and relmax(varn_c, varn) < 4e-6 and relmax(mu_c, mu) < 4e-6
    for k in grads:
        assert relmax(grads_c[k], grads[k]) < 5e-5, k
    # (0) the ELBO fast path (Gram formulation) and the per-output path agree at full size
    assert abs(loss_f.item() - loss.item()) < 2e-5 * abs(loss.item())
    for k in grads:
        assert relmax(grads_f[k], grads[k]) < 5e-3, k
    ell, s, noise = O.constrained(P)
    # (1) L L^T == K_ZZ + 1e-3 I  and  L A == K_ZX  (residuals of the fp64 factor / panel solve)
    L = eng._buf["L"].tril()
    ops = dsvgp._ops
    ctx = ops.Context.get(dev)
    hyp = ops.hyp_forward(ctx, Pg["raw_lengthscale"], Pg["raw_outputscale"], Pg["raw_noise"])
    pz = ops.pack_points(ctx, Pg["inducing_points"], Pg["inducing_directions"], p, hyp, eng.center)
    Kzz = ops.kernel_fwd(ctx, pz, M, pz, M, d, p, hyp, jitter=1e-3, dtype=torch.float64)
    assert ((L @ L.t()).tril() - Kzz.tril()).abs().max().item() < 1e-10
    assert (Kzz - Kzz.t()).abs().max().item() < 1e-5
    Kzx = eng._buf["Kzx"].double()
    assert ((L @ eng._buf["A64"]) - Kzx).abs().max().item() < 1e-9
    # (2) with L_S = I: W == A, variance == prior diag + 1e-4 + noise
    expect = (s * O.kernel_diag(B, p, ell) + 1e-4 + noise).float()
    assert relmax(varn, expect) < 1e-5
    # (3) K_ZX == K_XZ^T (the redundant reference assembly)
    px = ops.pack_points(ctx, x.to(dev), D.to(dev), p, hyp, eng.center)
    Kxz = ops.kernel_fwd(ctx, px, B, pz, M, d, p, hyp)
    assert (Kxz.t() - eng._buf["Kzx"]).abs().max().item() < 1e-5
    # (4) linearity of the data-parallel split: two half batches, KL once, sum == full batch
    for fast in (False, True):
        _check_shard_additivity(eng, Pg, x, y, D, nd, B, p, dev, loss, grads, fast)


def _check_shard_additivity(eng, Pg, x, y, D, nd, B, p, dev, loss, grads, fast):
    h = B // 2
    q = p + 1
    l1, g1, _, _ = eng.loss_and_grads(Pg, x[:h].to(dev), y[:h * q].to(dev), D[:h * p].to(dev), nd, global_rows=B * q, include_kl=True, fast=fast)
    g1 = {k: v.clone() for k, v in g1.items()}
    l1 = l1.clone()
    l2, g2, _, _ = eng.loss_and_grads(Pg, x[h:].to(dev), y[h * q:].to(dev), D[h * p:].to(dev), nd, global_rows=B * q, include_kl=False, fast=fast)
    assert abs((l1 + l2).item() - loss.item()) < 1e-4 * abs(loss.item())
    for k in grads:
        assert relmax(g1[k] + g2[k], grads[k]) < 5e-3, k


def test_train_gp_eval_gp_drop_in(dsvgp, gpu_device, capsys):
    """reference tests/test_dsvgp.py (n=600, d=2, M=20, p=2, B=200) shortened: loss goes down,
    predictions have the reference's shapes/ordering and a sane error."""
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, n_test = 600, 2, 300
    train_x, test_x = torch.rand(n, dim), torch.rand(n_test, dim)
    train_y, test_y = O.testfun(train_x), O.testfun(test_x)
    ds, dst = TensorDataset(train_x, train_y), TensorDataset(test_x, test_y)
    model, likelihood = dsvgp.train_gp(ds, num_inducing=20, num_directions=2, minibatch_size=200, minibatch_dim=2,
                                       num_epochs=150, learning_rate_hypers=0.01, learning_rate_ngd=0.1,
                                       inducing_data_initialization=False, use_ngd=False, use_ciq=False,
                                       lr_sched=None, num_contour_quadrature=15, tqdm=False, verbose=True, seed=0)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 5 and losses[-1] < losses[0] - 0.5
    keys = set(model.state_dict().keys())
    for k in ["variational_strategy.inducing_points", "variational_strategy.inducing_directions",
              "variational_strategy._variational_distribution.variational_mean",
              "variational_strategy._variational_distribution.chol_variational_covar",
              "variational_strategy.updated_strategy", "variational_strategy.variational_params_initialized",
              "mean_module.constant", "covar_module.raw_outputscale", "covar_module.base_kernel.raw_lengthscale"]:
        assert k in keys, k
    assert "noise_covar.raw_noise" in likelihood.state_dict()
    means, variances = dsvgp.eval_gp(dst, model, likelihood, num_directions=2, minibatch_size=128, minibatch_dim=2)
    assert means.shape == (n_test * 3,) and variances.shape == (n_test * 3,) and not means.is_cuda
    assert (variances > 0).all()
    mse = ((means[::3] - test_y[:, 0]) ** 2).mean().item()
    assert mse < 0.25, mse          # variance of sin(2 pi r^2) on the unit square is ~0.5
    # eval path == oracle predictive on the trained parameters
    P = {k: v.detach().cpu() for k, v in model._param_dict(likelihood).items()}
    D = torch.eye(dim)[:2].repeat(n_test, 1)
    mu_ref, var_ref = O.predictive(P, test_x, D)
    _, _, noise = O.constrained(P)
    P64 = {k: v.double() for k, v in P.items()}
    mu64, var64 = O.predictive(P64, test_x.double(), D.double())
    # trained hyper-parameters give a worse conditioned K_ZZ; ONE named oracle (float64), the reference-precision oracle's own
    # distance to it printed beside the HIP path's
    errs = {"mean": relmax(means, mu64), "mean(ref-precision oracle)": relmax(mu_ref, mu64),
            "variance": relmax(variances, var64 + noise.double()), "variance(ref-precision oracle)": relmax(var_ref + noise, var64 + noise.double())}
    _report("train_gp / eval_gp drop-in, trained state", errs)
    assert errs["mean"] < 5e-4 and errs["variance"] < 5e-4, errs          # observed 1.3e-4 / 6.6e-5 (ref-precision oracle: 1.2e-4 / 8.2e-5)


def test_grad_svgp_drop_in(dsvgp, gpu_device, capsys):
    """full-gradient SVGP (reference grad_svgp.py / GradVariationalStrategy.py) == the DSVGP
    kernels with p = d and canonical directions; num_data = n_samples; reference tests/test_grad_svgp.py sizes."""
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, n_test = 600, 2, 200
    train_x, test_x = torch.rand(n, dim), torch.rand(n_test, dim)
    train_y, test_y = O.testfun(train_x), O.testfun(test_x)
    G = dsvgp.grad_svgp
    model, likelihood = G.train_gp(TensorDataset(train_x, train_y), dim, num_inducing=20, minibatch_size=200,
                                   num_epochs=100, mll_type="PLL", tqdm=False, seed=1)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 5 and losses[-1] < losses[0]
    assert "variational_strategy.inducing_directions" not in model.state_dict()
    means, variances = G.eval_gp(TensorDataset(test_x, test_y), model, likelihood, minibatch_size=64)
    assert means.shape == (n_test * 3,) and (variances > 0).all()
    # one ELBO step of the trained model against the oracle with p = d, V = I, D = I, num_data = n
    P = {k: v.detach().cpu() for k, v in model._param_dict(likelihood).items()}
    x, y = train_x[:64], train_y[:64].reshape(-1)
    D = torch.eye(dim).repeat(64, 1)
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, n)
    eng = dsvgp.ElboEngine(gpu_device)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), n, fast=False)
    # trained hyper-parameters (short lengthscale) -> worse conditioned K_ZZ: compare with the mixed AND the fp64
    # oracle; their own spread is ~1e-4 on the loss here
    P64 = {k: v.double() for k, v in P.items()}
    l64, g64, _, _ = O.elbo_loss_and_grads(P64, x.double(), y.double(), D.double(), n)
    errs = {"loss": abs(loss.item() - l64.item()) / abs(l64.item()), "loss(ref-precision oracle)": abs(l_ref.item() - l64.item()) / abs(l64.item())}
    assert errs["loss"] < 1e-4, errs                                                   # observed 2.9e-6 ... 2.6e-5 (the trained state differs run to run; ref-precision oracle: 6e-6 ... 8e-6)
    assert relmax(mu, mu_ref) < 2e-3 and relmax(varn, var_ref) < 2e-3
    for k in ("inducing_points", "variational_mean", "chol_variational_covar", "raw_lengthscale", "raw_noise"):
        errs[k] = relmax(grads[k], g64[k])                        # ONE named oracle: float64
        errs[k + "(ref-precision oracle)"] = relmax(g_ref[k], g64[k])
    _report("grad_svgp drop-in, trained state", errs)
    for k in ("inducing_points", "variational_mean", "chol_variational_covar", "raw_lengthscale", "raw_noise"):
        assert errs[k] < 4e-3, (k, errs[k])                                            # observed <= 1.2e-3 (ref-precision oracle: 1.0e-3)
    mu_e, var_e = O.predictive(P, test_x, torch.eye(dim).repeat(n_test, 1))
    _, _, noise = O.constrained(P)
    assert relmax(means, mu_e) < 2e-3 and relmax(variances, var_e + noise) < 2e-3


def test_c4_full_size_step_against_committed_oracle_vector(dsvgp, gpu_device):
    """BASELINE config 4 at FULL size (M'=3000, B'=24576): loss, predictive head and all gradients against the
    oracle run committed as tests/golden/c4_step.npz (oracle/make_c4_fixture.py; inputs regenerated from the seed)."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from make_c4_fixture import make_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4_step.npz"))
    P, x, y, D, nd = make_inputs()
    for fast in (True, False):
        loss, grads, mu, varn, _, _ = run_gpu(dsvgp, gpu_device, P, x, y, D, nd, trsm_nb=4096, fast=fast)
        # stated tolerances of the headline configuration (fp32 model against the fp64-solve oracle): loss 5e-6, predictive head 5e-5,
        # every gradient 1e-4 -- observed on MI355X (round 2): loss 3e-7, mu 1.6e-6, gradients <= 3e-6 (the [parity] line below)
        assert abs(loss.item() - float(g["loss"])) < 5e-6 * abs(float(g["loss"])), (fast, loss.item(), float(g["loss"]))
        assert relmax(mu[:256], torch.from_numpy(g["mu_head"])) < 5e-5
        if not fast:
            assert relmax(varn[:256], torch.from_numpy(g["varn_head"])) < 5e-5
        for k in O.PARAM_NAMES:
            if k == "chol_variational_covar":
                gl = grads[k]
                assert abs(gl.double().norm().item() - float(g["g_LS_norm"])) < 1e-4 * float(g["g_LS_norm"])
                assert relmax(gl[:96, :96], torch.from_numpy(g["g_LS_block"])) < 1e-4
                assert relmax(torch.diagonal(gl), torch.from_numpy(g["g_LS_diag"])) < 1e-4
                assert relmax(gl[-8:, :], torch.from_numpy(g["g_LS_lastrows"])) < 1e-4
            else:
                assert relmax(grads[k], torch.from_numpy(g["g_" + k])) < 1e-4, (fast, k, relmax(grads[k], torch.from_numpy(g["g_" + k])))
        errs = {"loss": abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), "mu": relmax(mu[:256], torch.from_numpy(g["mu_head"]))}
        errs.update({"g_" + k: relmax(grads[k], torch.from_numpy(g["g_" + k])) for k in O.PARAM_NAMES if k != "chol_variational_covar"})
        errs["g_LS_block"] = relmax(grads["chol_variational_covar"][:96, :96], torch.from_numpy(g["g_LS_block"]))
        _report("C4 ELBO %s" % ("fast" if fast else "per-output"), errs)


# ------------------------------------------------------------------ derivative-free data
@pytest.mark.parametrize("mll", ["ELBO", "ELBO-general", "PLL"])
def test_dfree_step_matches_oracle(dsvgp, gpu_device, mll):
    """reference DFreeDirectionalGradVariationalStrategy.py:113-136: inducing derivatives, value-only data."""
    fast = mll == "ELBO"
    mll = mll.split("-")[0]
    N, d, M, p, B = 400, 3, 16, 2, 96
    P, x, _, D, nd = make_problem(N, d, M, p, B, seed=11)
    y = O.testfun(x)[:, 0].contiguous()
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, nd, mll, data_outputs="values")
    eng = dsvgp.ElboEngine(gpu_device)
    eng.data_outputs = "values"
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll, fast=fast)
    assert mu.shape == (B,) and abs(loss.item() - l_ref.item()) < 2e-5 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 2e-4
    if not fast:
        assert relmax(varn, var_ref) < 2e-4
    for k in O.PARAM_NAMES:
        assert relmax(grads[k], g_ref[k]) < 2e-3, k
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 2e-4 and relmax(varn2, var_ref) < 2e-4
    with pytest.raises(ValueError):          # interleaved (p+1)-wide targets are the DSVGP layout, not this one
        eng.loss_and_grads(Pg, x.to(gpu_device), torch.zeros(B * (p + 1), device=gpu_device), D.to(gpu_device), nd)


def test_dfree_train_gp_drop_in(dsvgp, gpu_device, capsys):
    """reference tests/test_dfree_dsvgp.py: n=600, d=2, 20 inducing points with 2 directions, scalar targets."""
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, p = 600, 2, 2
    train_x, test_x = torch.rand(n, dim), torch.rand(200, dim)
    train_y, test_y = O.testfun(train_x)[:, 0].contiguous(), O.testfun(test_x)[:, 0].contiguous()
    F = dsvgp.dfree_directional_vi
    model, likelihood = F.train_gp(TensorDataset(train_x, train_y), num_inducing=20, num_directions=p,
                                   minibatch_size=200, minibatch_dim=p, num_epochs=120, learning_rate_hypers=0.01,
                                   inducing_data_initialization=False, tqdm=False, seed=5)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 5 and losses[-1] < losses[0]
    means, variances = F.eval_gp(TensorDataset(test_x, test_y), model, likelihood, num_directions=p, minibatch_size=100,
                                 minibatch_dim=p)
    assert means.shape == (200,) and (variances > 0).all()
    mse_model = ((means - test_y) ** 2).mean().item()
    mse_const = ((test_y.mean() - test_y) ** 2).mean().item()
    assert mse_model < 0.5 * mse_const                    # learned something about f from values alone
    assert type(model.variational_strategy).__module__.endswith("DFreeDirectionalGradVariationalStrategy")


# ------------------------------------------------------------------ shared inducing directions
@pytest.mark.parametrize("mll", ["ELBO", "PLL"])
def test_shared_directions_step_matches_oracle(dsvgp, gpu_device, mll):
    """reference SharedDirectionalGradVariationalStrategy.py:95-107,210-212."""
    N, d, M, p, B = 400, 4, 14, 2, 80
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=21)
    g = torch.Generator().manual_seed(4)
    P["inducing_directions"] = torch.eye(d)[:p] + 0.2 * torch.randn(p, d, generator=g)        # ONE shared set
    P["variational_mean"] = 0.3 * torch.randn(M + p, generator=g)
    P["chol_variational_covar"] = torch.eye(M + p) + 0.05 * torch.randn(M + p, M + p, generator=g)
    l_ref, g_ref, mu_ref, var_ref = O.shared_loss_and_grads(P, x, y, D, nd, mll)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.shared_directions = True
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll)
    assert abs(loss.item() - l_ref.item()) < 2e-5 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 2e-4 and relmax(varn, var_ref) < 2e-4
    for k in O.PARAM_NAMES:
        assert grads[k].shape == g_ref[k].shape and relmax(grads[k], g_ref[k]) < 2e-3, k
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 2e-4 and relmax(varn2, var_ref) < 2e-4


def test_shared_train_gp_drop_in(dsvgp, gpu_device, capsys):
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, p = 600, 2, 2
    train_x = torch.rand(n, dim)
    train_y = O.testfun(train_x)
    S = dsvgp.shared_directional_vi
    model, likelihood = S.train_gp(TensorDataset(train_x, train_y), num_inducing=20, num_directions=p,
                                   minibatch_size=200, minibatch_dim=p, num_epochs=80,
                                   inducing_data_initialization=False, tqdm=False, seed=2)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 3 and losses[-1] < losses[0]
    sd = model.state_dict()
    assert sd["variational_strategy.inducing_directions"].shape == (p, dim)
    assert sd["variational_strategy._variational_distribution.variational_mean"].shape == (20 + p,)
    means, variances = S.eval_gp(TensorDataset(train_x[:50], train_y[:50]), model, likelihood, num_directions=p,
                                 minibatch_size=25, minibatch_dim=p)
    assert means.shape == (150,) and (variances > 0).all()
    with pytest.raises(AssertionError):       # the reference tiles the directions in this branch and its forward assertion fails
        S.train_gp(TensorDataset(train_x, train_y), num_inducing=20, num_directions=p, minibatch_size=200,
                   minibatch_dim=p, num_epochs=1, inducing_data_initialization=True, tqdm=False, verbose=False)


# ------------------------------------------------------------------ plain SVGP harness (BASELINE config 0)
def test_traditional_vi_drop_in(dsvgp, gpu_device, capsys):
    """reference tests/test_traditional_vi.py: SVGP on a 1-D sine, 50 inducing points (the p = 0 path of the engine)."""
    import math as _m
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n = 500
    train_x = torch.rand(n, 1)
    train_y = torch.sin(2 * _m.pi * train_x[:, 0]) + 0.05 * torch.randn(n)
    test_x = torch.linspace(0.02, 0.98, 120).reshape(-1, 1)
    test_y = torch.sin(2 * _m.pi * test_x[:, 0])
    T = dsvgp.traditional_vi
    model, likelihood = T.train_gp(TensorDataset(train_x, train_y), 1, num_inducing=50, minibatch_size=100, num_epochs=150,
                                   learning_rate_hypers=0.02, tqdm=False, seed=4)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert "Using ELBO" in out and len(losses) >= 5 and losses[-1] < losses[0]
    sd = model.state_dict()
    assert sd["variational_strategy.inducing_points"].shape == (50, 1)
    assert sd["variational_strategy._variational_distribution.chol_variational_covar"].shape == (50, 50)
    means, variances = T.eval_gp(TensorDataset(test_x, test_y), model, likelihood, minibatch_size=60)
    assert means.shape == (120,) and (variances > 0).all()
    assert ((means - test_y) ** 2).mean().item() < 0.05
    # one step of the trained model against the oracle with no directions at all
    P = {k: v.detach().cpu() for k, v in model._param_dict(likelihood).items()}
    x, y = train_x[:64], train_y[:64].contiguous()
    D = torch.empty(0, 1)
    l_ref, g_ref, mu_ref, var_ref = O.elbo_loss_and_grads(P, x, y, D, n)
    eng = dsvgp.ElboEngine(gpu_device)
    loss, grads, mu, varn = eng.loss_and_grads({k: v.to(gpu_device) for k, v in P.items()}, x.to(gpu_device),
                                               y.to(gpu_device), D.to(gpu_device), n, fast=False)
    assert abs(loss.item() - l_ref.item()) < 1e-4 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 1e-3 and relmax(varn, var_ref) < 1e-3
    # natural-gradient variant of the same harness
    model2, lik2 = T.train_gp(TensorDataset(train_x, train_y), 1, num_inducing=30, minibatch_size=100, num_epochs=40,
                              use_ngd=True, learning_rate_ngd=0.1, tqdm=False, seed=4, verbose=False)
    m2, v2 = T.eval_gp(TensorDataset(test_x, test_y), model2, lik2, minibatch_size=60)
    assert torch.isfinite(m2).all() and (v2 > 0).all()
    assert "variational_strategy._variational_distribution.natural_vec" in model2.state_dict()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["all", "values"])
def test_joint_predictive_covariance_matches_oracle(dsvgp, gpu_device, mode):
    """Full covariance of likelihood(model(x)) (reference DGVS.py:199-208): HIP assembly + Gram products vs the oracle."""
    P, x, y, D, nd = make_problem(500, 5, 40, 2, 60, seed=4)
    mu_ref, Sig_ref = O.predictive_joint({k: v.double() for k, v in P.items()}, x.double(), D.double(), data_outputs=mode)
    _, _, noise = O.constrained(P)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.data_outputs = mode
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    mu, Sigma = eng.predict_joint(Pg, x.to(gpu_device), D.to(gpu_device))
    n = 60 * (3 if mode == "all" else 1)
    assert Sigma.shape == (n, n) and Sigma.dtype == torch.float32
    Sig_ref = Sig_ref + noise.double() * torch.eye(n, dtype=torch.float64)
    assert relmax(mu, mu_ref) < 5e-4 and relmax(Sigma, Sig_ref) < 5e-4
    assert (Sigma - Sigma.t()).abs().max().item() < 1e-5 * Sigma.abs().max().item()
    # its diagonal is what predict() returns
    _, varn = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(Sigma.diagonal(), varn) < 1e-4
    # Cholesky root reproduces Sigma; sampling through it is mean + root eps
    R = torch.tril(eng.covariance_root(Sigma))
    assert relmax(R @ R.t(), Sig_ref) < 5e-4
    eps = torch.randn(7, n, device=gpu_device)
    draws = eng.draw(mu, eng.covariance_root(Sigma), eps)
    assert relmax(draws, mu.double().cpu() + eps.double().cpu() @ R.cpu().t()) < 1e-5


@pytest.mark.gpu
def test_distribution_sample_protocol_of_bo_drivers(dsvgp, gpu_device):
    """``likelihood(model(x, derivative_directions=D)).sample(torch.Size([n]))[:, ::p+1]`` as the reference's BO drivers
    call it (experiments/GNN_bo/gcn_turbo.py:238-239): shapes, moments of the draws, base_samples determinism."""
    torch.manual_seed(0)
    dim, p, M, nx = 3, 2, 12, 20
    Z = torch.rand(M, dim)
    model = dsvgp.GPModel(Z, torch.eye(dim)[:p].repeat(M, 1), dim).to(gpu_device)
    likelihood = dsvgp.gp_shim.GaussianLikelihood().to(gpu_device)
    model.eval()
    likelihood.eval()
    x = torch.rand(nx, dim, device=gpu_device)
    D = torch.eye(dim)[:p].repeat(nx, 1).to(gpu_device)
    with torch.no_grad():
        preds = likelihood(model(x, derivative_directions=D))
        S = preds.sample(torch.Size([4000]))
        assert S.shape == (4000, nx * (p + 1)) and S[:, ::model.num_directions + 1].shape == (4000, nx)
        assert preds.sample().shape == (nx * (p + 1),)
        Sigma = preds.covariance_matrix
        assert relmax(Sigma.diagonal(), preds.variance) < 1e-4
        emp = torch.cov(S.t().double())
        assert (emp.cpu() - Sigma.double().cpu()).abs().max().item() < 0.12 * Sigma.abs().max().item()
        assert (S.mean(0) - preds.mean).abs().max().item() < 0.1 * Sigma.diagonal().max().sqrt().item()
        eps = torch.randn(5, nx * (p + 1), device=gpu_device)
        a, b = preds.rsample(torch.Size([5]), base_samples=eps), preds.rsample(torch.Size([5]), base_samples=eps)
        assert torch.equal(a, b)
        lo, hi = preds.confidence_region()
        assert torch.allclose(hi - lo, 4 * preds.stddev)
        # q(f) without the likelihood: noise removed from the diagonal only
        Sf = model(x, derivative_directions=D).covariance_matrix
        noise = likelihood.noise.reshape(())
        assert relmax(Sigma - Sf, noise * torch.eye(nx * (p + 1), device=gpu_device)) < 1e-3


def test_direct_gradient_step_equals_autograd_protocol(dsvgp, gpu_device):
    """TrainLoop's default step (mll.backward_step: engine gradients become .grad) against the reference's three lines
    ``loss = -mll(output, y); loss.backward()`` through torch.autograd: same losses and parameters (up to the run-to-run
    summation order of the split-K atomics)."""
    torch.manual_seed(0)
    n, dim = 400, 3
    X = torch.rand(n, dim, device=gpu_device)
    Y = O.testfun(X.cpu()).to(gpu_device)
    states = []
    for protocol in (False, True):
        torch.manual_seed(1)                      # the 1e-3 randn initialisation of the variational mean
        loop = dsvgp.setup_training(None, num_inducing=16, num_directions=2, minibatch_size=100, minibatch_dim=2,
                                    num_epochs=1, learning_rate_hypers=0.01, seed=3, tensors=(X, Y))
        loop.autograd_protocol = protocol
        loop.col_rng.seed(5)
        losses = []
        perm = loop.epoch_permutation()
        for k in range(4):
            loss, _, _ = loop.step(perm[k * 100:(k + 1) * 100])
            losses.append(loss.item())
        states.append((losses, {k: v.detach().clone() for k, v in loop.model._param_dict(loop.likelihood).items()}))
    for a, b in zip(*[st[0] for st in states]):
        assert abs(a - b) < 1e-5 * abs(a), (a, b)
    for k in states[0][1]:
        assert (states[0][1][k] - states[1][1][k]).abs().max().item() < 2e-4, k      # 4 Adam steps of 1e-2


def test_panel_regime_beyond_the_explicit_inverse_limit(dsvgp, gpu_device):
    """M' = 9000 > 8192: the engine switches by itself to 512-wide panels (potrf + inverted diagonal blocks + panel solves,
    general ELBO schedule on fp64 A).  Its step must agree with the explicit-inverse regime forced on the same inputs --
    two independent code paths through the factorisation, the solves and the Cholesky backward."""
    N, d, M, p, B = 4000, 20, 1500, 5, 256
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=12)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xg, yg, Dg = x.to(gpu_device), y.to(gpu_device), D.to(gpu_device)
    out = {}
    for name, nb in (("panel", None), ("explicit", 16384)):
        eng = dsvgp.ElboEngine(gpu_device, trsm_nb=nb)
        loss, grads, mu, _ = eng.loss_and_grads(Pg, xg, yg, Dg, nd)
        torch.cuda.synchronize()
        assert eng.trsm_nb == (512 if nb is None else nb)
        out[name] = (loss.item(), {k: v.clone() for k, v in grads.items()}, mu.clone())
        del eng
        torch.cuda.empty_cache()
    la, ga, ma = out["panel"]
    lb, gb, mb = out["explicit"]
    assert math.isfinite(la) and abs(la - lb) < 2e-5 * abs(lb), (la, lb)
    assert relmax(ma, mb) < 2e-4
    for k in ga:
        assert relmax(ga[k], gb[k]) < 5e-3, (k, relmax(ga[k], gb[k]))


@pytest.mark.parametrize("mll", ["ELBO", "PLL"])
def test_ten_step_trajectory_matches_oracle_training(dsvgp, gpu_device, mll):
    """Ten optimisation steps (engine gradients + the two fused Adam optimisers, fresh minibatch every step) against the same
    loop on the CPU oracle with torch.optim.Adam: the parameter trajectories stay together (losses to 1e-4, parameters to 2e-3
    of their range after 10 steps of lr = 0.01)."""
    N, d, M, p, B = 600, 4, 24, 2, 64
    P, _, _, _, nd = make_problem(N, d, M, p, B, seed=31)
    g = torch.Generator().manual_seed(31)
    X = torch.rand(N, d, generator=g)
    Y = O.testfun(X)
    names = list(O.PARAM_NAMES)
    var_names = ("variational_mean", "chol_variational_covar")
    Pc = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Pd = {k: torch.nn.Parameter(v.clone().to(gpu_device)) for k, v in P.items()}
    oc = [torch.optim.Adam([Pc[k] for k in var_names], lr=0.01),
          torch.optim.Adam([Pc[k] for k in names if k not in var_names], lr=0.01)]
    od = [dsvgp.FusedAdam([Pd[k] for k in var_names], lr=0.01),
          dsvgp.FusedAdam([Pd[k] for k in names if k not in var_names], lr=0.01)]
    eng = dsvgp.ElboEngine(gpu_device)
    for step in range(10):
        idx = torch.randperm(N, generator=g)[:B]
        cols = sorted([0] + (torch.randperm(d, generator=g)[:p] + 1).tolist())
        x, y = X[idx].contiguous(), Y[idx][:, cols].reshape(-1).contiguous()
        D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1)
        l_ref, g_ref, _, _ = O.elbo_loss_and_grads({k: v.detach() for k, v in Pc.items()}, x, y, D, nd, mll)
        for k in names:
            Pc[k].grad = g_ref[k].clone()
        for o in oc:
            o.step()
        loss, grads, _, _ = eng.loss_and_grads({k: v.detach() for k, v in Pd.items()}, x.to(gpu_device), y.to(gpu_device),
                                               D.to(gpu_device), nd, mll)
        assert abs(loss.item() - l_ref.item()) < 1e-4 * abs(l_ref.item()), (step, loss.item(), l_ref.item())
        for k in names:
            Pd[k].grad = grads[k]
        for o in od:
            o.step()
    for k in names:
        ref = Pc[k].detach()
        if k == "chol_variational_covar":
            ref = torch.tril(ref)                                    # the strict upper part never receives a gradient
            got = torch.tril(Pd[k].detach().cpu())
        else:
            got = Pd[k].detach().cpu()
        scale = max(ref.abs().max().item(), 1e-2)
        assert (got - ref).abs().max().item() < 2e-3 * scale, (k, (got - ref).abs().max().item(), scale)


def test_two_engines_of_equal_size_keep_their_own_cholesky_scratch(dsvgp, gpu_device):
    """Two models with the same M' (BO drivers), eval-mode Cholesky cache, and a LARGER batch after the other model has
    factored its own K_ZZ: the re-allocated solve workspace is re-seeded from THIS engine's inverted diagonal blocks, never
    from another factor of the same size (the potrf scratch is owned per engine and per factor)."""
    d, M, p = 4, 50, 2                                  # M' = 150: three 64-blocks
    out = []
    for nb in (64, None):                               # panel regime (trtri from the 64 x 64 seeds) and explicit inverse
        PA, xa, _, Da, _ = make_problem(400, d, M, p, 40, seed=71)
        PB, xb, _, Db, _ = make_problem(400, d, M, p, 40, seed=72)
        PB["raw_lengthscale"] = torch.tensor([[-0.4]])
        ea, eb = dsvgp.ElboEngine(gpu_device, trsm_nb=nb), dsvgp.ElboEngine(gpu_device, trsm_nb=nb)
        ga = {k: v.to(gpu_device) for k, v in PA.items()}
        gb = {k: v.to(gpu_device) for k, v in PB.items()}
        g = torch.Generator().manual_seed(3)
        xbig = torch.rand(300, d, generator=g)
        Dbig = torch.eye(d)[:p].repeat(300, 1)
        ea.predict(ga, xa.to(gpu_device), Da.to(gpu_device), cache=True)          # A factors, small batch
        eb.predict(gb, xb.to(gpu_device), Db.to(gpu_device), cache=True)          # B factors a different K_ZZ of the same size
        eb.covariance_root(torch.eye(M * (p + 1), device=gpu_device) * 2.0)       # ... and another potrf of size M'
        mu, varn = ea.predict(ga, xbig.to(gpu_device), Dbig.to(gpu_device), cache=True)   # A: cache hit, larger batch
        mu_ref, var_ref = O.predictive(PA, xbig, Dbig)
        _, _, noise = O.constrained(PA)
        assert relmax(mu, mu_ref) < 2e-4 and relmax(varn, var_ref + noise) < 2e-4, nb
        mu2, varn2 = eb.predict(gb, xbig.to(gpu_device), Dbig.to(gpu_device), cache=True)
        mu2_ref, var2_ref = O.predictive(PB, xbig, Dbig)
        _, _, noise2 = O.constrained(PB)
        assert relmax(mu2, mu2_ref) < 5e-4 and relmax(varn2, var2_ref + noise2) < 5e-4, nb


@pytest.mark.parametrize("mll", ["ELBO", "PLL"])
def test_c3_full_size_step_against_committed_oracle_vector(dsvgp, gpu_device, mll):
    """BASELINE config 3 at FULL size: full-gradient SVGP d=10, M=300 -> M' = 3300, B=512 -> B' = 5632 (reference
    GradVariationalStrategy.py:87-137, grad_svgp.py:119,143) against the oracle run committed as tests/golden/c3_step.npz
    (oracle/make_c3_fixture.py; inputs regenerated from the seed).  Tolerances: loss 2e-5, mean / variance 2e-4,
    gradients 5e-3 of the max magnitude per parameter (fp32 model, fp64 solves, M' = 3300)."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from make_c3_fixture import make_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c3_step.npz"))
    P, x, y, D, nd = make_inputs()
    pre = mll + "_"
    for fast in ((True, False) if mll == "ELBO" else (False,)):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.chol_jitter = 1e-8                              # GradVariationalStrategy.py:72 (fp64 psd_safe_cholesky default)
        Pg = {k: v.to(gpu_device) for k, v in P.items()}
        loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll, fast=fast)
        torch.cuda.synchronize()
        errs = {"loss": abs(loss.item() - float(g[pre + "loss"])) / abs(float(g[pre + "loss"])),
                "mu": relmax(mu[:256], torch.from_numpy(g[pre + "mu_head"]))}
        if not fast:
            errs["var"] = relmax(varn[:256], torch.from_numpy(g[pre + "varn_head"]))
        gl = grads["chol_variational_covar"]
        errs["g_LS_norm"] = abs(gl.double().norm().item() - float(g[pre + "g_LS_norm"])) / float(g[pre + "g_LS_norm"])
        errs["g_LS_block"] = relmax(gl[:96, :96], torch.from_numpy(g[pre + "g_LS_block"]))
        errs["g_LS_diag"] = relmax(torch.diagonal(gl), torch.from_numpy(g[pre + "g_LS_diag"]))
        errs["g_LS_lastrows"] = relmax(gl[-8:, :], torch.from_numpy(g[pre + "g_LS_lastrows"]))
        for k in O.PARAM_NAMES:
            if k not in ("chol_variational_covar", "inducing_directions"):
                errs["g_" + k] = relmax(grads[k], torch.from_numpy(g[pre + "g_" + k]))
        _report("C3 %s %s" % (mll, "fast" if fast else "per-output"), errs)
        # stated tolerances: loss 5e-6, mean 2e-4, variance 5e-5, gradients 2e-3 (1e-8 Cholesky jitter: K_ZZ is worse conditioned here than
        # at C4) -- observed: loss 1e-7, mean 2.4e-5, variance 2e-6, gradients <= 2.7e-4 (last rows of L_S-bar), <= 5e-5 otherwise
        assert errs["loss"] < 5e-6 and errs["mu"] < 2e-4 and errs.get("var", 0.0) < 5e-5, errs
        assert errs["g_LS_norm"] < 1e-4
        for k, v in errs.items():
            if k.startswith("g_"):
                assert v < 2e-3, (k, v)


def _graph_loop(dsvgp, gpu_device, graph, lr_sched=None, seed=7, M=24):
    torch.manual_seed(11)                                    # the 1e-3 randn initialisation of the variational mean
    g = torch.Generator().manual_seed(5)
    X = torch.rand(900, 4, generator=g).to(gpu_device)
    Y = O.testfun(X.cpu()).to(gpu_device)
    loop = dsvgp.setup_training(None, num_inducing=M, num_directions=2, minibatch_size=128, minibatch_dim=2, num_epochs=3,
                                learning_rate_hypers=0.01, lr_sched=lr_sched, seed=seed, tensors=(X, Y))
    loop.graph = graph
    return loop


@pytest.mark.parametrize("lr_sched", [None, "step_lr"])
def test_graph_replay_matches_eager_steps(dsvgp, gpu_device, lr_sched):
    """HIP-graph replay of the step (gather -> ELBO forward / backward -> both Adam updates captured once, replayed with
    per-step indices, derivative columns, learning rates and step counts read from device memory) against the eager
    loop: same losses (to the run-to-run spread of the split-K atomics), same parameters, same optimizer state, and the
    per-iteration LR schedule honoured (``step_lr`` puts both milestones inside the 21 steps)."""
    out = {}
    for graph in (False, True):
        loop = _graph_loop(dsvgp, gpu_device, graph, lr_sched)
        losses = []
        for epoch in range(3):
            perm = loop.epoch_permutation()
            for k in range(7):                                # 7 full batches of 128 (the ragged tail is left out)
                loss, _, _ = loop.step(perm[k * 128:(k + 1) * 128])
                losses.append(loss.item())                    # (reads the static loss tensor after the replay)
        loop.finish()
        assert bool(loop._graphs) == graph
        P = {k: v.detach().clone() for k, v in loop.model._param_dict(loop.likelihood).items()}
        steps = [loop.variational_optimizer.state[p]["step"] for p in loop.variational_optimizer.param_groups[0]["params"]]
        out[graph] = (losses, P, steps, [g["lr"] for g in loop.hyperparameter_optimizer.param_groups])
    (le, Pe, se, lre), (lg, Pg, sg, lrg) = out[False], out[True]
    assert se == sg == [21, 21] and lre == lrg
    for a, b in zip(le, lg):
        assert abs(a - b) < 2e-5 * abs(a), (a, b)
    for k in Pe:
        # (Adam moves a parameter by ~lr per step whatever the gradient's size: last-bit differences of the split-K atomics in a
        #  near-zero gradient component show up as ~1e-4..1e-3 after 21 steps of lr = 0.01, in eager-vs-eager runs too)
        assert (Pe[k] - Pg[k]).abs().max().item() < 2e-3 * max(Pe[k].abs().max().item(), 1e-2), k


def test_graph_replay_failed_factorisation_leaves_parameters_untouched(dsvgp, gpu_device):
    """A replayed step whose K_ZZ is not positive definite: the captured Adam kernels are guarded by the potrf status word, so
    nothing is updated; the host sees the status before the next launch, redoes the step eagerly through the jitter ladder and
    raises NotPSDError like the reference's psd_safe_cholesky."""
    loop = _graph_loop(dsvgp, gpu_device, True)
    perm = loop.epoch_permutation()
    for k in range(5):
        loop.step(perm[k * 128:(k + 1) * 128])
    loop.finish()
    assert loop._graphs
    before = {k: v.detach().clone() for k, v in loop.model._param_dict(loop.likelihood).items()}
    with torch.no_grad():
        loop.model.covar_module.raw_outputscale.fill_(float("nan"))
    loop.step(perm[5 * 128:6 * 128])                          # replayed: every gradient is NaN, the guard holds the update back
    with pytest.raises(dsvgp.NotPSDError):
        loop.finish()
    after = loop.model._param_dict(loop.likelihood)
    for k in before:
        if k != "raw_outputscale":
            assert torch.equal(before[k], after[k]), k


@pytest.mark.parametrize("lr_sched", [None, "step_lr"])
def test_deferred_status_loop_matches_the_waiting_loop(dsvgp, gpu_device, lr_sched):
    """the eager loop with the factorisation's status read one step late (the optimizers' update queued at once, guarded on the device by the
    status word: TrainLoop._eager_step, optim.step_together(guard=)) against the loop that waits for the status before the update: same losses,
    parameters, optimizer state and learning rates, as between graph replay and eager steps"""
    out = {}
    for defer in (False, True):
        loop = _graph_loop(dsvgp, gpu_device, False, lr_sched, M=40)
        loop.defer_status = defer
        losses, deferred = [], 0
        for epoch in range(3):
            perm = loop.epoch_permutation()
            for k in range(7):
                loss, _, _ = loop.step(perm[k * 128:(k + 1) * 128])
                deferred += loop._deferred_pending is not None
                losses.append(loss.item())
        loop.finish()
        assert loop._deferred_pending is None and loop.model.engine._deferred is None
        assert deferred == (21 if defer else 0) and loop.model.engine.c_step_used
        P = {k: v.detach().clone() for k, v in loop.model._param_dict(loop.likelihood).items()}
        steps = [loop.variational_optimizer.state[q]["step"] for q in loop.variational_optimizer.param_groups[0]["params"]]
        out[defer] = (losses, P, steps, [g["lr"] for g in loop.hyperparameter_optimizer.param_groups])
    (le, Pe, se, lre), (lg, Pg, sg, lrg) = out[False], out[True]
    assert se == sg == [21, 21] and lre == lrg
    for a, b in zip(le, lg):
        assert abs(a - b) < 2e-5 * abs(a), (a, b)
    for k in Pe:
        assert (Pe[k] - Pg[k]).abs().max().item() < 2e-3 * max(Pe[k].abs().max().item(), 1e-2), k


def test_deferred_status_failed_factorisation_leaves_parameters_untouched(dsvgp, gpu_device):
    """a deferred step whose K_ZZ is not positive definite: the guarded update does nothing, the status is read before the next step (here:
    finish()), the step is redone through the jitter ladder and raises NotPSDError like the reference's psd_safe_cholesky -- parameters and
    step counts as before the step"""
    loop = _graph_loop(dsvgp, gpu_device, False, M=40)
    loop.defer_status = True
    perm = loop.epoch_permutation()
    for k in range(4):
        loop.step(perm[k * 128:(k + 1) * 128])
    loop.finish()
    before = {k: v.detach().clone() for k, v in loop.model._param_dict(loop.likelihood).items()}
    with torch.no_grad():
        loop.model.covar_module.raw_outputscale.fill_(float("nan"))
    loop.step(perm[4 * 128:5 * 128])                          # every gradient is NaN; the guard holds the update back
    assert loop._deferred_pending is not None
    with pytest.raises(dsvgp.NotPSDError):
        loop.finish()
    after = loop.model._param_dict(loop.likelihood)
    for k in before:
        if k != "raw_outputscale":
            assert torch.equal(before[k], after[k]), k


def test_guarded_adam_launch_is_held_back_by_a_non_zero_guard_word(dsvgp, gpu_device):
    dev = gpu_device
    w0 = torch.randn(300, 7)
    pa, pb = [torch.nn.Parameter(w0.clone().to(dev)), torch.nn.Parameter(torch.randn(5).to(dev))], None
    oa = dsvgp.FusedAdam(pa, lr=0.05)
    for q in pa:
        q.grad = torch.randn_like(q)
    guard = torch.ones(1, dtype=torch.int32, device=dev)
    assert dsvgp.optim.step_together([oa], guard=guard)
    assert torch.equal(pa[0].detach().cpu(), w0)             # held back
    guard.zero_()
    for q in pa:
        oa.state[q]["step"] -= 1
    assert dsvgp.optim.step_together([oa], guard=guard)
    ref = torch.optim.Adam([w0.clone().requires_grad_(True)], lr=0.05)
    ref.param_groups[0]["params"][0].grad = pa[0].grad.cpu()
    ref.step()
    assert relmax(pa[0].detach(), ref.param_groups[0]["params"][0].detach()) < 2e-6


def test_legacy_unwhitened_checkpoint_is_converted_on_first_call(dsvgp, gpu_device):
    """reference DGVS.py:210-240: a checkpoint whose q(u) = N(m_u, L_u L_u^T) is NOT whitened (no ``updated_strategy`` key) is
    re-parameterised on the first call, m_w = L^-1 (m_u - c), L_w = chol(L^-1 S_u L^-T) with L = chol(K_ZZ + 1e-3 I); afterwards
    the model predicts what the un-whitened q(u) means:  mu = c + K_XZ K^-1 (m_u - c),
    var = diag K_XX + 1e-4 - diag(K_XZ K^-1 K_ZX) + diag(K_XZ K^-1 S_u K^-1 K_ZX)  (+ noise)."""
    import warnings
    d, M, p, nx = 3, 14, 2, 40
    g = torch.Generator().manual_seed(8)
    Z = torch.rand(M, d, generator=g)
    V = torch.eye(d)[:p].repeat(M, 1) + 0.1 * torch.randn(M * p, d, generator=g)
    Mp = M * (p + 1)
    m_u = 0.5 * torch.randn(Mp, generator=g)
    L_u = torch.tril(0.3 * torch.eye(Mp) + 0.05 * torch.randn(Mp, Mp, generator=g))
    src = dsvgp.GPModel(Z, V, d)
    lik = dsvgp.gp_shim.GaussianLikelihood()
    with torch.no_grad():
        src.mean_module.constant.fill_(0.2)
        src.covar_module.raw_outputscale.fill_(0.3)
        src.covar_module.base_kernel.raw_lengthscale.fill_(0.4)
        src.variational_strategy._variational_distribution.variational_mean.copy_(m_u)
        src.variational_strategy._variational_distribution.chol_variational_covar.copy_(L_u)
        src.variational_strategy.variational_params_initialized.fill_(1)
    sd = src.state_dict()
    del sd["variational_strategy.updated_strategy"]                      # what an old gpytorch wrote
    model = dsvgp.GPModel(torch.rand(M, d), torch.eye(d)[:p].repeat(M, 1), d)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.load_state_dict(sd)
    model, lik = model.to(gpu_device), lik.to(gpu_device)
    model.eval(); lik.eval()
    x = torch.rand(nx, d, generator=g)
    D = torch.eye(d)[:p].repeat(nx, 1)
    with torch.no_grad():
        preds = lik(model(x.to(gpu_device), derivative_directions=D.to(gpu_device)))
        mean, var = preds.mean.cpu().double(), preds.variance.cpu().double()
    assert bool(model.variational_strategy.updated_strategy)
    # fp64 ground truth of the un-whitened predictive
    P = {k: v.detach().cpu().double() for k, v in src._param_dict(lik.cpu()).items()}
    ell, s, noise = O.constrained(P)
    Kzz = s * O.kernel_matrix(P["inducing_points"], P["inducing_points"], P["inducing_directions"], P["inducing_directions"], ell)
    Kzz = Kzz + 1e-3 * torch.eye(Mp, dtype=torch.float64)
    Kzx = s * O.kernel_matrix(P["inducing_points"], x.double(), P["inducing_directions"], D.double(), ell)
    Kinv_Kzx = torch.linalg.solve(Kzz, Kzx)
    S_u = L_u.double() @ L_u.double().t()
    mu_ref = 0.2 + Kinv_Kzx.t() @ (m_u.double() - 0.2)
    var_ref = (s * O.kernel_diag(nx, p, ell) + 1e-4 - (Kzx * Kinv_Kzx).sum(0) + (Kinv_Kzx * (S_u @ Kinv_Kzx)).sum(0) + noise)
    assert relmax(mean, mu_ref) < 5e-4 and relmax(var, var_ref) < 5e-4
    # the prior p(u) the reference evaluates with ``prior=True``
    pr = model.variational_strategy(model.variational_strategy.inducing_points, prior=True)
    assert relmax(pr.covariance_matrix, Kzz - 1e-3 * torch.eye(Mp, dtype=torch.float64)) < 2e-5 and relmax(pr.loc, torch.full((Mp,), 0.2)) < 1e-6


@pytest.mark.parametrize("harness", ["grad_svgp", "traditional_vi"])
def test_plain_ciq_strategy_of_the_other_harnesses(dsvgp, gpu_device, harness, capsys):
    """``use_ciq=True`` in grad_svgp / traditional_vi builds gpytorch's plain CiqVariationalStrategy (grad_svgp.py:25-27,
    traditional_vi.py:22-24): the same CIQ + NGD terms with K_ZZ.add_jitter(1e-2) and diag K_XX + 1e-4 (its forward is quoted in the
    reference at CiqDGVS.py:243-251).  One step of the engine against the oracle with those constants, then the drop-in."""
    from torch.utils.data import TensorDataset
    from test_ngd import make_ngd_problem
    full = harness == "grad_svgp"
    d = 3
    p = d if full else 0
    P, x, y, D, nd = make_ngd_problem(400, d, 18, p, 60, seed=77) if p else make_ngd_problem(400, d, 40, 0, 60, seed=78)
    if full:
        P["inducing_directions"] = torch.eye(d).repeat(18, 1)
        D = torch.eye(d).repeat(60, 1)
        g = torch.Generator().manual_seed(1)
        y = O.testfun(x).reshape(-1).contiguous()
    st = {}
    l_ref, g_ref, mu_ref, var_ref = O.ciq_loss_and_grads(P, x, y, D, nd, stats=st, kzz_jitter=1e-2, kxx_jitter=1e-4)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.whitening, eng.kzz_jitter, eng.ciq_kxx_jitter = "ciq", 1e-2, 1e-4
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    assert abs(loss.item() - l_ref.item()) < 1e-3 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 5e-3 and relmax(varn, var_ref) < 5e-3
    for k in O.NGD_PARAM_NAMES:
        if g_ref[k].numel() and g_ref[k].abs().max() > 0 and not (full and k == "inducing_directions"):
            assert relmax(grads[k], g_ref[k]) < 2e-2, k
    # and without the jitter constants the numbers differ (the knobs are live)
    eng2 = dsvgp.ElboEngine(gpu_device)
    eng2.whitening = "ciq"
    loss2, _, _, _ = eng2.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    assert abs(loss2.item() - loss.item()) > 1e-4 * abs(loss.item())      # (HIP vs HIP: the two jitters move the loss by ~3e-4)
    # drop-in
    torch.manual_seed(0)
    n = 400
    tx = torch.rand(n, 2)
    ty = O.testfun(tx) if full else O.testfun(tx)[:, 0].contiguous()
    H = getattr(dsvgp, harness)
    model, likelihood = H.train_gp(TensorDataset(tx, ty), 2, num_inducing=16, minibatch_size=100, num_epochs=30, use_ciq=True,
                                   learning_rate_ngd=0.1, num_contour_quadrature=15, tqdm=False, seed=3)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 2 and all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
    assert model.engine.whitening == "ciq" and model.engine.kzz_jitter == 1e-2 and model.engine.ciq_kxx_jitter == 1e-4
    assert "variational_strategy._variational_distribution.natural_mat" in model.state_dict()
    means, variances = H.eval_gp(TensorDataset(tx[:50], ty[:50]), model, likelihood, minibatch_size=25)
    assert torch.isfinite(means).all() and (variances > 0).all()


# ------------------------------------------------------------------ the reference's own strategy forward, as vectors
@pytest.mark.parametrize("path", STRATEGY, ids=[__import__("os").path.basename(p) for p in STRATEGY])
def test_predictive_matches_reference_strategy_vectors(dsvgp, gpu_device, path):
    """q(f) of the HIP engine (mean, full covariance) against what the reference's strategy ``forward`` text computed in fp64
    (oracle/make_strategy_fixtures.py; DGVS.py:89-208, DFree / Shared siblings).  fp32 model: 2e-4 of the max magnitude."""
    P, x, D, fl, mean_ref, cov_ref = strategy_problem(path, torch.float32)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.data_outputs = fl["outputs"]
    eng.shared_directions = fl["shared"]
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    mu, Sigma = eng.predict_joint(Pg, x.to(gpu_device), D.to(gpu_device))
    noise = torch.nn.functional.softplus(torch.zeros(())) + 1e-4
    Sigma = Sigma.double().cpu() - noise.double() * torch.eye(Sigma.shape[0], dtype=torch.float64)
    errs = dict(mean=relmax(mu, mean_ref), cov=relmax(Sigma, cov_ref))
    _report("strategy vector " + __import__("os").path.basename(path), errs)
    assert errs["mean"] < 2e-4 and errs["cov"] < 2e-4
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mean_ref) < 2e-4 and relmax(varn2.double().cpu() - noise.double(), torch.diagonal(cov_ref)) < 2e-4


@pytest.mark.parametrize("path", GRADIENT, ids=[__import__("os").path.basename(p) for p in GRADIENT])
@pytest.mark.parametrize("fast", [True, False])
def test_step_matches_autograd_through_the_reference_forward(dsvgp, gpu_device, path, fast):
    """loss and every gradient of the HIP step (Gram fast path and per-output path) against torch autograd run THROUGH the
    reference's own strategy forward and kernel file in fp64 (tests/golden/strategy_grad_*.npz)"""
    P, x, y, D, nd, fl, loss_ref, g_ref = gradient_problem(path, torch.float32)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.data_outputs, eng.shared_directions = fl["outputs"], fl["shared"]
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO", fast=fast)
    errs = {"loss": abs(loss.item() - loss_ref) / abs(loss_ref)}
    for k in PARAM_KEYS:
        gk = torch.tril(grads[k]) if k == "chol_variational_covar" else grads[k]
        errs[k] = relmax(gk, g_ref[k])
    _report("reference-forward gradient vector %s (fast=%s)" % (__import__("os").path.basename(path), fast), errs)
    assert errs["loss"] < 2e-5 and max(errs[k] for k in PARAM_KEYS) < 2e-3, errs


# ------------------------------------------------------------------ directions stated as an index list (canonical-direction assembly)
def _stated(dsvgp, D, cols, dev):
    """the batch's direction matrix on the device with the caller's statement attached (what TrainLoop._device_step does):
    row j p + b = e_{cols[b + 1] - 1}"""
    idx = torch.tensor(cols, dtype=torch.int32, device=dev)
    return dsvgp._ops.state_directions(D.to(dev), idx[1:], 1)


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,M,p,B", [(600, 5, 40, 2, 128), (500, 20, 30, 5, 96), (700, 20, 30, 5, 131), (3000, 5, 200, 2, 512), (900, 28, 17, 5, 77)])
@pytest.mark.parametrize("path", ["one-call", "piecewise", "per-output", "PLL"])
def test_step_with_stated_one_hot_directions(dsvgp, gpu_device, monkeypatch, N, d, M, p, B, path):
    """K_ZX and its backward on the canonical-direction kernels (io->dir_idx / _ops.state_directions) against the same step on the
    general kernels (same inputs, the statement left out) and against the float64 oracle at the tolerance of test_step_matches_oracle.
    DSVGP_CHECK_DIRS=1: the statement itself is checked against D."""
    monkeypatch.setenv("DSVGP_CHECK_DIRS", "1")
    assert dsvgp._ops.canon_supported(d, p)
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=N + d + 3)
    cols = [0] + [int(r.argmax()) + 1 for r in D[:p]]        # (make_problem drew the columns from its generator: read them off D)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xd, yd = x.to(gpu_device), y.to(gpu_device)
    mll = "PLL" if path == "PLL" else "ELBO"
    fast = path in ("one-call", "piecewise")
    out = {}
    for stated in (False, True):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.c_step = path != "piecewise"
        Dd = _stated(dsvgp, D, cols, gpu_device) if stated else D.to(gpu_device)
        out[stated] = eng.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=fast)
        torch.cuda.synchronize()
        assert (eng._zx_dirs is not None or eng.c_step_used) if stated else eng._zx_dirs is None
    (l0, g0, mu0, v0), (l1, g1, mu1, v1) = out[False], out[True]
    errs = {"loss": abs(l1.item() - l0.item()) / abs(l0.item()), "mu": relmax(mu1, mu0)}
    # (two assemblies of K_ZX that differ in the last bits of its entries -- 2e-7 -- seen through L^-1: observed 1.4e-5 on mu at M' = 600,
    #  1e-5 on the inducing-point gradients; a wrong operand shows at 1e-3+.  The float64 oracle below is the bound that counts.)
    assert errs["loss"] < 4e-6 and errs["mu"] < 1e-4, errs
    if not fast:
        errs["varn"] = relmax(v1, v0)
        assert errs["varn"] < 1e-4, errs
    for k in O.PARAM_NAMES:
        errs[k] = relmax(g1[k], g0[k])
        assert errs[k] < 2e-4, (k, errs[k])
    P64 = {k: v.double() for k, v in P.items()}
    _, g64, _, _ = O.elbo_loss_and_grads(P64, x.double(), y.double(), D.double(), nd, mll)
    for k in O.PARAM_NAMES:
        e = relmax(g1[k], g64[k])
        errs[k + "(f64 oracle)"] = e
        assert e < GRAD_TOL_FP64, (k, e)
    _report("stated directions N=%d d=%d M=%d p=%d B=%d %s" % (N, d, M, p, B, path), errs)


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,B", [(900, 37, 96), (2000, 100, 131), (3000, 300, 512)])
@pytest.mark.parametrize("path", ["one-call", "per-output", "PLL"])
def test_full_gradient_step_with_unit_directions_stated_on_both_sides(dsvgp, gpu_device, monkeypatch, N, M, B, path):
    """the full-gradient SVGP (d = p = 10, V = I_d at every inducing point, D = I_d at every data point: reference
    GradVariationalStrategy.py:89-99) with both direction sets stated as the index list 0 .. d-1: K_ZZ, K_ZX and their backwards on the
    both-sides one-hot kernels (io->v_one_hot) against the same step on the general kernels and against the float64 oracle."""
    monkeypatch.setenv("DSVGP_CHECK_DIRS", "1")
    d = p = 10
    ops = dsvgp._ops
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=N + M)
    P["inducing_directions"] = torch.eye(d).repeat(M, 1)
    assert torch.equal(D, torch.eye(d).repeat(B, 1))
    xd, yd = x.to(gpu_device), y.to(gpu_device)
    mll = "PLL" if path == "PLL" else "ELBO"
    fast = path == "one-call"
    out = {}
    for stated in (False, True):
        Pg = {k: v.to(gpu_device) for k, v in P.items()}
        Dd = D.to(gpu_device)
        if stated:
            rng = ops.index_range(gpu_device, d)
            ops.state_directions(Dd, rng, 0)
            ops.state_directions(Pg["inducing_directions"], rng, 0)
        eng = dsvgp.ElboEngine(gpu_device)
        out[stated] = eng.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=fast)
        torch.cuda.synchronize()
        assert eng.c_step_used
        plan = list(eng._plans.values())[0]
        assert bool(plan.io.v_one_hot) == stated and bool(plan.io.dir_idx) == stated
    (l0, g0, mu0, v0), (l1, g1, mu1, v1) = out[False], out[True]
    errs = {"loss": abs(l1.item() - l0.item()) / abs(l0.item()), "mu": relmax(mu1, mu0)}
    assert errs["loss"] < 4e-6 and errs["mu"] < 1e-4, errs
    # (fixed directions: no gradient is formed -- unless an upstream matrix's rows are not 16-byte pieces, M' odd or B' not a multiple of 4:
    #  that backward then runs on the general kernel, which forms one)
    if (M * (p + 1)) % 2 == 0 and (B * (p + 1)) % 4 == 0:
        assert g1["inducing_directions"].abs().max().item() == 0.0
    P64 = {k: v.double() for k, v in P.items()}
    _, g64, _, _ = O.elbo_loss_and_grads(P64, x.double(), y.double(), D.double(), nd, mll)
    for k in O.PARAM_NAMES:
        if k == "inducing_directions":
            continue
        errs[k] = relmax(g1[k], g0[k])
        errs[k + "(f64 oracle)"] = relmax(g1[k], g64[k])
        assert errs[k] < 2e-4 and errs[k + "(f64 oracle)"] < GRAD_TOL_FP64, (k, errs[k], errs[k + "(f64 oracle)"])
    _report("full gradient, stated N=%d M=%d B=%d %s" % (N, M, B, path), errs)


@pytest.mark.gpu
def test_a_wrong_statement_about_the_directions_is_caught_by_the_check(dsvgp, gpu_device, monkeypatch):
    monkeypatch.setenv("DSVGP_CHECK_DIRS", "1")
    P, x, y, D, nd = make_problem(600, 5, 40, 2, 128, seed=11)
    cols = [0] + [int(r.argmax()) + 1 for r in D[:2]]
    cols[1] = cols[1] % 5 + 1 if cols[1] % 5 + 1 != cols[2] else (cols[1] + 1) % 5 + 1
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    with pytest.raises(dsvgp._lib.DsvgpError):
        dsvgp.ElboEngine(gpu_device).loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), _stated(dsvgp, D, cols, gpu_device), nd)


# ------------------------------------------------------------------ the whole step from one host call (csrc/step.hip)
@pytest.mark.parametrize("N,d,M,p,B", [(600, 5, 40, 2, 128), (500, 20, 30, 5, 96), (300, 6, 70, 0, 64), (3000, 5, 200, 2, 512),
                                       (6000, 20, 370, 5, 700)])
def test_one_call_step_equals_the_piecewise_step(dsvgp, gpu_device, N, d, M, p, B):
    """dsvgp_elbo_step_f32 queues the same library calls in the same order as the Python-orchestrated fast path: loss, mean and
    every gradient agree to the run-order noise of the split-K atomics (1e-5 typical, 5e-5 asserted), with and without the second stream, and against
    the oracle like the piecewise path.  (M' = 600 and 2220 >= 512: the overlap schedule is on by default there.)"""
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=N + d + 1)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xd, yd, Dd = x.to(gpu_device), y.to(gpu_device), D.to(gpu_device)
    ref = dsvgp.ElboEngine(gpu_device)
    ref.c_step = False
    l0, g0, mu0, _ = ref.loss_and_grads(Pg, xd, yd, Dd, nd)
    assert not ref.c_step_used
    for overlap in (None, True, False):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.overlap = overlap
        l1, g1, mu1, varn1 = eng.loss_and_grads(Pg, xd, yd, Dd, nd)
        torch.cuda.synchronize()
        assert eng.c_step_used and varn1.numel() == 0
        # (two runs of the same arithmetic: what differs is the order in which fp32 atomics meet -- a few ulps of the loss)
        assert abs(l1.item() - l0.item()) < 4e-6 * abs(l0.item()), (overlap, l1.item(), l0.item())
        assert relmax(mu1, mu0) < 4e-6, (overlap, relmax(mu1, mu0))
        for k in O.PARAM_NAMES:
            if g0[k].numel():
                # (observed up to 2.3e-5 on inducing_directions at M' = 120 in one run of eight: the direction gradients are sums
                #  with cancellation of terms whose split-K partial sums meet in a different order; a wrong operand shows at 1e-3+)
                assert relmax(g1[k], g0[k]) < 5e-5, (overlap, k, relmax(g1[k], g0[k]))
        assert g1["chol_variational_covar"].triu(1).abs().max().item() == 0.0
        l2, g2, _, _ = eng.loss_and_grads(Pg, xd, yd, Dd, nd)        # second call on the same plan / workspace
        assert abs(l2.item() - l1.item()) < 4e-6 * abs(l1.item()), (overlap, l2.item(), l1.item())
        assert relmax(g2["inducing_points"], g1["inducing_points"]) < 5e-5, (overlap, relmax(g2["inducing_points"], g1["inducing_points"]))
    l_ref, g_ref, mu_ref, _ = O.elbo_loss_and_grads(P, x, y, D, nd)
    assert abs(l1.item() - l_ref.item()) < 2e-5 * abs(l_ref.item()) and relmax(mu1, mu_ref) < 2e-4


def test_one_call_step_with_the_solve_in_row_ranges_under_the_chain(dsvgp, gpu_device):
    """Flag 128 of dsvgp_elbo_step_f32 (DSVGP_SOLVE_PIPE; measured and left off by default): the forward solve A = L^-1 K_ZX queued as
    three row ranges, two of them on the side stream behind events the Cholesky chain records after its launches k1 / k2.  Every piece is
    a row range of the SAME triangular product (tri_off pieces of gemm64.hip's wide kernel), so [A ; mu-bar^T] must equal the serial
    schedule's BIT FOR BIT (the solve has no split-K), and loss / gradients to the run-order noise of the later atomics.  Geometry:
    M' = 1024 (16 block rows), B' = 32768 -> 8192 tiles of 64 x 64, the smallest the flag accepts."""
    N, d, M, p, B = 9000, 6, 256, 3, 8192
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=31)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xd, yd, Dd = x.to(gpu_device), y.to(gpu_device), D.to(gpu_device)
    res = []
    for pipe in (False, True):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.solve_pipe = pipe
        l1, g1, mu1, _ = eng.loss_and_grads(Pg, xd, yd, Dd, nd)
        torch.cuda.synchronize()
        assert eng.c_step_used
        _, plan, ws, _ = eng._last_fast
        res.append((l1.item(), {k: v.clone() for k, v in g1.items()}, mu1.clone(), plan.locate(ws, 0).clone()))
    (l0, g0, mu0, A0), (l1, g1, mu1, A1) = res
    assert A0.shape == (M * (p + 1) + 1, B * (p + 1)) and torch.isfinite(A0).all()
    assert torch.equal(A0[:-1], A1[:-1])                         # rows of A: the same product, piece by piece
    assert relmax(A1[-1], A0[-1]) < 4e-6 and relmax(mu1, mu0) < 4e-6
    assert abs(l1 - l0) < 4e-6 * abs(l0), (l1, l0)
    for k in O.PARAM_NAMES:
        if g0[k].numel():
            assert relmax(g1[k], g0[k]) < 5e-5, (k, relmax(g1[k], g0[k]))


@pytest.mark.parametrize("N,d,M,p,B", [(600, 5, 40, 2, 128), (500, 20, 30, 5, 96), (3000, 5, 200, 2, 512), (400, 6, 300, 6, 64)])
@pytest.mark.parametrize("mll", ["PLL", "ELBO"])
def test_one_call_per_output_step_equals_the_piecewise_step(dsvgp, gpu_device, N, d, M, p, B, mll):
    """dsvgp_elbo_step_po_f32 (round 5): the per-output step -- mll_type="PLL" (what the reference's tests/test_grad_svgp.py trains
    with, directional_vi.py:218-219) or the ELBO with per-output variances -- queues the library calls of the Python-orchestrated
    per-output path from ONE host call: loss, mean, per-output variance and every gradient agree with that path to the run-order
    noise of the split-K atomics, and with the oracle like it."""
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=N + d + 3)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xd, yd, Dd = x.to(gpu_device), y.to(gpu_device), D.to(gpu_device)
    ref = dsvgp.ElboEngine(gpu_device)
    ref.c_step = False
    l0, g0, mu0, vn0 = ref.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=False)
    assert not ref.c_step_used and vn0.numel() == B * (p + 1)
    for overlap in (None, False):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.overlap = overlap
        l1, g1, mu1, vn1 = eng.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=False)
        torch.cuda.synchronize()
        assert eng.c_step_used and vn1.numel() == B * (p + 1)
        assert abs(l1.item() - l0.item()) < 4e-6 * abs(l0.item()), (overlap, l1.item(), l0.item())
        assert relmax(mu1, mu0) < 4e-6 and relmax(vn1, vn0) < 4e-6, (overlap, relmax(mu1, mu0), relmax(vn1, vn0))
        for k in O.PARAM_NAMES:
            if g0[k].numel():
                assert relmax(g1[k], g0[k]) < 5e-5, (overlap, k, relmax(g1[k], g0[k]))
        assert g1["chol_variational_covar"].triu(1).abs().max().item() == 0.0
        l2, g2, _, _ = eng.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=False)        # second call on the same plan / workspace
        assert abs(l2.item() - l1.item()) < 4e-6 * abs(l1.item())
    l_ref, g_ref, mu_ref, vn_ref = O.elbo_loss_and_grads(P, x, y, D, nd, mll)
    assert abs(l1.item() - l_ref.item()) < 2e-5 * abs(l_ref.item()) and relmax(mu1, mu_ref) < 2e-4 and relmax(vn1, vn_ref) < 2e-4
    for k in O.PARAM_NAMES:
        if g_ref[k].numel() and g_ref[k].abs().max().item() > 0:
            assert relmax(g1[k], g_ref[k]) < 3e-3, (k, relmax(g1[k], g_ref[k]))


def test_one_call_step_falls_back_to_the_jitter_ladder(dsvgp, gpu_device):
    """a K_ZZ that needs psd_safe_cholesky's retries: the one-call step reports the failed factorisation through its status word and
    the engine redoes the step on the piecewise path (same result as with the one-call path switched off)"""
    N, d, M, p, B = 300, 3, 30, 1, 40
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=5)
    P["inducing_points"][1] = P["inducing_points"][0]                # duplicated inducing point ...
    P["inducing_directions"][1] = P["inducing_directions"][0]
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    out = []
    for c_step in (True, False):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.c_step = c_step
        eng.kzz_jitter = 0.0                                         # ... and no add_jitter: singular K_ZZ, the ladder has to act
        loss, grads, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
        assert torch.isfinite(loss).item() and not eng.c_step_used
        out.append((loss.item(), grads["variational_mean"].cpu()))
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[1][0]) and relmax(out[0][1], out[1][1]) < 1e-4


def test_one_call_step_alternating_batch_shapes(dsvgp, gpu_device):
    """an epoch's ragged tail batch gets its own plan AND its own workspace; going back and forth between the two batch shapes
    gives the piecewise path's numbers every time (the plan clears the zero padding of [Q' | a] once per workspace only)"""
    N, d, M, p = 900, 5, 43, 2                      # M' = 129: [Q' | a] has 130 columns, padded to 132
    P, x, y, D, nd = make_problem(N, d, M, p, 200, seed=8)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    eng, ref = dsvgp.ElboEngine(gpu_device), dsvgp.ElboEngine(gpu_device)
    ref.c_step = False
    for B in (200, 77, 200, 77, 131):
        xb, yb, Db = x[:B].to(gpu_device), y[:B * (p + 1)].to(gpu_device), D[:B * p].to(gpu_device)
        l1, g1, mu1, _ = eng.loss_and_grads(Pg, xb, yb, Db, nd)
        l0, g0, mu0, _ = ref.loss_and_grads(Pg, xb, yb, Db, nd)
        assert eng.c_step_used and not ref.c_step_used
        assert abs(l1.item() - l0.item()) < 4e-6 * abs(l0.item()) and relmax(mu1, mu0) < 4e-6, B
        for k in O.PARAM_NAMES:
            assert relmax(g1[k], g0[k]) < 5e-5, (B, k, relmax(g1[k], g0[k]))     # (run-order noise of the atomics, see above)


@pytest.mark.parametrize("N,d,M,p,B", [(40, 2, 1, 1, 1), (40, 3, 2, 0, 3), (60, 1, 5, 1, 7), (80, 4, 3, 4, 2), (200, 6, 11, 2, 65)])
def test_one_call_step_on_degenerate_shapes(dsvgp, gpu_device, N, d, M, p, B):
    """one inducing point, one data row, p = 0, d = 1, p = d, a batch one row past a 64-row tile: the one-call step against the
    piecewise path and the oracle (what the reference's own tiny smoke runs exercise: tests/test_dsvgp.py with small n)"""
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=3 * N + d)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    xd, yd, Dd = x.to(gpu_device), y.to(gpu_device), D.to(gpu_device)
    eng, ref = dsvgp.ElboEngine(gpu_device), dsvgp.ElboEngine(gpu_device)
    ref.c_step = False
    l1, g1, mu1, _ = eng.loss_and_grads(Pg, xd, yd, Dd, nd)
    l0, g0, mu0, _ = ref.loss_and_grads(Pg, xd, yd, Dd, nd)
    assert eng.c_step_used and not ref.c_step_used
    l_ref, g_ref, mu_ref, _ = O.elbo_loss_and_grads(P, x, y, D, nd)
    assert abs(l1.item() - l0.item()) < 4e-6 * abs(l0.item()) and abs(l1.item() - l_ref.item()) < 2e-5 * abs(l_ref.item())
    assert relmax(mu1, mu_ref) < 2e-4
    for k in O.PARAM_NAMES:
        if g0[k].numel() == 0:
            continue
        if g_ref[k].abs().max().item() < 1e-6:          # (d = 1: a normalised direction has no gradient; both sides are round-off)
            assert g1[k].abs().max().item() < 1e-5 and g0[k].abs().max().item() < 1e-5, k
            continue
        assert relmax(g1[k], g0[k]) < 5e-5, (k, relmax(g1[k], g0[k]))
        assert relmax(g1[k], g_ref[k]) < 1e-3, (k, relmax(g1[k], g_ref[k]))


@pytest.mark.parametrize("c_step", [True, False])
def test_value_variances_of_the_reporting_step(dsvgp, gpu_device, c_step):
    """The reference's every-50th-step nll print reads ``output.variance.sqrt()[::p+1]`` of the forward pass it differentiates
    (directional_vi.py:255-260).  ``TrainLoop.step(need_variance="values")`` keeps that step on the ELBO fast path and forms
    those rows from the A it leaves behind: held to the oracle's predictive at the parameters the step STARTED from."""
    torch.manual_seed(1)
    n, d, M, p, B = 2000, 4, 30, 2, 256
    X = torch.rand(n, d)
    Y = O.testfun(X)
    loop = dsvgp.setup_training(None, num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p, num_epochs=1,
                                learning_rate_hypers=0.05, seed=3, tensors=(X.to(gpu_device), Y.to(gpu_device)))
    eng = loop.model.engine
    eng.c_step = c_step
    idx = torch.arange(B, device=gpu_device)
    for _ in range(10):                         # plain steps first: q(u) away from N(0, I), so that W != A matters
        loop.step(idx)
    P = {k: v.detach().cpu().clone() for k, v in loop.model._param_dict(loop.likelihood).items()}
    state = loop.col_rng.getstate()
    cols = sorted(loop.col_rng.sample(range(1, d + 1), p) + [0])      # the columns the next step will draw (directional_vi.py:68-90)
    loop.col_rng.setstate(state)
    loss, output, yb = loop.step(idx, need_variance="values")
    vv, mean = output.value_variance, output.mean
    torch.cuda.synchronize()
    assert getattr(output, "_value_varn", None) is not None and eng.c_step_used == c_step, "the reporting step left the fast path"
    assert vv.shape == (B,) and mean.shape == (B * (p + 1),)
    D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1)
    P64 = {k: v.double() for k, v in P.items()}
    mu64, var64 = O.predictive(P64, X[:B].double(), D.double())
    _, _, noise = O.constrained(P64)
    errs = {"value variance": relmax(vv, (var64 + noise)[::p + 1]), "mean": relmax(mean, mu64),
            "nll": abs((-torch.distributions.Normal(mean[::p + 1].cpu().double(), vv.cpu().double().sqrt()).log_prob(yb[::p + 1].cpu().double()).mean()
                        + torch.distributions.Normal(mu64[::p + 1], (var64 + noise)[::p + 1].sqrt()).log_prob(yb[::p + 1].cpu().double()).mean()).item())}
    _report("reporting step, value rows from the fast path vs float64 oracle (one-call step: %s)" % c_step, errs)
    assert errs["value variance"] < 2e-5 and errs["mean"] < 4e-5 and errs["nll"] < 1e-6, errs          # observed 1.4e-6 / 4.0e-6 / 3.6e-8
    # a step whose FULL variance vector is read still takes the per-output path and agrees with its own stride
    loss, output, yb = loop.step(idx, need_variance=True)
    assert torch.equal(output.value_variance, output.variance[::p + 1])
