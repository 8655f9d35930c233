"""NaturalVariationalDistribution + NGD (reference directional_vi.py:35-37,164-167,186-191, ``use_ngd=True``).

gpytorch 1.4.0 holds the arithmetic (un-vendored, parity unpinned): the oracle restates ``_NaturalToMuVarSqrt`` and
``optim.NGD``; the CPU tests hold the restatement to its defining identities, the GPU tests hold the HIP engine to the
oracle.  Tolerance (fp32 model, fp64 factorisations): loss 2e-5 relative, natural-gradient blocks 5e-3 of the max."""
import pytest
import torch

import dsvgp_oracle as O


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def make_ngd_problem(N, d, M, p, B, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(N, d, generator=g)
    Y = O.testfun(X)
    V = torch.eye(d)[:p].repeat(M, 1) + 0.1 * torch.randn(M * p, d, generator=g)
    P = O.init_natural_params(X[:M].clone(), V, dtype, mean_init_std=0.2, generator=g)
    Mp = M * (p + 1)
    R = 0.15 * torch.randn(Mp, Mp, generator=g)
    P["natural_mat"] = (-0.5 * (torch.eye(Mp) + R @ R.t())).to(dtype)      # generic SPD precision
    P["constant"] = torch.tensor([0.1], dtype=dtype)
    P["raw_outputscale"] = torch.tensor(0.2, dtype=dtype)
    P["raw_lengthscale"] = torch.tensor([[0.3]], dtype=dtype)
    P["raw_noise"] = torch.tensor([-0.5], dtype=dtype)
    cols = sorted([0] + (torch.randperm(d, generator=g)[:p] + 1).tolist())
    x = X[M:M + B].to(dtype).contiguous()
    y = Y[M:M + B][:, cols].reshape(-1).to(dtype).contiguous()
    D = torch.eye(d, dtype=dtype)[[c - 1 for c in cols[1:]]].repeat(B, 1)
    return P, x, y, D, (d + 1) * N


# ------------------------------------------------------------------ oracle (CPU)
def test_natural_to_mu_chol_identities():
    P, *_ = make_ngd_problem(200, 3, 6, 2, 10, dtype=torch.float64)
    mu, LS = O.natural_to_mu_chol(P["natural_vec"], P["natural_mat"])
    S = LS @ LS.t()
    prec = -2.0 * P["natural_mat"]
    assert (S @ prec - torch.eye(S.shape[0], dtype=torch.float64)).abs().max() < 1e-10       # S = (-2 theta_2)^-1
    assert (prec @ mu - P["natural_vec"]).abs().max() < 1e-10                               # theta_1 = S^-1 mu
    # the prior initialisation is q(u) = N(noise, I)
    P0 = O.init_natural_params(torch.rand(4, 3), torch.eye(3)[:2].repeat(4, 1), torch.float64, 0.0)
    mu0, LS0 = O.natural_to_mu_chol(P0["natural_vec"], P0["natural_mat"])
    assert mu0.abs().max() == 0 and (LS0 - torch.eye(12, dtype=torch.float64)).abs().max() < 1e-14


def test_expectation_gradients_match_closed_form():
    """autograd through (eta_1, eta_2) == gpytorch's closed form: dS via the Cholesky backward, d eta_1 = dm - 2 dS m."""
    P, x, y, D, nd = make_ngd_problem(300, 3, 8, 2, 20, dtype=torch.float64)
    l, g, _, _ = O.ngd_loss_and_grads(P, x, y, D, nd)
    m, LS = O.natural_to_mu_chol(P["natural_vec"], P["natural_mat"])
    Pc = {k: v for k, v in P.items() if not k.startswith("natural_")}
    Pc["variational_mean"], Pc["chol_variational_covar"] = m, LS
    l2, g2, _, _ = O.elbo_loss_and_grads(Pc, x, y, D, nd)
    dL, dm = torch.tril(g2["chol_variational_covar"]), g2["variational_mean"]
    Phi = torch.tril(LS.t() @ dL)
    Phi.diagonal().mul_(0.5)
    Li = torch.linalg.inv(LS)
    dS = 0.5 * Li.t() @ (Phi + Phi.t()) @ Li
    assert abs(l.item() - l2.item()) < 1e-12
    assert (dS - g["natural_mat"]).abs().max() < 1e-12 and ((dm - 2 * dS @ m) - g["natural_vec"]).abs().max() < 1e-12
    for k in ("inducing_points", "raw_lengthscale", "raw_noise", "constant"):
        assert (g[k] - g2[k]).abs().max() < 1e-12


def test_ngd_steps_decrease_the_loss_and_keep_precision_spd():
    P, x, y, D, nd = make_ngd_problem(300, 2, 6, 1, 40, dtype=torch.float64)
    losses = []
    for _ in range(5):
        l, g, _, _ = O.ngd_loss_and_grads(P, x, y, D, nd)
        losses.append(l.item())
        O.ngd_step(P, g, nd, lr=0.1)
    assert losses[-1] < losses[0]
    assert torch.linalg.eigvalsh(-2.0 * P["natural_mat"]).min() > 0


# ------------------------------------------------------------------ HIP engine vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("N,d,M,p,B", [(400, 2, 20, 2, 200), (500, 20, 30, 5, 96), (300, 6, 40, 0, 64)])
@pytest.mark.parametrize("nb", [64, 4096])
def test_natural_step_matches_oracle(dsvgp, gpu_device, N, d, M, p, B, nb):
    P, x, y, D, nd = make_ngd_problem(N, d, M, p, B, seed=N + d)
    l_ref, g_ref, mu_ref, var_ref = O.ngd_loss_and_grads(P, x, y, D, nd)
    eng = dsvgp.ElboEngine(gpu_device, trsm_nb=nb)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd)
    assert set(grads) == set(O.NGD_PARAM_NAMES)
    assert abs(loss.item() - l_ref.item()) < 2e-5 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 2e-4
    for k in O.NGD_PARAM_NAMES:
        if g_ref[k].numel() and g_ref[k].abs().max() > 0:
            assert relmax(grads[k], g_ref[k]) < 5e-3, k
    # prediction from natural parameters
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 2e-4 and relmax(varn2, var_ref) < 2e-4


@pytest.mark.gpu
def test_bad_precision_raises(dsvgp, gpu_device):
    P, x, y, D, nd = make_ngd_problem(200, 2, 6, 1, 20)
    P["natural_mat"] = P["natural_mat"].neg()            # precision -2 theta_2 negative definite
    eng = dsvgp.ElboEngine(gpu_device)
    with pytest.raises(dsvgp.NotPSDError):
        eng.loss_and_grads({k: v.to(gpu_device) for k, v in P.items()}, x.to(gpu_device), y.to(gpu_device),
                           D.to(gpu_device), nd)


@pytest.mark.gpu
def test_train_gp_use_ngd_drop_in(dsvgp, gpu_device, capsys):
    """reference tests/test_dsvgp.py sizes with use_ngd=True (directional_vi.py:164-167)."""
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, p = 600, 2, 2
    train_x = torch.rand(n, dim)
    train_y = O.testfun(train_x)
    model, likelihood = dsvgp.train_gp(TensorDataset(train_x, train_y), num_inducing=20, num_directions=p,
                                       minibatch_size=200, minibatch_dim=p, num_epochs=60, use_ngd=True,
                                       learning_rate_ngd=0.1, tqdm=False, seed=3)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 3 and losses[-1] < losses[0]
    sd = model.state_dict()
    assert "variational_strategy._variational_distribution.natural_vec" in sd
    assert "variational_strategy._variational_distribution.natural_mat" in sd
    assert "variational_strategy._variational_distribution.variational_mean" not in sd
    # the natural parameters moved and still define an SPD precision; one step of the trained model matches the oracle
    nat = model.variational_strategy._variational_distribution.natural_mat.detach().cpu().double()
    assert torch.linalg.eigvalsh(-2.0 * 0.5 * (nat + nat.t())).min() > 0
    means, variances = dsvgp.eval_gp(TensorDataset(train_x[:100], train_y[:100]), model, likelihood,
                                     num_directions=p, minibatch_size=50, minibatch_dim=p)
    assert means.shape == (300,) and (variances > 0).all()
    Pm = {k: v.detach().cpu() for k, v in model._param_dict(likelihood).items()}
    x, y = train_x[:64], train_y[:64][:, [0, 1, 2]].reshape(-1)
    D = torch.eye(dim).repeat(64, 1)
    l_ref, g_ref, _, _ = O.ngd_loss_and_grads(Pm, x, y, D, (dim + 1) * n)
    eng = dsvgp.ElboEngine(gpu_device)
    loss, grads, _, _ = eng.loss_and_grads({k: v.to(gpu_device) for k, v in Pm.items()}, x.to(gpu_device),
                                           y.to(gpu_device), D.to(gpu_device), (dim + 1) * n)
    assert abs(loss.item() - l_ref.item()) < 2e-4 * abs(l_ref.item())
    assert relmax(grads["natural_vec"], g_ref["natural_vec"]) < 1e-2
    assert relmax(grads["natural_mat"], g_ref["natural_mat"]) < 1e-2


@pytest.mark.gpu
def test_grad_svgp_use_ngd(dsvgp, gpu_device, capsys):
    """reference grad_svgp.py:21-22,66-67,87-88: the full-gradient SVGP with natural-gradient q(u) updates."""
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim = 400, 2
    train_x = torch.rand(n, dim)
    train_y = O.testfun(train_x)
    G = dsvgp.grad_svgp
    model, likelihood = G.train_gp(TensorDataset(train_x, train_y), dim, num_inducing=16, minibatch_size=100,
                                   num_epochs=40, use_ngd=True, learning_rate_ngd=0.1, tqdm=False, seed=1)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 3 and losses[-1] < losses[0]
    assert "variational_strategy._variational_distribution.natural_mat" in model.state_dict()
    means, variances = G.eval_gp(TensorDataset(train_x[:40], train_y[:40]), model, likelihood, minibatch_size=20)
    assert means.shape == (120,) and (variances > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mll", ["ELBO", "PLL"])
def test_shared_directions_with_natural_distribution(dsvgp, gpu_device, mll):
    """shared inducing directions + NaturalVariationalDistribution over the M + p shared values
    (reference shared_directional_vi.py:37-39,170-171,188-189 with SharedDirectionalGradVariationalStrategy.py:95-107)"""
    N, d, M, p, B = 400, 4, 14, 2, 80
    P, x, y, D, nd = make_ngd_problem(N, d, M, p, B, seed=33)
    g = torch.Generator().manual_seed(7)
    P["inducing_directions"] = torch.eye(d)[:p] + 0.2 * torch.randn(p, d, generator=g)        # ONE shared set
    P["natural_vec"] = 0.3 * torch.randn(M + p, generator=g)
    R = 0.15 * torch.randn(M + p, M + p, generator=g)
    P["natural_mat"] = -0.5 * (torch.eye(M + p) + R @ R.t())
    l_ref, g_ref, mu_ref, var_ref = O.ngd_loss_and_grads(P, x, y, D, nd, mll, forward=O.shared_forward)
    eng = dsvgp.ElboEngine(gpu_device)
    eng.shared_directions = True
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll)
    assert set(grads) == set(O.NGD_PARAM_NAMES)
    assert abs(loss.item() - l_ref.item()) < 2e-5 * abs(l_ref.item())
    assert relmax(mu, mu_ref) < 2e-4 and relmax(varn, var_ref) < 2e-4
    for k in O.NGD_PARAM_NAMES:
        assert grads[k].shape == g_ref[k].shape, k
        if g_ref[k].abs().max() > 0:
            assert relmax(grads[k], g_ref[k]) < 5e-3, k
    mu2, varn2 = eng.predict(Pg, x.to(gpu_device), D.to(gpu_device))
    assert relmax(mu2, mu_ref) < 2e-4 and relmax(varn2, var_ref) < 2e-4


@pytest.mark.gpu
def test_shared_train_gp_with_ngd(dsvgp, gpu_device, capsys):
    from torch.utils.data import TensorDataset
    torch.manual_seed(0)
    n, dim, p = 600, 2, 2
    train_x = torch.rand(n, dim)
    train_y = O.testfun(train_x)
    S = dsvgp.shared_directional_vi
    model, likelihood = S.train_gp(TensorDataset(train_x, train_y), num_inducing=20, num_directions=p, minibatch_size=200,
                                   minibatch_dim=p, num_epochs=40, inducing_data_initialization=False, use_ngd=True, seed=2,
                                   learning_rate_ngd=2e-4)     # (the shared strategy's zero middle term leaves S at the prior: the
                                                                #  natural step on the mean is an un-preconditioned full-data step)
    out = capsys.readouterr().out
    losses = [float(l.split("loss: ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) >= 2 and losses[-1] < losses[0]
    sd = model.state_dict()
    assert sd["variational_strategy._variational_distribution.natural_vec"].shape == (20 + p,)
    assert sd["variational_strategy._variational_distribution.natural_mat"].shape == (20 + p, 20 + p)
    with pytest.raises(NotImplementedError):              # the reference's own shared + CIQ call is shape-inconsistent
        S.train_gp(TensorDataset(train_x, train_y), num_inducing=20, num_directions=p, minibatch_size=200, minibatch_dim=p,
                   num_epochs=1, inducing_data_initialization=False, use_ciq=True, verbose=False)
