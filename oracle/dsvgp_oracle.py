"""CPU oracle for the DSVGP minibatch-ELBO hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (CPU) restatement of what the reference computes for one
training step of ``directional_vi.train_gp`` -- it is the *checker* for the HIP path and
the ``cpu_baseline`` leg of ``bench.py``.  Nothing under ``gp-derivatives-variational-inference_amd/``
may import it; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` do.

Parity status
-------------
* Rows a1/a2 (``RBFKernelDirectionalGrad.forward``) are PINNED: ``tests/golden/kernel_*.npz`` were
  produced by importing the reference kernel file itself (``oracle/make_golden.py``) and
  ``tests/test_oracle_golden.py`` checks :func:`kernel_matrix` / :func:`kernel_diag` against them.
* Rows a3-a8 (ScaleKernel, constraints, jitter, Cholesky/solve composition, variational
  distribution, Gaussian likelihood, ELBO, KL) live in the un-vendored third-party dependency
  ``gpytorch==1.4.0`` (reference ``graphite_environment.yml:101``), which is not installable here.
  They are restated from its published algorithm and anchored on the reference call sites cited
  below.  The reference's own tests hold no golden values for them: **parity unpinned** for those rows.

All functions work in whatever dtype their inputs have; :func:`elbo_forward` additionally
implements the reference's mixed precision (model dtype + fp64 Cholesky / triangular solves).
"""
import math

import torch
import torch.nn.functional as F

KZZ_JITTER = 1e-3      # LazyTensor.add_jitter() default, reference DGVS.py:144
KXX_JITTER = 1e-4      # data_data_covar.add_jitter(1e-4), reference DGVS.py:197,203
NOISE_FLOOR = 1e-4     # GaussianLikelihood noise constraint GreaterThan(1e-4) (gpytorch 1.4.0)
CHOL_JITTER = 1e-6     # settings.cholesky_jitter.value() (float default), reference DGVS.py:74
CHOL_TRIES = 3         # psd_safe_cholesky max_tries (gpytorch 1.4.0)
MIN_VARIANCE = 1e-6    # MultivariateNormal.variance clamp (gpytorch settings.min_variance, float)


# --------------------------------------------------------------------------------------
# a1 / a2: RBFKernelDirectionalGrad
# --------------------------------------------------------------------------------------
def normalize_rows(v):
    """reference RBFKernelDirectionalGrad.py:57-58 -- directions are L2-normalised per call."""
    return v / v.norm(dim=1, keepdim=True)


def kernel_matrix(x1, x2, v1, v2, lengthscale):
    """Dense interleaved block kernel, reference RBFKernelDirectionalGrad.py:41-108.

    Row ``i*(p+1)`` is the function value at ``x1[i]``, rows ``i*(p+1)+1+a`` the derivative along
    the a-th (normalised) direction of point i; columns likewise for ``x2``/``v2``.
    Written pair-wise (difference form) instead of the reference's block-assemble-then-shuffle.
    """
    n1, d = x1.shape
    n2 = x2.shape[0]
    p = v1.shape[0] // n1
    assert v2.shape[0] // n2 == p, "v1 and v2 must contain same number of directions"
    ell = lengthscale.reshape(()) if torch.is_tensor(lengthscale) else lengthscale
    V1 = normalize_rows(v1).reshape(n1, p, d)
    V2 = normalize_rows(v2).reshape(n2, p, d)
    r = (x1[:, None, :] - x2[None, :, :]) / ell                    # :67-68
    k = torch.exp(-0.5 * (r * r).sum(-1))                          # :71-73
    u = torch.einsum("ijd,iad->ija", r, V1)                        # r . v1_{i,a}
    w = torch.einsum("ijd,jbd->ijb", r, V2)                        # r . v2_{j,b}
    G = torch.einsum("iad,jbd->iajb", V1, V2)
    K = x1.new_zeros(n1, p + 1, n2, p + 1)
    K[:, 0, :, 0] = k
    K[:, 0, :, 1:] = w * k[..., None] / ell                        # :77-83
    K[:, 1:, :, 0] = (-u * k[..., None] / ell).permute(0, 2, 1)    # :86-93
    uw = u.permute(0, 2, 1)[:, :, :, None] * w[:, None, :, :]      # [i,a,j,b]
    K[:, 1:, :, 1:] = (G - uw) * k[:, None, :, None] / ell ** 2    # :97-102
    return K.reshape(n1 * (p + 1), n2 * (p + 1))                   # interleaved, :105-107


def _rbf_sq_dist(a, b):
    """gpytorch 1.4.0 ``Kernel.covar_dist(square_dist=True, dist_postprocess_func=postprocess_rbf)`` (un-vendored; reached
    from reference RBFKernelDirectionalGrad.py:71): both point sets shifted by the mean of the first, the squared
    distance from ONE matmul of the padded operands [-2a, |a|^2, 1] [b, 1, |b|^2]^T, clamped at 0, then exp(-r/2)."""
    shift = a.mean(-2, keepdim=True)
    a, b = a - shift, b - shift
    an, bn = (a * a).sum(-1, keepdim=True), (b * b).sum(-1, keepdim=True)
    lhs = torch.cat([-2.0 * a, an, torch.ones_like(an)], dim=-1)
    rhs = torch.cat([b, torch.ones_like(bn), bn], dim=-1)
    return (lhs @ rhs.t()).clamp_min(0).div(-2).exp()


def kernel_matrix_refseq(x1, x2, v1, v2, lengthscale):
    """The SAME matrix as :func:`kernel_matrix`, evaluated with the reference's op structure
    (RBFKernelDirectionalGrad.py:57-107) instead of pair-wise differences: blocked layout
    [[values, right derivatives], [left derivatives, Hessian]] filled through matmul / bmm projections of the scaled
    points on the directions, column-permutation gathers, ``repeat`` broadcasts of the value block, and one final
    perfect-shuffle gather of rows and columns into the interleaved order.  Used by the ``cpu_baseline`` leg of bench.py
    (what the reference costs on a CPU) and pinned to :func:`kernel_matrix` / the golden vectors in tests."""
    n1, d = x1.shape
    n2 = x2.shape[0]
    p = v1.shape[0] // n1
    assert v2.shape[0] // n2 == p, "v1 and v2 must contain same number of directions"
    ell = lengthscale.reshape(1, 1) if torch.is_tensor(lengthscale) else torch.tensor([[lengthscale]], dtype=x1.dtype)
    u1 = (v1.t() / v1.norm(dim=1)).t()                                  # :57-58
    u2 = (v2.t() / v2.norm(dim=1)).t()
    K = torch.zeros(n1 * (p + 1), n2 * (p + 1), dtype=x1.dtype)         # :61-62 (the reference allocates it twice)
    K = torch.zeros(n1 * (p + 1), n2 * (p + 1), dtype=x1.dtype)
    a, b = x1 / ell, x2 / ell                                           # :67-68
    kv = _rbf_sq_dist(a, b)                                             # :71-73
    K[:n1, :n2] = kv
    if p > 0:
        # right-derivative block (:77-83): (a_i - b_j) . u2_{j,t} from a dense product minus the per-point self terms
        b_u2 = torch.bmm(b.reshape(n2, 1, d), u2.reshape(n2, p, d).transpose(-2, -1))
        right = a @ u2.t() - b_u2.flatten()
        by_dir_c = torch.arange(n2 * p).view(n2, p).t().reshape(n2 * p)         # columns grouped by direction index
        right = right[:, by_dir_c] / ell
        K[:n1, n2:] = right * kv.repeat(1, p)
        # left-derivative block (:86-93)
        a_u1 = torch.bmm(a.reshape(n1, 1, d), u1.reshape(n1, p, d).transpose(-2, -1))
        left = a_u1.flatten() - b @ u1.t()
        by_dir_r = torch.arange(n1 * p).view(n1, p).t().reshape(n1 * p)
        left = left[:, by_dir_r].t() / ell
        K[n1:, :n2] = -left * kv.repeat(p, 1)
        # Hessian block (:97-102)
        cross = right.repeat(p, 1) * left.repeat(1, p)
        gram = (u1 @ u2.t() / ell.pow(2))[:, by_dir_c][by_dir_r, :]
        K[n1:, n2:] = (gram - cross) * kv.repeat(p, p)
    rows = torch.arange(n1 * (p + 1)).view(p + 1, n1).t().reshape(n1 * (p + 1))   # perfect shuffle, :105-107
    cols = torch.arange(n2 * (p + 1)).view(p + 1, n2).t().reshape(n2 * (p + 1))
    return K[rows, :][:, cols]


def kernel_diag(n, p, lengthscale, dtype=torch.float64):
    """diag=True branch, reference RBFKernelDirectionalGrad.py:110-119."""
    ell = lengthscale.reshape(()) if torch.is_tensor(lengthscale) else torch.tensor(lengthscale, dtype=dtype)
    dg = torch.ones(n, p + 1, dtype=ell.dtype if torch.is_tensor(ell) else dtype)
    dg = dg * torch.cat([torch.ones(1, dtype=dg.dtype), (1.0 / ell ** 2).expand(p).to(dg.dtype)])
    return dg.reshape(n * (p + 1))


# --------------------------------------------------------------------------------------
# a3-a8: one ELBO evaluation
# --------------------------------------------------------------------------------------
def constrained(params):
    """softplus constraints (gpytorch Positive / GreaterThan(1e-4)); raw inits are 0."""
    ell = F.softplus(params["raw_lengthscale"]).reshape(())
    s = F.softplus(params["raw_outputscale"]).reshape(())
    noise = F.softplus(params["raw_noise"]).reshape(()) + NOISE_FLOOR
    return ell, s, noise


def psd_safe_cholesky(K):
    """gpytorch.utils.cholesky.psd_safe_cholesky (1.4.0): retry with jitter*10^i, i=0..2."""
    L, info = torch.linalg.cholesky_ex(K)
    if int(info) == 0:
        return L
    prev = 0.0
    Kp = K.clone()
    for i in range(CHOL_TRIES):
        new = CHOL_JITTER * (10 ** i)
        Kp.diagonal().add_(new - prev)
        prev = new
        L, info = torch.linalg.cholesky_ex(Kp)
        if int(info) == 0:
            return L
    raise RuntimeError("Matrix not positive definite after repeatedly adding jitter up to %g" % prev)


def predictive(params, x, D, solve_dtype=torch.float64, data_outputs="all", assembly=None):
    """DirectionalGradVariationalStrategy.forward, reference DGVS.py:89-208.

    Returns (mu, var) with ``var = diag(Sigma)`` of q(f) (no likelihood noise).
    ``data_outputs="values"``: the derivative-free-data variant (reference DFreeDirectionalGradVariationalStrategy.py:
    113-136): the inducing side keeps its p directional derivatives, the data side only function values, i.e. every
    (p+1)-th column / row of the same kernel blocks."""
    Z, V = params["inducing_points"], params["inducing_directions"]
    m = params["variational_mean"]
    L_S = torch.tril(params["chol_variational_covar"])            # CholeskyVariationalDistribution mask
    c = params["constant"].reshape(())
    ell, s, _ = constrained(params)
    M, d = Z.shape
    p = V.shape[0] // M
    B = x.shape[0]
    assert D.shape[0] // B == p, "Need minibatch dim to be same as number of directions for kernel"
    dt = x.dtype
    kernel_matrix = assembly or globals()["kernel_matrix"]         # (``kernel_matrix_refseq``: the reference's op sequence)
    K_ZX = s * kernel_matrix(Z, x, V, D, ell)                      # :128-132
    K_XZ = s * kernel_matrix(x, Z, D, V, ell)                      # :133-137
    K_ZZ = s * kernel_matrix(Z, Z, V, V, ell)
    K_ZZ = K_ZZ + KZZ_JITTER * torch.eye(K_ZZ.shape[0], dtype=dt)  # :140-144
    dg = s * kernel_diag(B, p, ell).to(dt)                         # :145-149 (diag only)
    if data_outputs == "values":                                   # DFree :119,124,136
        K_ZX, K_XZ, dg = K_ZX[:, ::p + 1], K_XZ[::p + 1, :], dg[::p + 1]
    L = psd_safe_cholesky(K_ZZ.to(solve_dtype))                    # :72-75,172
    A = torch.linalg.solve_triangular(L, K_ZX.to(solve_dtype), upper=False).to(dt)          # :181
    A_t = torch.linalg.solve_triangular(L, K_XZ.t().to(solve_dtype), upper=False).to(dt)    # :183
    mu = A_t.t() @ m + c                                           # :126,188
    SA = L_S @ (L_S.t() @ A) - A                                   # (S - I) A, :192-194
    var = dg + KXX_JITTER + (A_t * SA).sum(0)                      # diag of :202-205
    return mu, var


def predictive_joint(params, x, D, solve_dtype=torch.float64, data_outputs="all"):
    """Mean and FULL covariance of q(f) at the batch (no likelihood noise): the MultivariateNormal that
    ``model(x, derivative_directions=D)`` returns in eval mode, reference DGVS.py:199-208
    (``data_data_covar.add_jitter(1e-4) + A_t^T (S - I) A``); ``likelihood(.)`` adds ``noise * I``.  The BO drivers draw
    joint samples from it (experiments/GNN_bo/gcn_turbo.py:238-239)."""
    Z, V = params["inducing_points"], params["inducing_directions"]
    m = params["variational_mean"]
    L_S = torch.tril(params["chol_variational_covar"])
    c = params["constant"].reshape(())
    ell, s, _ = constrained(params)
    M = Z.shape[0]
    p = V.shape[0] // M
    dt = x.dtype
    K_ZX = s * kernel_matrix(Z, x, V, D, ell)
    K_XX = s * kernel_matrix(x, x, D, D, ell)
    K_ZZ = s * kernel_matrix(Z, Z, V, V, ell)
    K_ZZ = K_ZZ + KZZ_JITTER * torch.eye(K_ZZ.shape[0], dtype=dt)
    if data_outputs == "values":
        K_ZX, K_XX = K_ZX[:, ::p + 1], K_XX[::p + 1, ::p + 1]
    L = psd_safe_cholesky(K_ZZ.to(solve_dtype))
    A = torch.linalg.solve_triangular(L, K_ZX.to(solve_dtype), upper=False).to(dt)
    mu = A.t() @ m + c
    W = L_S.t() @ A
    Sigma = K_XX + KXX_JITTER * torch.eye(K_XX.shape[0], dtype=dt) + W.t() @ W - A.t() @ A
    return mu, Sigma


def kl_whitened(m, L_S):
    """KL(q(u) || N(0, I)); gpytorch kl_mvn_mvn with the whitened prior of DGVS.py:77-87."""
    Mp = m.shape[0]
    return 0.5 * ((m * m).sum() + (L_S * L_S).sum() - Mp - torch.log(torch.diagonal(L_S) ** 2).sum())


def elbo_forward(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, solve_dtype=torch.float64,
                 data_outputs="all", assembly=None):
    """loss = -mll(likelihood(model(x)), y), reference directional_vi.py:245-246.

    ``y`` is the interleaved target vector of length B*(p+1) (:241).  ``global_rows`` lets a
    data-parallel rank normalise by the global batch (defaults to the local one).
    Note the reference feeds the *noised* marginal into the mll, so the noise enters twice.
    """
    mu, var = predictive(params, x, D, solve_dtype, data_outputs, assembly)
    _, _, noise = constrained(params)
    L_S = torch.tril(params["chol_variational_covar"])
    Bp = y.shape[0] if global_rows is None else global_rows
    varn = (var + noise).clamp_min(MIN_VARIANCE)                   # likelihood(model(x)).variance
    if mll_type == "ELBO":                                         # GaussianLikelihood.expected_log_prob
        ll = -0.5 * (((y - mu) ** 2 + varn) / noise + torch.log(noise) + math.log(2 * math.pi))
    elif mll_type == "PLL":                                        # GaussianLikelihood.log_marginal
        tot = (varn + noise).clamp_min(1e-8)
        ll = -0.5 * ((y - mu) ** 2 / tot + torch.log(tot) + math.log(2 * math.pi))
    else:
        raise ValueError(mll_type)
    kl = kl_whitened(params["variational_mean"], L_S)
    loss = -(ll.sum() / Bp - kl / num_data)
    return loss, mu, varn


PARAM_NAMES = ("inducing_points", "inducing_directions", "variational_mean", "chol_variational_covar",
               "constant", "raw_outputscale", "raw_lengthscale", "raw_noise")


def elbo_loss_and_grads(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, solve_dtype=torch.float64,
                        data_outputs="all"):
    """Forward + autograd backward, exactly how the reference gets its gradients (:249)."""
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss, mu, varn = elbo_forward(ps, x, y, D, num_data, mll_type, global_rows, solve_dtype, data_outputs)
    loss.backward()
    grads = {k: (ps[k].grad if ps[k].grad is not None else torch.zeros_like(ps[k])) for k in ps}
    return loss.detach(), grads, mu.detach(), varn.detach()


def init_params(Z, V, dtype=torch.float32, mean_init_std=0.0, generator=None):
    """Initial parameter set of GPModel / GaussianLikelihood (reference directional_vi.py:25-56,172)."""
    M = Z.shape[0]
    p = V.shape[0] // M
    Mp = M * (p + 1)
    m = torch.zeros(Mp, dtype=dtype)
    if mean_init_std:
        m = m + mean_init_std * torch.randn(Mp, dtype=dtype, generator=generator)
    return {
        "inducing_points": Z.to(dtype).clone(),
        "inducing_directions": V.to(dtype).clone(),
        "variational_mean": m,
        "chol_variational_covar": torch.eye(Mp, dtype=dtype),
        "constant": torch.zeros(1, dtype=dtype),
        "raw_outputscale": torch.zeros((), dtype=dtype),
        "raw_lengthscale": torch.zeros(1, 1, dtype=dtype),
        "raw_noise": torch.zeros(1, dtype=dtype),
    }


# --------------------------------------------------------------------------------------
# shared inducing directions (reference SharedDirectionalGradVariationalStrategy.py:89-212,
# shared_directional_vi.py:25-63): ONE set of p directions for all inducing points, q(u) over M function values +
# p shared derivative values, and -- as the reference computes it -- a ZERO middle term (:210-212), so the predictive
# covariance is K_XX + 1e-4 I and q(u)'s covariance enters the objective through the KL term only.
# --------------------------------------------------------------------------------------
def shared_expand(V_s, m_s, M):
    """(:95-107) tile the p shared directions over the M inducing points and interleave the values:
    row i(p+1) <- m_s[i], rows i(p+1)+1.. <- the p shared derivative values m_s[M:]."""
    p = V_s.shape[0]
    V = V_s.repeat(M, 1)
    iv = torch.cat([m_s[:M].reshape(M, 1), m_s[M:].reshape(1, p).expand(M, p)], dim=1).reshape(-1)
    return V, iv


def shared_forward(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, solve_dtype=torch.float64):
    Z, V_s = params["inducing_points"], params["inducing_directions"]
    m_s = params["variational_mean"]
    L_S = torch.tril(params["chol_variational_covar"])
    c = params["constant"].reshape(())
    ell, s, noise = constrained(params)
    M = Z.shape[0]
    p = V_s.shape[0]
    B = x.shape[0]
    assert D.shape[0] // B == p, "Need minibatch dim to be same as number of directions for kernel"
    dt = x.dtype
    V, iv = shared_expand(V_s, m_s, M)
    K_ZX = s * kernel_matrix(Z, x, V, D, ell)                                   # :147-151 (K_XZ is its transpose)
    K_ZZ = s * kernel_matrix(Z, Z, V, V, ell) + KZZ_JITTER * torch.eye(M * (p + 1), dtype=dt)   # :158-162
    dg = s * kernel_diag(B, p, ell).to(dt)
    L = psd_safe_cholesky(K_ZZ.to(solve_dtype))
    A = torch.linalg.solve_triangular(L, K_ZX.to(solve_dtype), upper=False).to(dt)              # :194-196
    mu = A.t() @ iv + c                                                         # :201
    var = dg + KXX_JITTER                                                       # :210-226 with the zero middle term
    Bp = y.shape[0] if global_rows is None else global_rows
    varn = (var + noise).clamp_min(MIN_VARIANCE)
    if mll_type == "ELBO":
        ll = -0.5 * (((y - mu) ** 2 + varn) / noise + torch.log(noise) + math.log(2 * math.pi))
    else:
        tot = (varn + noise).clamp_min(1e-8)
        ll = -0.5 * ((y - mu) ** 2 / tot + torch.log(tot) + math.log(2 * math.pi))
    kl = kl_whitened(m_s, L_S)                                                  # q(u) over M + p values
    return -(ll.sum() / Bp - kl / num_data), mu, varn


def shared_loss_and_grads(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, solve_dtype=torch.float64):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss, mu, varn = shared_forward(ps, x, y, D, num_data, mll_type, global_rows, solve_dtype)
    loss.backward()
    grads = {k: (ps[k].grad if ps[k].grad is not None else torch.zeros_like(ps[k])) for k in ps}
    return loss.detach(), grads, mu.detach(), varn.detach()


# --------------------------------------------------------------------------------------
# SURVEY 8f rank 3, first half: NaturalVariationalDistribution + gpytorch.optim.NGD
# (reference directional_vi.py:35-37,164-167,186-191: ``use_ngd=True`` keeps the Cholesky-whitened
# strategy and swaps q(u)'s parameterisation and its optimizer).  gpytorch 1.4.0, un-vendored: restated from
# its published algorithm (Salimbeni et al. 2018, natural gradients in practice) -- parity unpinned.
# --------------------------------------------------------------------------------------
NGD_PARAM_NAMES = ("inducing_points", "inducing_directions", "natural_vec", "natural_mat",
                   "constant", "raw_outputscale", "raw_lengthscale", "raw_noise")


def natural_to_mu_chol(natural_vec, natural_mat):
    """gpytorch ``_NaturalToMuVarSqrt.forward``: precision P = -2 theta_2 = L_P L_P^T, S = P^-1 = L_P^-T L_P^-1,
    mu = S theta_1, and the Cholesky factor of S that CholLazyTensor wraps."""
    P = -2.0 * natural_mat
    L_P = torch.linalg.cholesky(0.5 * (P + P.t()))
    X = torch.linalg.solve_triangular(L_P, torch.eye(P.shape[0], dtype=P.dtype), upper=False)
    S = X.t() @ X
    mu = S @ natural_vec
    return mu, torch.linalg.cholesky(0.5 * (S + S.t()))


def init_natural_params(Z, V, dtype=torch.float32, mean_init_std=0.0, generator=None):
    """NaturalVariationalDistribution.initialize_variational_distribution(N(0, I)): theta_1 = noise, theta_2 = -I/2."""
    P = init_params(Z, V, dtype, 0.0, generator)
    Mp = P.pop("variational_mean").shape[0]
    P.pop("chol_variational_covar")
    nv = torch.zeros(Mp, dtype=dtype)
    if mean_init_std:
        nv = nv + mean_init_std * torch.randn(Mp, dtype=dtype, generator=generator)
    P["natural_vec"] = nv
    P["natural_mat"] = -0.5 * torch.eye(Mp, dtype=dtype)
    return P


def ngd_loss_and_grads(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, solve_dtype=torch.float64,
                       forward=None):
    """(``forward``: ``elbo_forward`` by default; ``shared_forward`` for the shared-directions strategy with a natural q(u),
    reference shared_directional_vi.py:37-39,170-171.)
    One step's loss and the gradients gpytorch hands to its optimizers when q(u) is a
    NaturalVariationalDistribution: ordinary gradients for the hyper-parameters, and for (theta_1, theta_2) the
    gradients with respect to the EXPECTATION parameters eta_1 = mu, eta_2 = S + mu mu^T
    (``_NaturalToMuVarSqrt.backward``) -- stepping theta along them is natural gradient descent."""
    mu0, LS0 = natural_to_mu_chol(params["natural_vec"].detach().double(), params["natural_mat"].detach().double())
    dt = params["natural_vec"].dtype
    eta1 = mu0.clone().requires_grad_(True)
    eta2 = (LS0 @ LS0.t() + torch.outer(mu0, mu0)).requires_grad_(True)
    S = eta2 - torch.outer(eta1, eta1)
    L_S = torch.linalg.cholesky(0.5 * (S + S.t()))
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items() if not k.startswith("natural_")}
    ps["variational_mean"] = eta1.to(dt)
    ps["chol_variational_covar"] = L_S.to(dt)
    loss, mu, varn = (forward or elbo_forward)(ps, x, y, D, num_data, mll_type, global_rows, solve_dtype)
    loss.backward()
    grads = {k: (ps[k].grad if ps[k].grad is not None else torch.zeros_like(ps[k])) for k in ps
             if k not in ("variational_mean", "chol_variational_covar")}
    grads["natural_vec"] = eta1.grad.to(dt)
    g2 = eta2.grad
    grads["natural_mat"] = (0.5 * (g2 + g2.t())).to(dt)
    return loss.detach(), grads, mu.detach(), varn.detach()


def ngd_step(params, grads, num_data, lr=0.1):
    """gpytorch.optim.NGD.step: theta <- theta - lr * num_data * grad (in place)."""
    for k in ("natural_vec", "natural_mat"):
        params[k].add_(grads[k], alpha=-lr * num_data)


# --------------------------------------------------------------------------------------
# SURVEY 8f rank 3, second half: CIQ whitening (reference CiqDirectionalGradVariationalStrategy.py:197-295,
# ``use_ciq=True``): interp_term = K_ZZ^{-1/2} K_ZX by contour-integral quadrature + msMINRES, and the NGD
# interpolation terms of ``_NgdInterpTerms`` (:19-123).  The quadrature / msMINRES / custom backward live in
# gpytorch 1.4.0 (``utils/contour_integral_quad.py``, ``utils/minres.py``, ``functions/_sqrt_inv_matmul.py``),
# un-vendored: restated from the published algorithm (Pleiss et al. 2020; Hale, Higham & Trefethen 2008,
# method 3; Paige & Saunders MINRES with shifts) -- parity unpinned.
# --------------------------------------------------------------------------------------
NUM_CONTOUR_QUADRATURE = 15   # train_gp(num_contour_quadrature=15), reference directional_vi.py:101,165
MINRES_TOLERANCE = 1e-4       # gpytorch settings.minres_tolerance
MAX_MINRES_ITER = 1000        # gpytorch settings.max_cg_iterations
MAX_LANCZOS_ITER = 20         # contour_integral_quad(max_lanczos_iter=20)


def lanczos_eig_bounds(K, v0, iters=MAX_LANCZOS_ITER):
    """Extreme Ritz values of ``iters`` Lanczos steps started at v0 (what contour_integral_quad's
    linear_cg(n_tridiag=1) tridiagonal yields); falls back to diag(K) when they are not positive."""
    n = K.shape[0]
    iters = min(iters, n)
    q_prev = torch.zeros_like(v0)
    q = v0 / v0.norm()
    beta = 0.0
    alphas, betas = [], []
    for k in range(iters):
        w = K @ q - beta * q_prev
        alpha = torch.dot(q, w)
        w = w - alpha * q
        alphas.append(alpha)
        beta = w.norm()
        if k + 1 < iters:
            if float(beta) < 1e-12 * float(abs(alpha)):
                break
            betas.append(beta)
            q_prev, q = q, w / beta
    a = torch.stack(alphas)
    T = torch.diag(a)
    if len(alphas) > 1:
        b = torch.stack(betas[:len(alphas) - 1])
        T = T + torch.diag(b, 1) + torch.diag(b, -1)
    eigs = torch.linalg.eigvalsh(T)
    if eigs.min() <= 0:
        eigs = torch.diagonal(K)
    return float(eigs.min()), float(eigs.max())


def ciq_quadrature(lmin, lmax, Q=NUM_CONTOUR_QUADRATURE):
    """K^{-1/2} ~= sum_q omega_q (K + sigma_q I)^-1 on [lmin, lmax] (HHT method 3 through Jacobi elliptic functions,
    as contour_integral_quad does with scipy): sigma_q = lmin sn^2/cn^2, omega_q = 2 K' sqrt(lmin) dn / (pi Q cn^2)."""
    import numpy as np
    import scipy.special
    k2 = lmin / lmax
    Kp = scipy.special.ellipk(1.0 - k2)
    u = (np.arange(1, Q + 1) - 0.5) * Kp / Q
    sn, cn, dn, _ = scipy.special.ellipj(u, 1.0 - k2)
    sigma = lmin * (sn / cn) ** 2
    omega = 2.0 * Kp * math.sqrt(lmin) / (math.pi * Q) * dn / cn ** 2
    return torch.tensor(sigma, dtype=torch.float64), torch.tensor(omega, dtype=torch.float64)


def msminres(K, R, sigma, tol=MINRES_TOLERANCE, max_iter=MAX_MINRES_ITER):
    """Shifted MINRES: X[q] = (K + sigma_q I)^-1 R for all shifts from ONE Lanczos process per right-hand side
    (gpytorch.utils.minres).  Columns are normalised first; convergence = mean over (shift, column) of
    |last update| / |solution| < tol, tested every 10 iterations."""
    n, t = R.shape
    Q = sigma.shape[0]
    dt = R.dtype
    sig = sigma.to(dt).reshape(Q, 1)
    rnorm = R.norm(dim=0)
    rnorm = torch.where(rnorm < 1e-10, torch.ones_like(rnorm), rnorm)
    q_prev = torch.zeros_like(R)
    q = R / rnorm
    beta = torch.ones(t, dtype=dt)              # beta_1 of the normalised system
    cs = -torch.ones(Q, t, dtype=dt)
    sn = torch.zeros(Q, t, dtype=dt)
    dbar = torch.zeros(Q, t, dtype=dt)
    eps = torch.zeros(Q, t, dtype=dt)
    phibar = torch.ones(Q, t, dtype=dt)
    w1 = torch.zeros(Q, n, t, dtype=dt)
    w2 = torch.zeros(Q, n, t, dtype=dt)
    X = torch.zeros(Q, n, t, dtype=dt)
    beta_prev = torch.zeros(t, dtype=dt)
    its = 0
    for it in range(max_iter):
        its = it + 1
        v = K @ q - beta_prev * q_prev
        alpha = (q * v).sum(0)
        v = v - alpha * q
        beta_next = v.norm(dim=0)
        # Paige-Saunders recurrences, one set per shift
        alfa = alpha.unsqueeze(0) + sig
        oldeps = eps
        delta = cs * dbar + sn * alfa
        gbar = sn * dbar - cs * alfa
        eps = sn * beta_next
        dbar = -cs * beta_next
        gamma = torch.sqrt(gbar * gbar + beta_next * beta_next).clamp_min(1e-30)
        cs = gbar / gamma
        sn = beta_next / gamma
        phi = cs * phibar
        phibar = sn * phibar
        w = (q.unsqueeze(0) - oldeps.unsqueeze(1) * w1 - delta.unsqueeze(1) * w2) / gamma.unsqueeze(1)
        upd = phi.unsqueeze(1) * w
        X = X + upd
        w1, w2 = w2, w
        if (it + 1) % 10 == 0 or it + 1 == max_iter:
            conv = (upd.norm(dim=1) / X.norm(dim=1).clamp_min(1e-30)).mean()
            if float(conv) < tol:
                break
        safe = beta_next.clamp_min(1e-30)
        q_prev, q = q, v / safe
        beta_prev = beta_next
    return X * rnorm, its


class _SqrtInvMatmul(torch.autograd.Function):
    """T = K^{-1/2} R by CIQ; backward re-uses the quadrature: dR = K^{-1/2} G, dK = -sym sum_q omega_q Y_q X_q^T with
    X_q = (K + sigma_q)^-1 R, Y_q = (K + sigma_q)^-1 G (gpytorch functions/_sqrt_inv_matmul.py)."""

    @staticmethod
    def forward(ctx, K, R, Q, stats):
        with torch.no_grad():
            lmin, lmax = lanczos_eig_bounds(K, R[:, 0])
            sigma, omega = ciq_quadrature(lmin, lmax, Q)
            X, its = msminres(K, R, sigma)
            T = (omega.to(R.dtype).reshape(-1, 1, 1) * X).sum(0)
        ctx.save_for_backward(K, X)
        ctx.quad = (sigma, omega)
        if stats is not None:
            stats.update(lmin=lmin, lmax=lmax, iterations=its, sigma=sigma, omega=omega)
        return T

    @staticmethod
    def backward(ctx, G):
        K, X = ctx.saved_tensors
        sigma, omega = ctx.quad
        with torch.no_grad():
            Y, _ = msminres(K, G.contiguous(), sigma)
            om = omega.to(G.dtype).reshape(-1, 1, 1)
            dR = (om * Y).sum(0)
            dK = -torch.einsum("qik,qjk->ij", om * Y, X)
            dK = 0.5 * (dK + dK.t())
        return dK, dR, None, None


def sqrt_inv_matmul(K, R, Q=NUM_CONTOUR_QUADRATURE, stats=None):
    return _SqrtInvMatmul.apply(K, R, Q, stats)


def sqrt_inv_matmul_exact(K, R):
    """K^{-1/2} R through the symmetric eigendecomposition (differentiable): what CIQ approximates."""
    lam, U = torch.linalg.eigh(0.5 * (K + K.t()))
    return U @ ((U.t() @ R) / lam.sqrt().unsqueeze(1))


class _NgdInterpTermsFn(torch.autograd.Function):
    """reference CiqDirectionalGradVariationalStrategy.py:19-123: mean / variance interpolation terms from the
    natural parameters; the gradients returned for (natural_vec, natural_mat) are those w.r.t. the expectation
    parameters.  The reference's preconditioned linear_cg on the precision is restated as a direct solve."""

    @staticmethod
    def forward(ctx, interp_term, natural_vec, natural_mat):
        prec = natural_mat * -2.0                                             # :41
        sol = torch.linalg.solve(prec, torch.cat([natural_vec.unsqueeze(-1), interp_term], dim=-1))   # :51-59
        expec_vec, s_times = sol[:, 0], sol[:, 1:]                            # :60-61
        interp_mean = s_times.t() @ natural_vec                               # :65
        interp_var = (s_times * interp_term).sum(0)                           # :69
        kl_div = torch.zeros((), dtype=interp_term.dtype)                     # :74 (not computed in the forward pass)
        ctx.save_for_backward(interp_term, s_times, interp_mean, natural_vec, expec_vec, prec)
        return interp_mean, interp_var, kl_div

    @staticmethod
    def backward(ctx, g_mean, g_var, g_kl):
        interp_term, s_times, interp_mean, natural_vec, expec_vec, prec = ctx.saved_tensors
        gm, gv = g_mean.unsqueeze(0), g_var.unsqueeze(0)
        d_interp = 2.0 * gv * s_times + gm * expec_vec.unsqueeze(1)                                  # :94-96
        d_vec = (-2.0 * (gv * interp_mean.unsqueeze(0) * interp_term).sum(1) + (gm * interp_term).sum(1)
                 + g_kl * natural_vec)                                                                # :102-108
        eye = torch.eye(expec_vec.shape[0], dtype=expec_vec.dtype)
        d_mat = (gv * interp_term) @ interp_term.t() + g_kl * 0.5 * (eye - prec)                      # :115-118
        return d_interp, d_vec, d_mat


def ciq_predictive(params, x, D, Q=NUM_CONTOUR_QUADRATURE, exact=False, stats=None, kzz_jitter=KZZ_JITTER, kxx_jitter=0.0,
                   assembly=None):
    """CiqDirectionalGradVariationalStrategy.forward with a NaturalVariationalDistribution (:197-268).
    Returns (mu, var, kl) with kl == 0 as in the reference's forward (:74)."""
    Z, V = params["inducing_points"], params["inducing_directions"]
    c = params["constant"].reshape(())
    ell, s, _ = constrained(params)
    M, d = Z.shape
    p = V.shape[0] // M
    B = x.shape[0]
    assert D.shape[0] // B == p, "Need minibatch dim to be same as number of directions for kernel"
    dt = x.dtype
    kernel_matrix = assembly or globals()["kernel_matrix"]                     # (``kernel_matrix_refseq``: the reference's op sequence)
    K_ZX = s * kernel_matrix(Z, x, V, D, ell)                                  # :218-222
    K_ZZ = s * kernel_matrix(Z, Z, V, V, ell)
    # (gpytorch's plain CiqVariationalStrategy -- grad_svgp.py:25-27, traditional_vi.py:22-24, forward quoted at CiqDGVS.py:243-251
    #  -- uses add_jitter(1e-2) here and data_data_covar.add_jitter(1e-4) below)
    K_ZZ = K_ZZ + kzz_jitter * torch.eye(K_ZZ.shape[0], dtype=dt)              # :230-234
    dg = s * kernel_diag(B, p, ell).to(dt)                                     # :235-239, diag only (:265)
    T = sqrt_inv_matmul_exact(K_ZZ, K_ZX) if exact else sqrt_inv_matmul(K_ZZ, K_ZX, Q, stats)     # :255-256
    interp_mean, interp_var, kl = _NgdInterpTermsFn.apply(T, params["natural_vec"], params["natural_mat"])  # :261-263
    var = (dg + kxx_jitter - (T * T).sum(0) + interp_var).clamp_min(MIN_VARIANCE)           # :265-266
    return interp_mean + c, var, kl                                            # :126 (constant mean on all rows), :293


def ciq_forward(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, Q=NUM_CONTOUR_QUADRATURE, exact=False,
                stats=None, kzz_jitter=KZZ_JITTER, kxx_jitter=0.0, assembly=None):
    mu, var, kl = ciq_predictive(params, x, D, Q, exact, stats, kzz_jitter, kxx_jitter, assembly)
    _, _, noise = constrained(params)
    Bp = y.shape[0] if global_rows is None else global_rows
    varn = (var + noise).clamp_min(MIN_VARIANCE)
    if mll_type == "ELBO":
        ll = -0.5 * (((y - mu) ** 2 + varn) / noise + torch.log(noise) + math.log(2 * math.pi))
    elif mll_type == "PLL":
        tot = (varn + noise).clamp_min(1e-8)
        ll = -0.5 * ((y - mu) ** 2 / tot + torch.log(tot) + math.log(2 * math.pi))
    else:
        raise ValueError(mll_type)
    loss = -(ll.sum() / Bp - kl / num_data)
    return loss, mu, varn


def ciq_loss_and_grads(params, x, y, D, num_data, mll_type="ELBO", global_rows=None, Q=NUM_CONTOUR_QUADRATURE,
                       exact=False, stats=None, kzz_jitter=KZZ_JITTER, kxx_jitter=0.0, assembly=None):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss, mu, varn = ciq_forward(ps, x, y, D, num_data, mll_type, global_rows, Q, exact, stats, kzz_jitter, kxx_jitter, assembly)
    loss.backward()
    grads = {k: (ps[k].grad if ps[k].grad is not None else torch.zeros_like(ps[k])) for k in ps}
    grads["natural_mat"] = 0.5 * (grads["natural_mat"] + grads["natural_mat"].t())
    return loss.detach(), grads, mu.detach(), varn.detach()


# --------------------------------------------------------------------------------------
# synthetic data of the reference's smoke tests
# --------------------------------------------------------------------------------------
def testfun(x):
    """f(x)=sin(2 pi |x|^2) with gradient, reference tests/testfun.py:4-12 (any d)."""
    sq = (x ** 2).sum(1)
    fx = torch.sin(2 * math.pi * sq)
    gx = 4 * math.pi * torch.cos(2 * math.pi * sq)[:, None] * x
    return torch.cat([fx[:, None], gx], 1)
