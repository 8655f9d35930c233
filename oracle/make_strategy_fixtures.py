"""Generate tests/golden/strategy_*.npz by executing the REFERENCE variational-strategy files in this container.

What this pins: the COMPOSITION of ``DirectionalGradVariationalStrategy.forward`` (reference
directionalvi/DirectionalGradVariationalStrategy.py:89-208) and of its derivative-free and shared-direction siblings
(DFreeDirectionalGradVariationalStrategy.py, SharedDirectionalGradVariationalStrategy.py) -- which kernel blocks are built
from which (points, directions) pairs, the 1e-3 jitter on K_ZZ, the double-precision Cholesky and the two triangular solves,
which of K_ZX / K_XZ^T feeds the mean and which the covariance, the (S - I) middle term, the 1e-4 jitter on K_XX -- together
with the reference kernel file they call (RBFKernelDirectionalGrad.py).  The reference text is executed from where it lies
(never copied); the oracle (oracle/dsvgp_oracle.py) and the HIP engines are then held to the stored (mean, covariance).

gpytorch (un-vendored dependency, absent here) is replaced by throw-away DENSE containers that carry only the documented
semantics of the handful of calls those forwards make:
    LazyTensor.evaluate / add_jitter(1e-3 default: adds to the diagonal) / mul(c) / @ / transpose / shape,
    TriangularLazyTensor.inv_matmul (lower triangular solve), Sum / Matmul / Diag lazy tensors as their dense values,
    MultivariateNormal as a (mean, covariance) pair, psd_safe_cholesky's first attempt (plain Cholesky),
    settings.cholesky_jitter.value() = 1e-6, trace_mode off, ``cached`` as the identity.
They are not a gpytorch re-implementation (no lazy algebra, no caching, no batching) and nothing else in the repo uses them.
What stays unpinned after this: gpytorch's own primitives behind those names (the likelihood / ELBO / KL formulas, the jitter
retry ladder, the variational distributions), which no file of the reference contains.

Usage:  python oracle/make_strategy_fixtures.py   (writes tests/golden/strategy_*.npz; needs /root/reference)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REFDIR = "/root/reference/directionalvi"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# ------------------------------------------------------------------ throw-away dense stand-ins
class LazyT:
    def __init__(self, t):
        self.t = t.t if isinstance(t, LazyT) else t

    def evaluate(self):
        return self.t

    def add_jitter(self, jitter_val=1e-3):
        return LazyT(self.t + jitter_val * torch.eye(self.t.shape[-1], dtype=self.t.dtype))

    def mul(self, c):
        return LazyT(self.t * c)

    def double(self):
        return LazyT(self.t.double())

    def transpose(self, a, b):
        return LazyT(self.t.transpose(a, b))

    def __matmul__(self, other):
        return LazyT(self.t @ (other.t if isinstance(other, LazyT) else other))

    def __rmatmul__(self, other):
        return LazyT(other @ self.t)

    def __getitem__(self, idx):
        return LazyT(self.t[idx])

    @property
    def shape(self):
        return self.t.shape

    def diag(self):
        return torch.diagonal(self.t, dim1=-2, dim2=-1)


class DiagLazyTensor(LazyT):
    def __init__(self, diag):
        super().__init__(torch.diag_embed(diag))


class TriangularLazyTensor(LazyT):
    def inv_matmul(self, rhs):
        return torch.linalg.solve_triangular(self.t, rhs, upper=False)


class SumLazyTensor(LazyT):
    def __init__(self, *parts):
        super().__init__(sum(_dense(p) for p in parts))


class MatmulLazyTensor(LazyT):
    def __init__(self, a, b):
        super().__init__(_dense(a) @ _dense(b))


class CholLazyTensor(LazyT):
    """what CholeskyVariationalDistribution hands to the strategy: S = L_S L_S^T from the (masked) factor"""
    def __init__(self, L):
        L = _dense(L)
        super().__init__(L @ L.transpose(-1, -2))


def _dense(x):
    return x.t if isinstance(x, LazyT) else x


class MultivariateNormal:
    def __init__(self, mean, covar):
        self.loc = self.mean = mean
        self.lazy_covariance_matrix = covar if isinstance(covar, LazyT) else LazyT(covar)

    @property
    def covariance_matrix(self):
        return self.lazy_covariance_matrix.evaluate()


def _cached(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]
    return lambda fn: fn


def _install_stand_ins():
    def postprocess_rbf(dist_mat):
        return dist_mat.div(-2).exp()

    class RBFKernel(torch.nn.Module):
        """base class of the reference kernel file: lengthscale, covar_dist (squared distances, difference form), diag"""
        def __init__(self):
            super().__init__()
            self._ell = torch.tensor([[0.6931471805599453]], dtype=torch.float64)

        @property
        def lengthscale(self):
            return self._ell

        def covar_dist(self, x1, x2, square_dist=False, dist_postprocess_func=None, **params):
            assert square_dist
            diff = x1.unsqueeze(-2) - x2.unsqueeze(-3)
            return dist_postprocess_func((diff * diff).sum(-1))

        def forward(self, x1, x2, diag=False, **params):
            assert diag
            return torch.ones(x1.shape[-2], dtype=x1.dtype)

    class _VariationalStrategy(torch.nn.Module):
        def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=True):
            super().__init__()
            object.__setattr__(self, "model", model)
            torch.nn.Module.register_parameter(self, "inducing_points", torch.nn.Parameter(inducing_points.clone()))
            self._variational_distribution = variational_distribution
            self.register_buffer("variational_params_initialized", torch.tensor(0))

        def register_parameter(self, name, parameter=None, param=None):       # (gpytorch.Module spells the keyword ``parameter``)
            torch.nn.Module.register_parameter(self, name, parameter if parameter is not None else param)

    class _Setting:
        def __init__(self, v):
            self._v = v

        def value(self, *a):
            return self._v

        def on(self):
            return bool(self._v)

    def psd_safe_cholesky(A, jitter=None, max_tries=3, **kw):
        return torch.linalg.cholesky(A)         # (first attempt of gpytorch's routine; the fixtures are well conditioned)

    def mk(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    settings = mk("gpytorch.settings", cholesky_jitter=_Setting(1e-6), trace_mode=_Setting(False))
    mk("gpytorch", settings=settings)
    mk("gpytorch.distributions", MultivariateNormal=MultivariateNormal)
    mk("gpytorch.lazy", DiagLazyTensor=DiagLazyTensor, MatmulLazyTensor=MatmulLazyTensor, RootLazyTensor=LazyT,
       SumLazyTensor=SumLazyTensor, TriangularLazyTensor=TriangularLazyTensor, delazify=_dense)
    mk("gpytorch.lazy.kronecker_product_lazy_tensor", KroneckerProductLazyTensor=object)
    mk("gpytorch.utils")
    mk("gpytorch.utils.cholesky", psd_safe_cholesky=psd_safe_cholesky)
    mk("gpytorch.utils.errors", CachingError=type("CachingError", (RuntimeError,), {}))
    mk("gpytorch.utils.memoize", cached=_cached, clear_cache_hook=lambda *a, **k: None,
       pop_from_cache_ignore_args=lambda *a, **k: None)
    mk("gpytorch.utils.warnings", OldVersionWarning=type("OldVersionWarning", (UserWarning,), {}))
    mk("gpytorch.variational")
    mk("gpytorch.variational._variational_strategy", _VariationalStrategy=_VariationalStrategy)
    mk("gpytorch.kernels")
    mk("gpytorch.kernels.rbf_kernel", RBFKernel=RBFKernel, postprocess_rbf=postprocess_rbf)

    # ---- CiqDirectionalGradVariationalStrategy.py: the file carries its own autograd function (_NgdInterpTerms, :19-123) and
    # the forward composition; gpytorch supplies the precision solve (preconditioned CG) and K^-1/2 R (contour-integral
    # quadrature + msMINRES).  Here: the EXACT solve and the EXACT symmetric inverse square root (what both iterations converge
    # to), so the stored vectors are the limit the reference's numbers approach within its own tolerances (1e-4 msMINRES).
    class _Ctx:
        def __init__(self, v=None):
            self._v = v

        def value(self, *a):
            return self._v

        def __call__(self, *a):
            return self

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class _MinVar(_Ctx):
        def value(self, dtype=None):
            return 1e-6

    settings.max_cg_iterations = _Ctx(1000)
    settings.eval_cg_tolerance = _Ctx(0.01)
    settings.cg_tolerance = _Ctx(1.0)
    settings.max_lanczos_quadrature_iterations = _Ctx(20)
    settings.max_preconditioner_size = _Ctx(0)
    settings.min_variance = _MinVar()

    def linear_cg(matmul_closure, rhs, **kw):
        n = rhs.shape[-2]
        A = matmul_closure(torch.eye(n, dtype=rhs.dtype))
        return torch.linalg.solve(A, rhs)

    class _SqrtInv(LazyT):
        def sqrt_inv_matmul(self, rhs):
            w, Q = torch.linalg.eigh(0.5 * (self.t + self.t.transpose(-1, -2)))
            return (Q * w.rsqrt()) @ (Q.transpose(-1, -2) @ rhs)

    class NaturalVariationalDistribution(torch.nn.Module):
        def __init__(self, natural_vec, natural_mat):
            super().__init__()
            self.natural_vec = torch.nn.Parameter(natural_vec.clone())
            self.natural_mat = torch.nn.Parameter(natural_mat.clone())
            self.dtype, self.device = natural_vec.dtype, natural_vec.device

        def shape(self):
            return self.natural_vec.shape

    sys.modules["gpytorch.distributions"].Delta = object
    sys.modules["gpytorch.lazy"].lazify = lambda t: _SqrtInv(t)
    mk("gpytorch.module", Module=torch.nn.Module)
    mk("gpytorch.utils.broadcasting", _mul_broadcast_shape=lambda *shapes: torch.broadcast_shapes(*shapes))
    sys.modules["gpytorch.utils"].linear_cg = linear_cg
    mk("gpytorch.variational.natural_variational_distribution", NaturalVariationalDistribution=NaturalVariationalDistribution)
    return NaturalVariationalDistribution


def _load(fname, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REFDIR, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Model:
    """the three lines of the reference ``GPModel.forward`` (directional_vi.py:59-62) around the REFERENCE kernel object:
    ConstantMean, ScaleKernel(RBFKernelDirectionalGrad()), MultivariateNormal(mean, covar)."""

    def __init__(self, kernel, constant, outputscale):
        self.base = kernel
        self.c = constant
        self.s = outputscale
        model = self

        class _Scale:
            base_kernel = kernel

            def __call__(self, x1, x2=None, **params):
                x2 = x1 if x2 is None else x2
                return LazyT(model.s * model.base.forward(x1, x2, **params))

        self.covar_module = _Scale()

    def mean_module(self, x):
        return torch.ones(x.shape[0], dtype=x.dtype) * self.c

    def forward(self, x, **params):
        return MultivariateNormal(self.mean_module(x), self.covar_module(x, **params))


class _VarDist:
    def __init__(self, n):
        self._n = n
        self.dtype = torch.float64
        self.device = torch.device("cpu")

    def shape(self):
        return torch.Size([self._n])


# (name, strategy file, class, N, d, M, p, B, data outputs per point, shared directions)
CASES = [
    ("dgvs_a", "DirectionalGradVariationalStrategy.py", 40, 3, 7, 2, 9, "all", False),
    ("dgvs_b_c2geom", "DirectionalGradVariationalStrategy.py", 80, 5, 12, 2, 17, "all", False),
    ("dgvs_c_c4geom", "DirectionalGradVariationalStrategy.py", 90, 20, 6, 5, 8, "all", False),
    ("dgvs_d_fullgrad", "DirectionalGradVariationalStrategy.py", 50, 4, 5, 4, 6, "all", False),
    ("dfree_a", "DFreeDirectionalGradVariationalStrategy.py", 60, 4, 8, 2, 11, "values", False),
    ("shared_a", "SharedDirectionalGradVariationalStrategy.py", 60, 4, 8, 2, 11, "all", True),
]


def main():
    torch.set_default_dtype(torch.float64)        # like the reference's experiment scripts (exp_script.py:56)
    NatDist = _install_stand_ins()
    Kern = _load("RBFKernelDirectionalGrad.py", "_ref_rbf_dirgrad").RBFKernelDirectionalGrad
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dsvgp_oracle as O
    for ci, (name, fname, N, d, M, p, B, outputs, shared) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + ci)
        X = torch.rand(N, d, generator=g, dtype=torch.float64)
        Z = X[:M].clone()
        x = X[M:M + B].contiguous()
        nV = p if shared else M * p
        V = torch.eye(d, dtype=torch.float64)[:p].repeat(1 if shared else M, 1) + 0.2 * torch.randn(nV, d, generator=g,
                                                                                                      dtype=torch.float64)
        D = torch.eye(d, dtype=torch.float64)[:p].repeat(B, 1) + 0.1 * torch.randn(B * p, d, generator=g, dtype=torch.float64)
        nq = M + p if shared else M * (p + 1)
        m = 0.3 * torch.randn(nq, generator=g, dtype=torch.float64)
        LS = torch.tril(torch.eye(nq, dtype=torch.float64) + 0.1 * torch.randn(nq, nq, generator=g, dtype=torch.float64))
        ell, s, c = 0.55 + 0.1 * ci, 1.3, 0.2
        kern = Kern()
        kern._ell = torch.tensor([[ell]], dtype=torch.float64)
        model = _Model(kern, c, s)
        Strat = getattr(_load(fname, "_ref_strategy_%d" % ci), "DirectionalGradVariationalStrategy")
        strat = Strat(model, Z, V, _VarDist(nq), learn_inducing_locations=True)
        with torch.no_grad():
            out = strat.forward(x, strat.inducing_points, m, CholLazyTensor(LS), derivative_directions=D)
        mean, cov = out.mean.detach(), out.covariance_matrix.detach()
        np.savez(os.path.join(OUT, "strategy_%s.npz" % name), x=x.numpy(), Z=Z.numpy(), V=V.numpy(), D=D.numpy(),
                 variational_mean=m.numpy(), chol_variational_covar=LS.numpy(), lengthscale=np.float64(ell),
                 outputscale=np.float64(s), constant=np.float64(c), p=np.int64(p), outputs=np.str_(outputs),
                 shared=np.bool_(shared), mean=mean.numpy(), covariance=cov.numpy())
        print("%-18s %s: mean %s, covariance %s, |mean| max %.3f, cov diag min %.3e" % (
            name, fname, tuple(mean.shape), tuple(cov.shape), mean.abs().max().item(), torch.diagonal(cov).min().item()))


    grad_case(Kern, O)
    ciq_cases(Kern, NatDist)
    elbo_gradient_cases(Kern)


def elbo_gradient_cases(Kern):
    """Gradients THROUGH the reference's strategy forward and kernel file (torch autograd over the reference's own operations,
    the dense stand-ins are plain torch): loss = -(sum_j ll_j / B' - KL / num_data) with the closed forms of gpytorch's
    GaussianLikelihood.expected_log_prob and of KL(N(m, S) || N(0, I)) written out here (those two formulas are restated, as
    in the oracle; everything they are applied to -- mean and variance of q(f) as functions of Z, V, m, L_S, lengthscale,
    outputscale, constant -- is the reference's text)."""
    import math
    import torch.nn.functional as F
    cases = [("dgvs", "DirectionalGradVariationalStrategy.py", 70, 4, 9, 2, 13, "all", False),
             ("dgvs_c4geom", "DirectionalGradVariationalStrategy.py", 80, 20, 5, 5, 7, "all", False),
             ("dfree", "DFreeDirectionalGradVariationalStrategy.py", 60, 4, 8, 2, 11, "values", False),
             ("shared", "SharedDirectionalGradVariationalStrategy.py", 60, 4, 8, 2, 11, "all", True)]
    for ci, (name, fname, N, d, M, p, B, outputs, shared) in enumerate(cases):
        g = torch.Generator().manual_seed(600 + ci)
        X = torch.rand(N, d, generator=g, dtype=torch.float64)
        Z, x = X[:M].clone(), X[M:M + B].contiguous()
        nV = p if shared else M * p
        V = torch.eye(d, dtype=torch.float64)[:p].repeat(1 if shared else M, 1) + 0.2 * torch.randn(nV, d, generator=g,
                                                                                                      dtype=torch.float64)
        D = torch.eye(d, dtype=torch.float64)[:p].repeat(B, 1) + 0.1 * torch.randn(B * p, d, generator=g, dtype=torch.float64)
        nq = M + p if shared else M * (p + 1)
        nout = B if outputs == "values" else B * (p + 1)
        y = torch.randn(nout, generator=g, dtype=torch.float64)
        m = (0.3 * torch.randn(nq, generator=g, dtype=torch.float64)).requires_grad_(True)
        LS = (torch.eye(nq, dtype=torch.float64) + 0.1 * torch.randn(nq, nq, generator=g, dtype=torch.float64)).requires_grad_(True)
        raw_ell = torch.tensor([[0.3]], dtype=torch.float64, requires_grad=True)
        raw_s = torch.tensor(0.2, dtype=torch.float64, requires_grad=True)
        raw_noise = torch.tensor([-0.5], dtype=torch.float64, requires_grad=True)
        const = torch.tensor([0.1], dtype=torch.float64, requires_grad=True)
        num_data = float((d + 1) * N)
        kern = Kern()
        kern._ell = F.softplus(raw_ell)
        model = _Model(kern, const.reshape(()), F.softplus(raw_s))
        Strat = getattr(_load(fname, "_ref_strategy_g%d" % ci), "DirectionalGradVariationalStrategy")
        strat = Strat(model, Z, V, _VarDist(nq), learn_inducing_locations=True)
        out = strat.forward(x, strat.inducing_points, m, CholLazyTensor(torch.tril(LS)), derivative_directions=D)
        mu, var = out.mean, torch.diagonal(out.covariance_matrix)
        noise = F.softplus(raw_noise).reshape(()) + 1e-4                       # GaussianLikelihood: GreaterThan(1e-4)
        varn = (var + noise).clamp_min(1e-6)                                   # likelihood(q(f)).variance
        ll = -0.5 * (((y - mu) ** 2 + varn) / noise + torch.log(noise) + math.log(2 * math.pi))
        Lt = torch.tril(LS)
        kl = 0.5 * ((m * m).sum() + (Lt * Lt).sum() - nq - torch.log(torch.diagonal(Lt) ** 2).sum())
        loss = -(ll.sum() / nout - kl / num_data)
        loss.backward()
        np.savez(os.path.join(OUT, "strategy_grad_%s.npz" % name), x=x.numpy(), y=y.numpy(), Z=Z.numpy(), V=V.numpy(), D=D.numpy(),
                 variational_mean=m.detach().numpy(), chol_variational_covar=LS.detach().numpy(), raw_lengthscale=raw_ell.detach().numpy(),
                 raw_outputscale=raw_s.detach().numpy(), raw_noise=raw_noise.detach().numpy(), constant=const.detach().numpy(),
                 num_data=np.float64(num_data), p=np.int64(p), outputs=np.str_(outputs), shared=np.bool_(shared),
                 loss=loss.detach().numpy(), d_inducing_points=strat.inducing_points.grad.numpy(),
                 d_inducing_directions=strat.inducing_directions.grad.numpy(), d_variational_mean=m.grad.numpy(),
                 d_chol_variational_covar=LS.grad.numpy(), d_raw_lengthscale=raw_ell.grad.numpy(), d_raw_outputscale=raw_s.grad.numpy(),
                 d_raw_noise=raw_noise.grad.numpy(), d_constant=const.grad.numpy())
        print("grad_%-16s loss %.6f, |dZ| max %.3e, |dV| max %.3e" % (name, loss.item(), strat.inducing_points.grad.abs().max().item(),
                                                                      strat.inducing_directions.grad.abs().max().item()))
    # the CIQ strategy with a natural q(u): its own autograd function returns the gradients w.r.t. the EXPECTATION parameters
    # (natural gradient descent); K^-1/2 R is the exact inverse square root (differentiated by autograd through eigh)
    ref = _load("CiqDirectionalGradVariationalStrategy.py", "_ref_ciq_strategy_g")
    NatDist = sys.modules["gpytorch.variational.natural_variational_distribution"].NaturalVariationalDistribution
    # (case 1, round 3: C2 geometry at M' = 300, B' = 288 -- five 64-column blocks of the Cholesky / GEMM tiling, several msMINRES
    #  convergence checks on the HIP side)
    for ci, (N, d, M, p, B) in enumerate(((60, 4, 8, 2, 11), (1500, 5, 100, 2, 96))):
        g = torch.Generator().manual_seed(700 + ci)
        X = torch.rand(N, d, generator=g, dtype=torch.float64)
        Z, x = X[:M].clone(), X[M:M + B].contiguous()
        V = torch.eye(d, dtype=torch.float64)[:p].repeat(M, 1) + 0.2 * torch.randn(M * p, d, generator=g, dtype=torch.float64)
        D = torch.eye(d, dtype=torch.float64)[:p].repeat(B, 1) + 0.1 * torch.randn(B * p, d, generator=g, dtype=torch.float64)
        nq, nout = M * (p + 1), B * (p + 1)
        y = torch.randn(nout, generator=g, dtype=torch.float64)
        nv = 0.3 * torch.randn(nq, generator=g, dtype=torch.float64)
        R = 0.2 * min(1.0, (24.0 / nq) ** 0.5) * torch.randn(nq, nq, generator=g, dtype=torch.float64)   # (same conditioning of the precision at every size)
        nm = -0.5 * (torch.eye(nq, dtype=torch.float64) + R @ R.t())
        raw_ell = torch.tensor([[0.3]], dtype=torch.float64, requires_grad=True)
        raw_s = torch.tensor(0.2, dtype=torch.float64, requires_grad=True)
        raw_noise = torch.tensor([-0.5], dtype=torch.float64, requires_grad=True)
        const = torch.tensor([0.1], dtype=torch.float64, requires_grad=True)
        num_data = float((d + 1) * N)
        kern = Kern()
        kern._ell = F.softplus(raw_ell)
        dist = NatDist(nv, nm)
        strat = ref.CiqDirectionalGradVariationalStrategy(_Model(kern, const.reshape(()), F.softplus(raw_s)), Z, V, dist)
        out = strat.forward(x, strat.inducing_points, None, None, derivative_directions=D)
        mu, var = out.mean, torch.diagonal(out.covariance_matrix)
        kl = strat._memoize_cache["kl"]                                         # (value 0 in the forward, :74; its gradient is what counts)
        noise = F.softplus(raw_noise).reshape(()) + 1e-4
        varn = (var + noise).clamp_min(1e-6)
        ll = -0.5 * (((y - mu) ** 2 + varn) / noise + torch.log(noise) + math.log(2 * math.pi))
        loss = -(ll.sum() / nout - kl / num_data)
        loss.backward()
        np.savez(os.path.join(OUT, "strategy_ciq_grad_%d.npz" % ci), x=x.numpy(), y=y.numpy(), Z=Z.numpy(), V=V.numpy(), D=D.numpy(),
                 natural_vec=nv.numpy(), natural_mat=nm.numpy(), raw_lengthscale=raw_ell.detach().numpy(),
                 raw_outputscale=raw_s.detach().numpy(), raw_noise=raw_noise.detach().numpy(), constant=const.detach().numpy(),
                 num_data=np.float64(num_data), p=np.int64(p), loss=loss.detach().numpy(),
                 d_inducing_points=strat.inducing_points.grad.numpy(), d_inducing_directions=strat.inducing_directions.grad.numpy(),
                 d_natural_vec=dist.natural_vec.grad.numpy(), d_natural_mat=dist.natural_mat.grad.numpy(),
                 d_raw_lengthscale=raw_ell.grad.numpy(), d_raw_outputscale=raw_s.grad.numpy(), d_raw_noise=raw_noise.grad.numpy(),
                 d_constant=const.grad.numpy())
        print("ciq_grad_%d           loss %.6f, |d natural_mat| max %.3e" % (ci, loss.item(), dist.natural_mat.grad.abs().max().item()))


def grad_case(Kern, O):
    """``GradVariationalStrategy.forward`` (GradVariationalStrategy.py:87-137, BASELINE config 3): joint covariance of [Z ; x] from
    ONE model call, sliced into its three blocks.  Its kernel is gpytorch's RBFKernelGrad (not a file of the reference); the
    reference's directional kernel with all d canonical directions at every point is the same matrix (value + full gradient per
    point, interleaved) and stands in for it here."""
    ref = _load("GradVariationalStrategy.py", "_ref_grad_strategy")
    for ci, (N, d, M, B) in enumerate(((40, 3, 6, 7), (60, 5, 8, 9))):
        g = torch.Generator().manual_seed(500 + ci)
        X = torch.rand(N, d, generator=g, dtype=torch.float64)
        Z, x = X[:M].clone(), X[M:M + B].contiguous()
        nq = M * (d + 1)
        m = 0.3 * torch.randn(nq, generator=g, dtype=torch.float64)
        LS = torch.tril(torch.eye(nq, dtype=torch.float64) + 0.1 * torch.randn(nq, nq, generator=g, dtype=torch.float64))
        ell, s, c = 0.7 + 0.1 * ci, 1.1, -0.1
        kern = Kern()
        kern._ell = torch.tensor([[ell]], dtype=torch.float64)
        model = _Model(kern, c, s)
        eye = torch.eye(d, dtype=torch.float64)

        def forward(xx, _model=model, _eye=eye):
            v = _eye.repeat(xx.shape[0], 1)
            return MultivariateNormal(_model.mean_module(xx), _model.covar_module(xx, xx, v1=v, v2=v))

        model.forward = forward
        strat = ref.GradVariationalStrategy(model, Z, _VarDist(nq), learn_inducing_locations=True)
        with torch.no_grad():
            out = strat.forward(x, strat.inducing_points, m, CholLazyTensor(LS))
        np.savez(os.path.join(OUT, "strategy_gradvs_%d.npz" % ci), x=x.numpy(), Z=Z.numpy(), V=eye.repeat(M, 1).numpy(),
                 D=eye.repeat(B, 1).numpy(), variational_mean=m.numpy(), chol_variational_covar=LS.numpy(),
                 lengthscale=np.float64(ell), outputscale=np.float64(s), constant=np.float64(c), p=np.int64(d),
                 outputs=np.str_("all"), shared=np.bool_(False), mean=out.mean.numpy(), covariance=out.covariance_matrix.numpy())
        print("gradvs_%d             GradVariationalStrategy.forward: mean %s" % (ci, tuple(out.mean.shape)))


def ciq_cases(Kern, NatDist):
    """(a) the reference file's own autograd function ``_NgdInterpTerms`` (forward + hand-written backward), (b) the forward of
    ``CiqDirectionalGradVariationalStrategy`` with a natural q(u): mean, DIAGONAL covariance and the memoised KL gradient path"""
    ref = _load("CiqDirectionalGradVariationalStrategy.py", "_ref_ciq_strategy")
    for ci, (n, t) in enumerate(((9, 5), (24, 17))):
        g = torch.Generator().manual_seed(300 + ci)
        T = torch.randn(n, t, generator=g, dtype=torch.float64, requires_grad=True)
        nv = torch.randn(n, generator=g, dtype=torch.float64, requires_grad=True)
        R = 0.2 * torch.randn(n, n, generator=g, dtype=torch.float64)
        nm = (-0.5 * (torch.eye(n, dtype=torch.float64) + R @ R.t())).requires_grad_(True)
        imean, ivar, kl = ref._NgdInterpTerms.apply(T, nv, nm)
        gm = torch.randn(t, generator=g, dtype=torch.float64)
        gv = torch.randn(t, generator=g, dtype=torch.float64)
        gk = torch.randn((), generator=g, dtype=torch.float64)
        (imean * gm).sum().add((ivar * gv).sum()).add(kl * gk).backward()
        np.savez(os.path.join(OUT, "strategy_ngd_interp_%d.npz" % ci), interp_term=T.detach().numpy(),
                 natural_vec=nv.detach().numpy(), natural_mat=nm.detach().numpy(), interp_mean=imean.detach().numpy(),
                 interp_var=ivar.detach().numpy(), kl_div=kl.detach().numpy(), g_mean=gm.numpy(), g_var=gv.numpy(),
                 g_kl=gk.numpy(), d_interp_term=T.grad.numpy(), d_natural_vec=nv.grad.numpy(), d_natural_mat=nm.grad.numpy())
        print("ngd_interp_%d         _NgdInterpTerms: n=%d t=%d, |d natural_mat| max %.3f" % (ci, n, t, nm.grad.abs().max().item()))
    for ci, (N, d, M, p, B) in enumerate(((40, 3, 6, 2, 7), (70, 5, 10, 2, 12))):
        g = torch.Generator().manual_seed(400 + ci)
        X = torch.rand(N, d, generator=g, dtype=torch.float64)
        Z, x = X[:M].clone(), X[M:M + B].contiguous()
        V = torch.eye(d, dtype=torch.float64)[:p].repeat(M, 1) + 0.2 * torch.randn(M * p, d, generator=g, dtype=torch.float64)
        D = torch.eye(d, dtype=torch.float64)[:p].repeat(B, 1) + 0.1 * torch.randn(B * p, d, generator=g, dtype=torch.float64)
        nq = M * (p + 1)
        nv = 0.3 * torch.randn(nq, generator=g, dtype=torch.float64)
        R = 0.2 * torch.randn(nq, nq, generator=g, dtype=torch.float64)
        nm = -0.5 * (torch.eye(nq, dtype=torch.float64) + R @ R.t())
        ell, s, c = 0.6 + 0.1 * ci, 1.2, 0.15
        kern = Kern()
        kern._ell = torch.tensor([[ell]], dtype=torch.float64)
        strat = ref.CiqDirectionalGradVariationalStrategy(_Model(kern, c, s), Z, V, NatDist(nv, nm))
        with torch.no_grad():
            out = strat.forward(x, strat.inducing_points, None, None, derivative_directions=D)
        cov = out.covariance_matrix
        assert (cov - torch.diag(torch.diagonal(cov))).abs().max().item() == 0.0       # DiagLazyTensor(predictive_var), :264-267
        np.savez(os.path.join(OUT, "strategy_ciq_%d.npz" % ci), x=x.numpy(), Z=Z.numpy(), V=V.numpy(), D=D.numpy(),
                 natural_vec=nv.numpy(), natural_mat=nm.numpy(), lengthscale=np.float64(ell), outputscale=np.float64(s),
                 constant=np.float64(c), p=np.int64(p), mean=out.mean.numpy(), variance=torch.diagonal(cov).numpy())
        print("ciq_%d                CiqDirectionalGradVariationalStrategy.forward: mean %s, var min %.3e" % (
            ci, tuple(out.mean.shape), torch.diagonal(cov).min().item()))
        # the SAME reference forward with ``sqrt_inv_matmul`` = the oracle's contour-integral quadrature + msMINRES (float64,
        # Q = 15, tolerance 1e-4 -- the restatement of gpytorch's iteration) instead of the exact root: what the HIP CIQ path
        # should reproduce to the iteration's own tolerance, not only in the limit
        import dsvgp_oracle as O

        class _Minres(LazyT):
            def sqrt_inv_matmul(self, rhs):
                return O.sqrt_inv_matmul(self.t, rhs)

        keep = ref.lazify
        ref.lazify = lambda t: _Minres(t)
        try:
            strat2 = ref.CiqDirectionalGradVariationalStrategy(_Model(kern, c, s), Z, V, NatDist(nv, nm))
            with torch.no_grad():
                out2 = strat2.forward(x, strat2.inducing_points, None, None, derivative_directions=D)
        finally:
            ref.lazify = keep
        var2 = torch.diagonal(out2.covariance_matrix)
        np.savez(os.path.join(OUT, "strategy_ciq_minres_%d.npz" % ci), x=x.numpy(), Z=Z.numpy(), V=V.numpy(), D=D.numpy(),
                 natural_vec=nv.numpy(), natural_mat=nm.numpy(), lengthscale=np.float64(ell), outputscale=np.float64(s),
                 constant=np.float64(c), p=np.int64(p), mean=out2.mean.numpy(), variance=var2.numpy())
        print("ciq_minres_%d         ... with the oracle's msMINRES quadrature: |mean - exact| %.2e, |var - exact| %.2e" % (
            ci, (out2.mean - out.mean).abs().max().item(), (var2 - torch.diagonal(cov)).abs().max().item()))


if __name__ == "__main__":
    main()
