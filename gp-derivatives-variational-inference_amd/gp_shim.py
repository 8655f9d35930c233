"""Minimal host-side stand-ins for the GPyTorch 1.4.0 objects the reference's hot path touches.

gpytorch is a third-party dependency of the reference (graphite_environment.yml:101) and is not
available here; these classes keep the parameter names / shapes / state_dict keys and the call
protocol (``model(x, derivative_directions=D)`` -> distribution, ``likelihood(dist)``,
``mll(output, y)``) that ``directional_vi.train_gp`` relies on (directional_vi.py:25-65,172,216-219,245-246),
and route the arithmetic to the HIP engine (``_step.ElboEngine``).  They contain no math of their own.
"""
import torch

from . import _ops
from ._step import ElboEngine, NGD_PARAM_NAMES, PARAM_NAMES


class OldVersionWarning(UserWarning):
    """gpytorch.utils.warnings.OldVersionWarning: a checkpoint written before the whitened VariationalStrategy was loaded"""


class PriorDistribution:
    """``strategy(x, prior=True)``: the un-whitened prior p(u) at the inducing points (``.loc``, ``.covariance_matrix``)"""

    def __init__(self, loc, covariance_matrix):
        self.loc = self.mean = loc
        self.covariance_matrix = covariance_matrix


class ConstantMean(torch.nn.Module):
    """gpytorch.means.ConstantMean: parameter ``constant`` of shape [1], init 0."""

    def __init__(self):
        super().__init__()
        self.register_parameter("constant", torch.nn.Parameter(torch.zeros(1)))


class ScaleKernel(torch.nn.Module):
    """gpytorch.kernels.ScaleKernel: ``raw_outputscale`` (0-dim, softplus-constrained, init 0)."""

    def __init__(self, base_kernel):
        super().__init__()
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(())))

    @property
    def outputscale(self):
        return torch.nn.functional.softplus(self.raw_outputscale)


class _NoiseCovar(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(1)))


class GaussianLikelihood(torch.nn.Module):
    """gpytorch.likelihoods.GaussianLikelihood: ``noise_covar.raw_noise`` [1], noise = softplus + 1e-4."""

    def __init__(self):
        super().__init__()
        self.noise_covar = _NoiseCovar()

    @property
    def noise(self):
        return torch.nn.functional.softplus(self.noise_covar.raw_noise) + 1e-4

    def forward(self, dist):
        return dist.with_likelihood(self)


class _VariationalDistribution(torch.nn.Module):
    pass


class CholeskyVariationalDistribution(_VariationalDistribution):
    """gpytorch.variational.CholeskyVariationalDistribution: mean zeros, Cholesky factor identity."""

    def __init__(self, num_inducing_points, mean_init_std=1e-3):
        super().__init__()
        self.mean_init_std = mean_init_std
        self.register_parameter("variational_mean", torch.nn.Parameter(torch.zeros(num_inducing_points)))
        self.register_parameter("chol_variational_covar", torch.nn.Parameter(torch.eye(num_inducing_points)))
        # (only the lower triangle is a parameter -- forward() masks the rest, its gradient is exactly zero there: optim.FusedAdam then
        #  walks the lower triangle only, half the traffic of the update's largest tensor)
        self.chol_variational_covar._dsvgp_tril = True

    def initialize_variational_distribution(self):
        """prior N(0, I): mean <- 0 + mean_init_std * randn, chol <- I (first training call in gpytorch)."""
        with torch.no_grad():
            self.variational_mean.zero_()
            self.variational_mean.add_(torch.randn_like(self.variational_mean), alpha=self.mean_init_std)
            self.chol_variational_covar.copy_(torch.eye(self.variational_mean.shape[0],
                                                        device=self.variational_mean.device))


class NaturalVariationalDistribution(_VariationalDistribution):
    """gpytorch.variational.NaturalVariationalDistribution (1.4.0): q(u) = N(m, S) held in natural parameters
    ``natural_vec`` = S^-1 m (init 0) and ``natural_mat`` = -S^-1 / 2 (init -I/2).  The gradients the engine returns
    for them are taken w.r.t. the expectation parameters, so ``optim.NGD`` performs natural gradient descent
    (reference directional_vi.py:35-37,186-187)."""

    def __init__(self, num_inducing_points, mean_init_std=1e-3):
        super().__init__()
        self.mean_init_std = mean_init_std
        self.register_parameter("natural_vec", torch.nn.Parameter(torch.zeros(num_inducing_points)))
        self.register_parameter("natural_mat", torch.nn.Parameter(torch.eye(num_inducing_points).mul_(-0.5)))

    def initialize_variational_distribution(self):
        """prior N(0, I): natural_vec <- P 0 + mean_init_std * randn, natural_mat <- -P / 2 with P = I."""
        with torch.no_grad():
            self.natural_vec.zero_()
            self.natural_vec.add_(torch.randn_like(self.natural_vec), alpha=self.mean_init_std)
            self.natural_mat.copy_(torch.eye(self.natural_vec.shape[0], device=self.natural_vec.device).mul_(-0.5))


class _ElboFunction(torch.autograd.Function):
    """Fused forward+backward of one minibatch objective on the HIP engine."""

    @staticmethod
    def forward(ctx, engine, x, y, D, num_data, mll_type, dp, names, *params):
        pd = dict(zip(names, [_ops.detach_keep(p) for p in params]))
        if dp is not None:
            loss, grads, mu, varn = dp.loss_and_grads(engine, pd, x, y, D, num_data, mll_type)
        else:
            loss, grads, mu, varn = engine.loss_and_grads(pd, x, y, D, num_data, mll_type)
        ctx.grads = [grads[k] for k in names]
        ctx.mark_non_differentiable(mu, varn)
        return -loss, mu, varn          # mll value = -loss

    @staticmethod
    def backward(ctx, g_elbo, _gm, _gv):
        # d loss / d param = -(d loss / d mll) * grads; the engine's buffers are consumed in place (multi-tensor
        # scale by the device scalar: no per-parameter temporaries, the 36 MB L_S gradient is not copied)
        grads = ctx.grads
        ctx.grads = None
        torch._foreach_mul_(grads, -g_elbo.detach().to(grads[0].dtype))
        return tuple([None] * 8 + list(grads))


class PredictiveDistribution:
    """What ``model(x, derivative_directions=D)`` returns: a handle whose ``mean`` / ``variance``
    (length B(p+1), interleaved) are produced on the GPU on demand."""

    def __init__(self, model, x, D, likelihood=None):
        self.model, self.x, self.D, self.likelihood = model, x, D, likelihood
        self._mu = self._varn = self._var = None

    def with_likelihood(self, likelihood):
        out = PredictiveDistribution(self.model, self.x, self.D, likelihood)
        return out

    def _ensure(self):
        if self._mu is None or self._varn is None:
            lik = self.likelihood
            params = self.model._param_dict(lik)
            mu, varn = self.model.engine.predict(params, self.x, self.D, cache=not self.model.training)
            self._mu, self._varn = mu, varn
            if lik is None:   # q(f) itself: remove the noise again
                self._var = (varn - torch.nn.functional.softplus(params["raw_noise"].reshape(())) - 1e-4)

    @property
    def mean(self):
        if self._mu is None:        # (a fast-path training step has left its mean here; only the variance is then on demand)
            self._ensure()
        return self._mu

    loc = mean

    @property
    def variance(self):
        self._ensure()
        return self._varn if self.likelihood is not None else self._var

    @property
    def value_variance(self):
        """``variance[::p+1]`` (the function-value rows).  After a training step taken with ``need_variance="values"`` it is
        the vector the engine formed from that step's own forward pass (ElboEngine.value_variances); otherwise a slice."""
        v = getattr(self, "_value_varn", None)
        if v is not None and self.likelihood is not None:
            return v
        return self.variance[::getattr(self, "_value_stride", 1)]

    @property
    def stddev(self):
        return self.variance.sqrt()

    def confidence_region(self):
        """(mean - 2 std, mean + 2 std), gpytorch MultivariateNormal.confidence_region"""
        std2 = self.stddev.mul(2)
        return self.mean - std2, self.mean + std2

    # ---- joint distribution over the B(p+1) outputs of the batch (BO drivers: ``preds.sample(torch.Size([n]))``,
    #      reference experiments/GNN_bo/gcn_turbo.py:238-239, experiments/rover/test_turbo.py:138) ----
    def _ensure_joint(self):
        if getattr(self, "_Sigma", None) is None:
            lik = self.likelihood
            params = self.model._param_dict(lik)
            mu, Sigma = self.model.engine.predict_joint(params, self.x, self.D, cache=not self.model.training)
            if lik is None:   # q(f) itself: remove the noise again
                Sigma.diagonal().sub_(torch.nn.functional.softplus(params["raw_noise"].reshape(())) + 1e-4)
            self._mu, self._Sigma, self._root = mu, Sigma, None

    @property
    def covariance_matrix(self):
        self._ensure_joint()
        return self._Sigma

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        """mean + chol(Sigma) eps: exact Cholesky root on the GPU (fp64 blocked MFMA factorisation) where gpytorch switches
        to a Lanczos root above ``max_cholesky_size``.  Shape ``sample_shape + [B(p+1)]``."""
        self._ensure_joint()
        eng = self.model.engine
        if self._root is None:
            self._root = eng.covariance_root(self._Sigma)
        n_out = self._mu.shape[0]
        sample_shape = torch.Size(sample_shape)
        n = int(sample_shape.numel()) if len(sample_shape) else 1
        if base_samples is None:
            eps = torch.randn(n, n_out, dtype=self._mu.dtype, device=self._mu.device)
        else:
            eps = base_samples.reshape(n, n_out).to(self._mu.dtype)
        return eng.draw(self._mu, self._root, eps).reshape(tuple(sample_shape) + (n_out,))

    def sample(self, sample_shape=torch.Size(), base_samples=None):
        with torch.no_grad():
            return self.rsample(sample_shape, base_samples)


class _ApproximateMLL(torch.nn.Module):
    mll_type = "ELBO"

    def __init__(self, likelihood, model, num_data, beta=1.0):
        super().__init__()
        if beta != 1.0:
            raise NotImplementedError("beta != 1 is not used by the reference harness")
        self.likelihood, self.model, self.num_data = likelihood, model, num_data

    def forward(self, output, target):
        if not isinstance(output, PredictiveDistribution) or output.likelihood is None:
            raise TypeError("mll expects likelihood(model(x, derivative_directions=D)) as in directional_vi.py:245")
        model = output.model
        plist = model._param_list(self.likelihood)
        dp = getattr(model, "data_parallel", None)
        elbo, mu, varn = _ElboFunction.apply(model.engine, output.x, target, output.D, float(self.num_data),
                                             self.mll_type, dp, model._param_names(), *plist)
        # the ELBO fast path does not form per-output variances; they are produced on demand
        output._mu, output._varn = mu, (varn if varn.numel() else None)
        return elbo


    @torch.no_grad()
    def backward_step(self, output, target):
        """``loss = -mll(output, target); loss.backward()`` of the reference loop (directional_vi.py:245-247) without the
        autograd round trip: the engine's gradient buffers become the parameters' ``.grad`` directly (the chain
        loss -> mll multiplies them by (-1)(-1) = 1, so the numbers are the same; six tiny elementwise launches and one pass
        over the 36 MB L_S gradient fewer per step).  Returns the loss as a device scalar."""
        if not isinstance(output, PredictiveDistribution) or output.likelihood is None:
            raise TypeError("mll expects likelihood(model(x, derivative_directions=D)) as in directional_vi.py:245")
        model = output.model
        plist = model._param_list(self.likelihood)
        names = model._param_names()
        pd = dict(zip(names, [_ops.detach_keep(p) for p in plist]))
        dp = getattr(model, "data_parallel", None)
        if dp is not None:
            loss, grads, mu, varn = dp.loss_and_grads(model.engine, pd, output.x, target, output.D, float(self.num_data),
                                                      self.mll_type)
        else:
            loss, grads, mu, varn = model.engine.loss_and_grads(pd, output.x, target, output.D, float(self.num_data),
                                                                self.mll_type)
        for p, k in zip(plist, names):
            if isinstance(p, torch.nn.Parameter) and p.requires_grad:
                g = grads[k].view_as(p)
                p.grad = g if p.grad is None else p.grad.add_(g)
        output._mu, output._varn = mu, (varn if varn.numel() else None)
        return loss


class VariationalELBO(_ApproximateMLL):
    mll_type = "ELBO"


class PredictiveLogLikelihood(_ApproximateMLL):
    mll_type = "PLL"


class ApproximateGP(torch.nn.Module):
    def __init__(self, variational_strategy):
        super().__init__()
        self.variational_strategy = variational_strategy
        self._engine = None

    @property
    def engine(self):
        Z = self.variational_strategy.inducing_points
        cls = ElboEngine
        if Z.dtype == torch.float64:        # model built under torch.set_default_dtype(torch.float64) (exp_script.py:56)
            from ._step64 import ElboEngine64 as cls
        if self._engine is None or self._engine.device != Z.device or type(self._engine) is not cls:
            self._engine = cls(Z.device)
        return self._engine

    def variational_parameters(self):
        for mod in self.modules():
            if isinstance(mod, _VariationalDistribution):
                for p in mod.parameters(recurse=False):
                    yield p

    def hyperparameters(self):
        for mod in self.modules():
            if not isinstance(mod, _VariationalDistribution):
                for p in mod.parameters(recurse=False):
                    yield p

    def __call__(self, inputs, prior=False, **kwargs):
        if inputs.dim() == 1:
            inputs = inputs.unsqueeze(-1)
        return self.variational_strategy(inputs, prior=prior, **kwargs)
