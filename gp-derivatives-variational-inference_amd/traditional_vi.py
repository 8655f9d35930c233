"""Plain (derivative-free) SVGP harness -- drop-in mirror of the reference ``directionalvi/traditional_vi.py``
(``GPModel`` :17-36, ``train_gp`` :39-150, ``eval_gp`` :153-178; BASELINE config 0, reference
``tests/test_traditional_vi.py``): ``ScaleKernel(RBFKernel)`` + gpytorch's whitened ``VariationalStrategy``.

That is the p = 0 case of the DSVGP step (no directions on either side, M' = M, B' = B): K_ZZ + 1e-3 I, fp64 Cholesky and
solve, ``Sigma = K_XX + 1e-4 I + A^T (S - I) A`` -- the same HIP engine with empty direction sets.  ``num_data = n``
(:98), scalar targets, the every-50-steps nll print without striding (:129-133).  ``use_ngd`` swaps in the
NaturalVariationalDistribution / NGD pair (:19-20,58-59,72-73); gpytorch's own ``CiqVariationalStrategy`` (:22-24) is not built.
"""
import sys

import numpy as np
import torch
import torch.distributed as dist

from .GradVariationalStrategy import GradVariationalStrategy
from .RBFKernelDirectionalGrad import RBFKernelDirectionalGrad
from .directional_vi import TrainLoop, _dataset_tensors
from .gp_shim import (ApproximateGP, CholeskyVariationalDistribution, ConstantMean, GaussianLikelihood,
                      NaturalVariationalDistribution, PredictiveDistribution, PredictiveLogLikelihood, ScaleKernel,
                      VariationalELBO)
from .optim import NGD, FusedAdam, make_adam
from .parallel import DataParallel


class VariationalStrategy(GradVariationalStrategy):
    """gpytorch.variational.VariationalStrategy (whitened) for a kernel without derivative outputs."""

    def forward(self, x, inducing_points=None, inducing_values=None, variational_inducing_covar=None, **kwargs):
        dim = self.inducing_points.size(1)
        if x.size(-1) != dim:
            raise RuntimeError("input dimension %d does not match the inducing points (%d)" % (x.size(-1), dim))
        return PredictiveDistribution(self.model, x, torch.empty(0, dim, device=x.device, dtype=x.dtype))


class GPModel(ApproximateGP):
    def __init__(self, inducing_points, **kwargs):
        torch.nn.Module.__init__(self)
        # gpytorch's plain CiqVariationalStrategy (traditional_vi.py:22-24): CIQ whitening + NGD interpolation terms on the
        # p = 0 path of the same engine, K_ZZ.add_jitter(1e-2) and diag K_XX + 1e-4 (its forward is quoted at CiqDGVS.py:243-251)
        self._ciq = kwargs.get("variational_strategy") == "CIQ"
        if kwargs.get("variational_distribution") == "NGD":                               # :19-20
            variational_distribution = NaturalVariationalDistribution(inducing_points.size(0))
        else:
            variational_distribution = CholeskyVariationalDistribution(inducing_points.size(0))
        self.variational_strategy = VariationalStrategy(self, inducing_points, variational_distribution,
                                                        learn_inducing_locations=True)
        self._engine = None
        self.data_parallel = None
        self.mean_module = ConstantMean()
        self.covar_module = ScaleKernel(RBFKernelDirectionalGrad())      # p = 0: the plain RBF kernel
        self.register_buffer("_no_directions", torch.empty(0, inducing_points.size(1)), persistent=False)

    @property
    def engine(self):
        eng = ApproximateGP.engine.fget(self)
        if getattr(self, "_ciq", False):
            eng.whitening, eng.kzz_jitter, eng.ciq_kxx_jitter = "ciq", 1e-2, 1e-4
        return eng

    def _param_list(self, likelihood=None):
        vs = self.variational_strategy
        vd = vs._variational_distribution
        raw_noise = (likelihood.noise_covar.raw_noise if likelihood is not None
                     else torch.zeros(1, device=vs.inducing_points.device))
        q = ([vd.natural_vec, vd.natural_mat] if isinstance(vd, NaturalVariationalDistribution)
             else [vd.variational_mean, vd.chol_variational_covar])
        return [vs.inducing_points, self._no_directions] + q + [
            self.mean_module.constant, self.covar_module.raw_outputscale,
            self.covar_module.base_kernel.raw_lengthscale, raw_noise]

    def _param_names(self):
        from ._step import NGD_PARAM_NAMES, PARAM_NAMES
        ngd = isinstance(self.variational_strategy._variational_distribution, NaturalVariationalDistribution)
        return NGD_PARAM_NAMES if ngd else PARAM_NAMES

    def _param_dict(self, likelihood=None):
        return {k: v.detach() for k, v in zip(self._param_names(), self._param_list(likelihood))}


def train_gp(train_dataset, dim, num_inducing=128,
             minibatch_size=1,
             num_epochs=1,
             use_ngd=False,
             use_ciq=False,
             learning_rate_hypers=0.01,
             learning_rate_ngd=0.1,
             lr_sched=None,
             mll_type="ELBO",
             num_contour_quadrature=15,
             watch_model=False, gamma=0.1,
             verbose=True,
             **args):
    """Argument meaning identical to the reference (traditional_vi.py:39-52); ``seed`` / ``max_steps`` via ``**args``."""
    if not torch.cuda.is_available():
        raise RuntimeError("train_gp needs an MI355X (HIP) device: this path has no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())
    X, Y = _dataset_tensors(train_dataset, device)
    if Y.dim() == 1:
        Y = Y.reshape(-1, 1).contiguous()
    n_samples = X.shape[0]

    inducing_points = torch.rand(num_inducing, dim).to(X)            # :57
    if use_ciq:                                                           # :59-61
        model = GPModel(inducing_points=inducing_points, variational_distribution="NGD", variational_strategy="CIQ").to(X)
        model.engine.ciq_num_quadrature = int(num_contour_quadrature)
    elif use_ngd:
        model = GPModel(inducing_points=inducing_points, variational_distribution="NGD").to(X)
    else:
        model = GPModel(inducing_points=inducing_points).to(X)
    likelihood = GaussianLikelihood().to(X)
    model.train()
    likelihood.train()

    dp = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dp = DataParallel()
        model.data_parallel = dp
    seed = args.get("seed")
    if seed is None and dp is not None:
        seed_t = torch.randint(0, 2 ** 31 - 1, (1,), device=device)
        dist.broadcast(seed_t, 0)
        seed = int(seed_t.item())
    perm_gen = torch.Generator(device=device)
    perm_gen.manual_seed(seed) if seed is not None else perm_gen.seed()
    model.variational_strategy._maybe_init()
    if dp is not None:
        for t in model._param_list(likelihood):
            if t.numel():
                dist.broadcast(t.data, 0)

    if use_ngd or use_ciq:                                                # :72-73
        variational_optimizer = NGD(list(model.variational_parameters()), num_data=n_samples, lr=learning_rate_ngd)
    else:
        variational_optimizer = make_adam([{"params": list(model.variational_parameters())}], lr=learning_rate_hypers)
    hyperparameter_optimizer = make_adam([
        {"params": list(model.hyperparameters())},
        {"params": list(likelihood.parameters())},
    ], lr=learning_rate_hypers)
    if lr_sched == "step_lr":
        num_batches = int(np.ceil(n_samples / minibatch_size))
        milestones = [int(num_epochs * num_batches / 3), int(2 * num_epochs * num_batches / 3)]
        hyperparameter_scheduler = torch.optim.lr_scheduler.MultiStepLR(hyperparameter_optimizer, milestones, gamma=gamma)
        variational_scheduler = torch.optim.lr_scheduler.MultiStepLR(variational_optimizer, milestones, gamma=gamma)
    else:
        if lr_sched is None:
            lr_sched = lambda epoch: 1.0
        hyperparameter_scheduler = torch.optim.lr_scheduler.LambdaLR(hyperparameter_optimizer, lr_lambda=lr_sched)
        variational_scheduler = torch.optim.lr_scheduler.LambdaLR(variational_optimizer, lr_lambda=lr_sched)

    if mll_type == "ELBO":
        print("Using ELBO")
        mll = VariationalELBO(likelihood, model, num_data=n_samples)      # :98
    elif mll_type == "PLL":
        print("Using PLL")
        mll = PredictiveLogLikelihood(likelihood, model, num_data=n_samples)
    else:
        raise ValueError("mll_type must be 'ELBO' or 'PLL'")

    loop = TrainLoop(X, Y, model, likelihood, mll, (variational_optimizer, hyperparameter_optimizer),
                     (variational_scheduler, hyperparameter_scheduler), 0, dp, None, perm_gen, full_gradient=True)
    loop.plain = True
    max_steps = args.get("max_steps")
    total_step = 0
    loss = None
    for i in range(num_epochs):
        perm = loop.epoch_permutation()
        for start in range(0, n_samples, minibatch_size):
            report = (total_step % 50 == 0) and verbose
            loss, output, y_batch = loop.step(perm[start:start + minibatch_size], need_variance=report)
            if report:
                means = output.mean
                stds = output.variance.sqrt()
                nll = -torch.distributions.Normal(means, stds).log_prob(y_batch).mean()
                print(f"Epoch: {i}; total_step: {total_step}, loss: {loss.item()}, nll: {nll}")
            total_step += 1
            sys.stdout.flush()
            if max_steps is not None and total_step >= max_steps:
                break
        if max_steps is not None and total_step >= max_steps:
            break
    loop.finish()
    if verbose and loss is not None:
        print(f"Done! loss: {loss.item()}")
        print("\nDone Training!")
    sys.stdout.flush()
    return model, likelihood


def eval_gp(test_dataset, model, likelihood, mll_type="ELBO", num_inducing=128, minibatch_size=1):
    """Predictive means / variances (with likelihood noise), CPU vectors of length N_test (traditional_vi.py:153-178)."""
    device = model.variational_strategy.inducing_points.device
    X, _ = _dataset_tensors(test_dataset, device, model.variational_strategy.inducing_points.dtype)
    model.eval()
    likelihood.eval()
    means, variances = [], []
    with torch.no_grad():
        for start in range(0, X.shape[0], minibatch_size):
            preds = likelihood(model(X[start:start + minibatch_size]))
            means.append(preds.mean.cpu())
            variances.append(preds.variance.cpu())
    means = torch.cat(means) if means else torch.zeros(0)
    variances = torch.cat(variances) if variances else torch.zeros(0)
    print("Done Testing!")
    return means, variances
