"""Full-gradient SVGP harness -- drop-in mirror of the reference ``directionalvi/grad_svgp.py``
(``GPModel`` :20-39, ``train_gp`` :41-178, ``eval_gp`` :180-202): same names, arguments, defaults and
interleaved outputs of length ``N(d+1)``.  BASELINE config 3 (d=10, N=50k, M=300 -> M(d+1)=3300).

Runs on the same HIP engine as ``directional_vi`` with ``p = d`` and fixed canonical directions
(gpytorch's ``RBFKernelGrad`` == ``RBFKernelDirectionalGrad`` with ``V = I_d``, reference
RBFKernelDirectionalGrad.py:157-161); ``num_data = n_samples`` (:119), all ``d+1`` target columns (:143),
``Z ~ U[0,1]^{M x d}`` (:61), nll print every 25 steps (:160-164).
"""
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import _ops
from .GradVariationalStrategy import GradVariationalStrategy
from .RBFKernelDirectionalGrad import RBFKernelDirectionalGrad
from .directional_vi import TrainLoop, _dataset_tensors
from .gp_shim import (ApproximateGP, CholeskyVariationalDistribution, ConstantMean, GaussianLikelihood,
                      NaturalVariationalDistribution, PredictiveLogLikelihood, ScaleKernel, VariationalELBO)
from .optim import NGD, FusedAdam, make_adam
from .parallel import DataParallel


class GPModel(ApproximateGP):
    def __init__(self, inducing_points, **kwargs):
        torch.nn.Module.__init__(self)
        dim = inducing_points.size(1)
        # gpytorch's plain CiqVariationalStrategy (grad_svgp.py:25-27): the same CIQ whitening + NGD interpolation terms as the
        # directional strategy (whose file quotes its forward, CiqDGVS.py:243-251) with K_ZZ.add_jitter(1e-2) and
        # diag K_XX + 1e-4; no lengthscale re-initialisation
        self._ciq = kwargs.get("variational_strategy") == "CIQ"
        if kwargs.get("variational_distribution") == "NGD":                               # grad_svgp.py:21-22
            variational_distribution = NaturalVariationalDistribution(inducing_points.size(0) * (dim + 1))
        else:
            variational_distribution = CholeskyVariationalDistribution(inducing_points.size(0) * (dim + 1))
        self.variational_strategy = GradVariationalStrategy(self, inducing_points, variational_distribution,
                                                            learn_inducing_locations=True)
        self._engine = None
        self.data_parallel = None
        self.mean_module = ConstantMean()
        self.covar_module = ScaleKernel(RBFKernelDirectionalGrad())      # stands in for gpytorch RBFKernelGrad
        self.register_buffer("_canonical_directions", torch.eye(dim).repeat(inducing_points.size(0), 1),
                             persistent=False)

    @property
    def engine(self):
        eng = ApproximateGP.engine.fget(self)
        eng.chol_jitter = 1e-8          # psd_safe_cholesky default for a double matrix (GradVariationalStrategy.py:72)
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
        V = self._canonical_directions
        if V.is_cuda:        # (fixed unit vectors, the same at every inducing point: stated as an index list, _ops.state_directions)
            _ops.state_directions(V, _ops.index_range(V.device, V.shape[1]), 0)
        return [vs.inducing_points, V] + q + [
            self.mean_module.constant, self.covar_module.raw_outputscale,
            self.covar_module.base_kernel.raw_lengthscale, raw_noise]

    def _param_names(self):
        from ._step import NGD_PARAM_NAMES, PARAM_NAMES
        ngd = isinstance(self.variational_strategy._variational_distribution, NaturalVariationalDistribution)
        return NGD_PARAM_NAMES if ngd else PARAM_NAMES

    def _param_dict(self, likelihood=None):
        return {k: _ops.detach_keep(v) for k, v in zip(self._param_names(), self._param_list(likelihood))}


def setup_training(train_dataset, dim, num_inducing=128, minibatch_size=1, num_epochs=1, use_ngd=False, use_ciq=False,
                   learning_rate_hypers=0.01, learning_rate_ngd=0.1, lr_sched=None, mll_type="ELBO", gamma=0.1,
                   tensors=None, **args):
    """Everything ``train_gp`` does before its loop (grad_svgp.py:41-127); returns a TrainLoop (``tensors``: an
    already-resident (X, Y) pair instead of a Dataset)."""
    if not torch.cuda.is_available():
        raise RuntimeError("train_gp needs an MI355X (HIP) device: this path has no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())
    X, Y = tensors if tensors is not None else _dataset_tensors(train_dataset, device)
    n_samples = X.shape[0]

    inducing_points = torch.rand(num_inducing, dim).to(X)            # :61
    if use_ciq:                                                           # grad_svgp.py:63-65
        model = GPModel(inducing_points=inducing_points, variational_distribution="NGD", variational_strategy="CIQ").to(X)
        model.engine.ciq_num_quadrature = int(args.get("num_contour_quadrature", 15))
    elif use_ngd:                                                         # grad_svgp.py:66-67
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
            dist.broadcast(t.data, 0)

    if use_ngd or use_ciq:                                                # grad_svgp.py:87-88
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
        mll = VariationalELBO(likelihood, model, num_data=n_samples)      # :119
    elif mll_type == "PLL":
        mll = PredictiveLogLikelihood(likelihood, model, num_data=n_samples)
    else:
        raise ValueError("mll_type must be 'ELBO' or 'PLL'")

    return TrainLoop(X, Y, model, likelihood, mll, (variational_optimizer, hyperparameter_optimizer),
                     (variational_scheduler, hyperparameter_scheduler), dim, dp, None, perm_gen, full_gradient=True)


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
    loop = setup_training(train_dataset, dim, num_inducing, minibatch_size, num_epochs, use_ngd, use_ciq,
                          learning_rate_hypers, learning_rate_ngd, lr_sched, mll_type, gamma,
                          num_contour_quadrature=num_contour_quadrature, **args)
    model, likelihood = loop.model, loop.likelihood
    n_samples = loop.X.shape[0]
    max_steps = args.get("max_steps")
    total_step = 0
    loss = None
    for i in range(num_epochs):
        perm = loop.epoch_permutation()
        mini_steps = 0
        for start in range(0, n_samples, minibatch_size):
            report = (total_step % 25 == 0) and verbose
            loss, output, y_batch = loop.step(perm[start:start + minibatch_size], need_variance="values" if report else False)
            if report:
                means = output.mean[::dim + 1]
                stds = output.value_variance.sqrt()          # = output.variance.sqrt()[::dim + 1]
                nll = -torch.distributions.Normal(means, stds).log_prob(y_batch[::dim + 1]).mean()
                print(f"Epoch: {i}; total_step: {mini_steps}, loss: {loss.item()}, nll: {nll}")
            mini_steps += 1
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
    return means, variances
