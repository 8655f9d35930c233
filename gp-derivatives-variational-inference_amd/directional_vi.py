"""DSVGP training / evaluation harness -- drop-in mirror of the reference
``directionalvi/directional_vi.py`` (``GPModel`` :25-65, ``select_cols_of_y`` :68-90, ``train_gp`` :93-268,
``eval_gp`` :271-305): same names, arguments, defaults, return types and interleaved output ordering.

What is different underneath (MI355X-first):
  * the dataset is made resident in HBM once; each step's minibatch is gathered on the GPU
    (``dsvgp_gather_batch``) from a per-epoch ``randperm`` instead of ``DataLoader``'s per-index collate
    plus a host->device copy per step (directional_vi.py:133,229-232);
  * ``likelihood(model(x))`` + ``mll`` + ``backward`` run as one fused HIP forward/backward (``_step.py``);
  * both Adam optimizers are the fused HIP Adam (``optim.FusedAdam``), stepped with the same
    per-iteration LR schedulers as the reference (:251-254);
  * under ``torch.distributed`` (one process per GPU) every global minibatch is sharded by rows and
    gradients are reduced with one RCCL all-reduce (``parallel.DataParallel``).
NGD / CIQ variants (``use_ngd`` / ``use_ciq``) are outside this hot path and raise.
"""
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import _ops
from .DirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy
from .RBFKernelDirectionalGrad import RBFKernelDirectionalGrad
from .gp_shim import (ApproximateGP, CholeskyVariationalDistribution, ConstantMean, GaussianLikelihood,
                      PredictiveLogLikelihood, ScaleKernel, VariationalELBO)
from .optim import FusedAdam
from .parallel import DataParallel


class GPModel(ApproximateGP):
    def __init__(self, inducing_points, inducing_directions, dim, learn_inducing_locations=True, **kwargs):
        torch.nn.Module.__init__(self)
        self.num_inducing = len(inducing_points)
        self.num_directions = int(len(inducing_directions) / self.num_inducing)  # num directions per point
        num_directional_derivs = self.num_directions * self.num_inducing
        if kwargs.get("variational_distribution") == "NGD" or kwargs.get("variational_strategy") == "CIQ":
            raise NotImplementedError("NGD / CIQ variants are outside the MI355X DSVGP hot path (SURVEY.md 8f)")
        variational_distribution = CholeskyVariationalDistribution(self.num_inducing + num_directional_derivs)
        variational_strategy = DirectionalGradVariationalStrategy(
            self, inducing_points, inducing_directions, variational_distribution,
            learn_inducing_locations=learn_inducing_locations)
        self.variational_strategy = variational_strategy
        self._engine = None
        self.data_parallel = None
        self.mean_module = ConstantMean()
        self.covar_module = ScaleKernel(RBFKernelDirectionalGrad())

    # --- parameter plumbing for the HIP engine (order = _step.PARAM_NAMES) ---
    def _param_list(self, likelihood=None):
        vs = self.variational_strategy
        vd = vs._variational_distribution
        raw_noise = (likelihood.noise_covar.raw_noise if likelihood is not None
                     else torch.zeros(1, device=vs.inducing_points.device))
        return [vs.inducing_points, vs.inducing_directions, vd.variational_mean, vd.chol_variational_covar,
                self.mean_module.constant, self.covar_module.raw_outputscale,
                self.covar_module.base_kernel.raw_lengthscale, raw_noise]

    def _param_dict(self, likelihood=None):
        from ._step import PARAM_NAMES
        return {k: v.detach() for k, v in zip(PARAM_NAMES, self._param_list(likelihood))}


def select_cols_of_y(y_batch, minibatch_dim, dim):
    """
    randomly select columns of y to train on, but always select function values as part of the batch
    (reference directional_vi.py:68-90).  Returns (y_batch[:, idx], canonical derivative directions).
    """
    idx_y = random.sample(range(1, dim + 1), minibatch_dim)  # ensures unique entries
    idx_y += [0]  # append 0 to the list for function values
    idx_y.sort()
    y_batch = y_batch[:, idx_y]
    E_canonical = torch.eye(dim).to(y_batch.device)
    derivative_directions = E_canonical[np.array(idx_y[1:]) - 1]
    return y_batch, derivative_directions


def _dataset_tensors(dataset, device):
    """Materialise a torch Dataset of (x[d], y[d+1]) pairs as two HBM-resident float32 matrices."""
    tensors = getattr(dataset, "tensors", None)
    if tensors is not None and len(tensors) == 2:
        X, Y = tensors
    else:
        xs, ys = zip(*[dataset[i] for i in range(len(dataset))])
        X, Y = torch.stack(xs), torch.stack(ys)
    return (X.to(device=device, dtype=torch.float32).contiguous(),
            Y.to(device=device, dtype=torch.float32).contiguous())


def train_gp(train_dataset, num_inducing=128,
             num_directions=1, minibatch_size=1, minibatch_dim=1, num_epochs=1,
             learning_rate_hypers=0.01, learning_rate_ngd=0.1,
             inducing_data_initialization=True,
             use_ngd=False,
             use_ciq=False,
             lr_sched=None,
             mll_type="ELBO",
             num_contour_quadrature=15,
             watch_model=False, gamma=0.1,
             verbose=True,
             fixed_inducing_locations=None,
             **args):
    """Train a Derivative GP with the Directional Derivative Variational Inference method
    (argument meaning identical to the reference, directional_vi.py:106-129).

    Extra keyword arguments understood through ``**args`` (all optional, ignored by the reference):
      ``seed`` (int): seeds the minibatch permutation and the derivative-column sampling (required to be
      equal on all ranks under torch.distributed; broadcast from rank 0 when omitted);
      ``max_steps`` (int): stop after this many optimisation steps.
    """
    assert num_directions == minibatch_dim
    if use_ngd or use_ciq:
        raise NotImplementedError("NGD / CIQ variants are outside the MI355X DSVGP hot path (SURVEY.md 8f)")
    if not torch.cuda.is_available():
        raise RuntimeError("train_gp needs an MI355X (HIP) device: the DSVGP hot path has no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())

    dim = len(train_dataset[0][0])
    n_samples = len(train_dataset)
    num_data = (dim + 1) * n_samples                                      # :136
    X, Y = _dataset_tensors(train_dataset, device)

    if inducing_data_initialization is True:
        inducing_points = X[:num_inducing].clone()                        # first M data rows, :140-145
    else:
        inducing_points = torch.rand(num_inducing, dim).to(device)        # :149
    inducing_directions = torch.eye(dim)[:num_directions].repeat(num_inducing, 1).to(device)

    learn_inducing_locations = True
    if fixed_inducing_locations is not None:
        inducing_points = fixed_inducing_locations.to(device)
        learn_inducing_locations = False

    model = GPModel(inducing_points, inducing_directions, dim, learn_inducing_locations=learn_inducing_locations)
    likelihood = GaussianLikelihood()
    model = model.to(device)
    likelihood = likelihood.to(device)
    model.train()
    likelihood.train()

    dp = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dp = DataParallel()
        model.data_parallel = dp
    seed = args.get("seed")
    if seed is None:
        seed_t = torch.randint(0, 2 ** 31 - 1, (1,), device=device)
        if dp is not None:
            dist.broadcast(seed_t, 0)
        seed = int(seed_t.item()) if dp is not None else None
    col_rng = random.Random(seed) if seed is not None else random      # reference: global `random`
    perm_gen = torch.Generator(device=device)
    if seed is not None:
        perm_gen.manual_seed(seed)
    else:
        perm_gen.seed()
    if dp is not None:   # identical initial q(u) on every rank
        model.variational_strategy._maybe_init()
        for t in model._param_list(likelihood):
            dist.broadcast(t.data, 0)

    variational_optimizer = FusedAdam([{"params": list(model.variational_parameters())}], lr=learning_rate_hypers)
    hyperparameter_optimizer = FusedAdam([
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
        mll = VariationalELBO(likelihood, model, num_data=num_data)
    elif mll_type == "PLL":
        mll = PredictiveLogLikelihood(likelihood, model, num_data=num_data)
    else:
        raise ValueError("mll_type must be 'ELBO' or 'PLL'")

    ctx = _ops.Context.get(device)
    E_canonical = torch.eye(dim, device=device)
    max_steps = args.get("max_steps")
    total_step = 0
    loss = None
    for i in range(num_epochs):
        perm = torch.randperm(n_samples, device=device, generator=perm_gen)     # DataLoader(shuffle=True)
        for start in range(0, n_samples, minibatch_size):
            idx = perm[start:start + minibatch_size]
            nb_global = idx.shape[0]
            if dp is not None:
                lo, hi = dp.shard_bounds(nb_global)
                idx = idx[lo:hi]
                dp.global_batch = nb_global
            # select random columns of y to train on (function values always included), :68-90
            idx_y = sorted(col_rng.sample(range(1, dim + 1), minibatch_dim) + [0])
            cols = torch.tensor(idx_y, dtype=torch.int32, device=device)
            nb = idx.shape[0]
            x_batch = torch.empty(nb, dim, dtype=torch.float32, device=device)
            y_batch = torch.empty(nb * (minibatch_dim + 1), dtype=torch.float32, device=device)
            _ops.gather_batch(ctx, X, Y, idx.contiguous(), cols, minibatch_dim, x_batch, y_batch)  # interleaved y, :241
            derivative_directions = E_canonical[np.array(idx_y[1:]) - 1]
            kwargs = {"derivative_directions": derivative_directions.repeat(nb, 1)}             # :238

            variational_optimizer.zero_grad()
            hyperparameter_optimizer.zero_grad()
            output = likelihood(model(x_batch, **kwargs))
            loss = -mll(output, y_batch)
            loss.backward()
            variational_optimizer.step()
            variational_scheduler.step()
            hyperparameter_optimizer.step()
            hyperparameter_scheduler.step()
            if total_step % 50 == 0 and verbose:
                means = output.mean[::num_directions + 1]
                stds = output.variance.sqrt()[::num_directions + 1]
                nll = -torch.distributions.Normal(means, stds).log_prob(y_batch[::num_directions + 1]).mean()
                print(f"Epoch: {i}; total_step: {total_step}, loss: {loss.item()}, nll: {nll}")
                sys.stdout.flush()
            total_step += 1
            if max_steps is not None and total_step >= max_steps:
                break
        if max_steps is not None and total_step >= max_steps:
            break

    if verbose and loss is not None:
        print(f"Done! loss: {loss.item()}")
        print("\nDone Training!")
    return model, likelihood


def eval_gp(test_dataset, model, likelihood,
            mll_type="ELBO", num_directions=1, minibatch_size=1, minibatch_dim=1):
    """Predictive means / variances (with likelihood noise) of all (p+1) outputs per test point,
    returned as CPU vectors of length N_test*(p+1), interleaved (reference directional_vi.py:271-305)."""
    assert num_directions == minibatch_dim
    dim = len(test_dataset[0][0])
    device = model.variational_strategy.inducing_points.device
    X, _ = _dataset_tensors(test_dataset, device)
    n_test = X.shape[0]

    model.eval()
    likelihood.eval()

    means = []
    variances = []
    with torch.no_grad():
        for start in range(0, n_test, minibatch_size):
            x_batch = X[start:start + minibatch_size]
            # redo derivative directions b/c batch size is not consistent
            derivative_directions = torch.eye(dim, device=device)[:num_directions]
            derivative_directions = derivative_directions.repeat(len(x_batch), 1)
            preds = likelihood(model(x_batch, derivative_directions=derivative_directions))
            means.append(preds.mean.cpu())
            variances.append(preds.variance.cpu())
    means = torch.cat(means) if means else torch.zeros(0)
    variances = torch.cat(variances) if variances else torch.zeros(0)
    print("Done Testing!")
    return means, variances
