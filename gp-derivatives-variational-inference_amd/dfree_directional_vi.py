"""DSVGP trained on derivative-FREE data -- drop-in mirror of the reference ``directionalvi/dfree_directional_vi.py``
(``GPModel`` :26-63, ``train_gp`` :66-262, ``eval_gp`` :265-292): the dataset yields ``(x[d], y)`` with scalar function
values, the model still learns inducing directional derivatives.  Same engine, ``data_outputs="values"``.
Differences from ``directional_vi`` that the reference has and are kept: no ``fixed_inducing_locations`` argument, the
derivative directions handed to the model are the first p canonical ones (:224-227, no column sampling),
``num_data = (dim+1) n`` (:130).

``use_ciq=True`` (:45-47,154-156) selects the ORDINARY ``CiqDirectionalGradVariationalStrategy`` in the reference (there is no
derivative-free CIQ strategy): its forward returns all B (p + 1) outputs (CiqDGVS.py:197-295) while the targets are the B function
values, so the reference's own loss evaluation fails on the shapes.  Nothing to mirror; refused up front here.
"""
import sys

import torch

from . import directional_vi as _dvi
from .DFreeDirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy
from .gp_shim import GaussianLikelihood  # noqa: F401


class GPModel(_dvi.GPModel):
    _strategy_class = DirectionalGradVariationalStrategy

    def __init__(self, inducing_points, inducing_directions, dim, **kwargs):
        kwargs.pop("learn_inducing_locations", None)
        if kwargs.get("variational_strategy") == "CIQ":
            raise NotImplementedError("derivative-free data with the CIQ strategy: the reference pairs B (p + 1) CIQ outputs with B "
                                      "targets (see the module docstring); not built")
        super().__init__(inducing_points, inducing_directions, dim, learn_inducing_locations=True, **kwargs)

    @property
    def engine(self):
        eng = _dvi.GPModel.engine.fget(self)
        eng.data_outputs = "values"
        return eng


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
             **args):
    """Train a derivative GP on function values only (argument meaning identical to the reference,
    dfree_directional_vi.py:66-123; ``seed`` / ``max_steps`` as in ``directional_vi.train_gp``)."""
    assert num_directions == minibatch_dim
    if use_ciq:
        raise NotImplementedError("derivative-free data with the CIQ strategy: the reference pairs B (p + 1) CIQ outputs with B "
                                      "targets (see the module docstring); not built")
    loop = _dvi.setup_training(train_dataset, num_inducing, num_directions, minibatch_size, minibatch_dim, num_epochs,
                               learning_rate_hypers, inducing_data_initialization, lr_sched, mll_type, gamma, None,
                               seed=args.get("seed"), use_ngd=use_ngd, learning_rate_ngd=learning_rate_ngd,
                               model_class=GPModel, dfree=True)
    n_samples = loop.X.shape[0]
    max_steps = args.get("max_steps")
    total_step = 0
    loss = None
    for i in range(num_epochs):
        perm = loop.epoch_permutation()
        for start in range(0, n_samples, minibatch_size):
            report = (total_step % 50 == 0) and verbose
            loss, output, y_batch = loop.step(perm[start:start + minibatch_size], need_variance=report)
            if report:        # :244-249 (the reference strides by num_directions+1 here although outputs are values only)
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
    loop.finish()
    if verbose and loss is not None:
        print(f"Done! loss: {loss.item()}")
        print("\nDone Training!")
    return loop.model, loop.likelihood


def eval_gp(test_dataset, model, likelihood,
            mll_type="ELBO", num_directions=1, minibatch_size=1, minibatch_dim=1):
    """Predictive means / variances (with likelihood noise) of the function values, CPU vectors of length N_test
    (reference dfree_directional_vi.py:265-292)."""
    return _dvi.eval_gp(test_dataset, model, likelihood, mll_type, num_directions, minibatch_size, minibatch_dim)
