"""Whitened variational strategy for DSVGP -- HIP-backed mirror of the reference plugin
``directionalvi/DirectionalGradVariationalStrategy.py`` (constructor :65-69, ``forward`` :89-208,
``__call__`` :210-240).

Parameter / buffer names are the reference's (``inducing_points``, ``inducing_directions``,
``_variational_distribution.*``, ``updated_strategy``, ``variational_params_initialized``) so its
``state_dict`` checkpoints load.  ``forward`` returns a handle with ``.mean`` / ``.variance`` of length
``B(p+1)`` (interleaved); the arithmetic is ``_step.ElboEngine`` (Cholesky of K_ZZ + 1e-3 I in fp64,
panel solve on MFMA, variance diag), the lazy ``K_XX`` is never materialised (only its diagonal is used).
"""
import torch

from .gp_shim import PredictiveDistribution


class DirectionalGradVariationalStrategy(torch.nn.Module):
    def __init__(self, model, inducing_points, inducing_directions, variational_distribution,
                 learn_inducing_locations=True):
        super().__init__()
        object.__setattr__(self, "model", model)      # not a submodule (gpytorch does the same)
        inducing_points = inducing_points.clone()
        if inducing_points.dim() == 1:
            inducing_points = inducing_points.unsqueeze(-1)
        if learn_inducing_locations:
            self.register_parameter("inducing_points", torch.nn.Parameter(inducing_points))
        else:
            self.register_buffer("inducing_points", inducing_points)
        self._variational_distribution = variational_distribution
        self.register_buffer("variational_params_initialized", torch.tensor(0))
        self._init_known = False
        self.register_buffer("updated_strategy", torch.tensor(True))
        self.register_parameter("inducing_directions", torch.nn.Parameter(inducing_directions.clone()))

    def _maybe_init(self):
        # gpytorch _VariationalStrategy.__call__: lazy init of q(u) from the (whitened) prior N(0, I)
        # the flag lives in a device buffer (state_dict compatibility); reading it is a device sync, so its value is
        # mirrored on the host once known (a loaded checkpoint invalidates the mirror)
        if self._init_known:
            return
        if not self.variational_params_initialized.item():
            self._variational_distribution.initialize_variational_distribution()
            self.variational_params_initialized.fill_(1)
        self._init_known = True

    def _load_from_state_dict(self, *args, **kwargs):
        self._init_known = False
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, inducing_points=None, inducing_values=None, variational_inducing_covar=None, **kwargs):
        derivative_directions = kwargs["derivative_directions"]
        num_induc = self.inducing_points.size(-2)
        num_directions = int(self.inducing_directions.size(-2) / num_induc)
        num_data = x.size(-2)
        num_derivative_directions = int(derivative_directions.size(-2) / num_data)
        assert num_derivative_directions == num_directions, \
            "Need minibatch dim to be same as number of directions for kernel"
        self.model.covar_module.base_kernel.set_num_directions(num_directions)
        return PredictiveDistribution(self.model, x, derivative_directions.to(x.device))

    def __call__(self, x, prior=False, **kwargs):
        if prior:
            raise NotImplementedError("prior=True is only used by the legacy un-whitened checkpoint path")
        if self.training:
            self._maybe_init()
        return self.forward(x, **kwargs)
