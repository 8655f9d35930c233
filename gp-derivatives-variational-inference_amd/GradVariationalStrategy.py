"""Whitened variational strategy for the full-gradient SVGP -- HIP-backed mirror of the reference plugin
``directionalvi/GradVariationalStrategy.py`` (constructor :65-68, ``forward`` :87-137).

The reference builds one joint ``(M+B)(d+1)`` ``RBFKernelGrad`` matrix over ``cat([Z, x])`` and slices it
(:89-99); mathematically that is the directional kernel with ``p = d`` and the canonical directions
``I_d`` at every inducing and data point, so the same HIP kernels serve it.
Differences honoured here: no learnable directions, ``psd_safe_cholesky`` default (fp64) jitter ladder
1e-8 * 10^t (:72), a single triangular solve (:113).
"""
import torch

from . import _ops
from .gp_shim import PredictiveDistribution


class GradVariationalStrategy(torch.nn.Module):
    def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=True):
        super().__init__()
        object.__setattr__(self, "model", model)
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

    def _maybe_init(self):
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
        dim = self.inducing_points.size(1)
        if x.size(-1) != dim:
            raise RuntimeError("input dimension %d does not match the inducing points (%d)" % (x.size(-1), dim))
        D = torch.eye(dim, device=x.device, dtype=x.dtype).repeat(x.size(-2), 1)     # RBFKernelGrad: all d partials
        if D.is_cuda:        # (said to the engine as an index list too: the both-sides one-hot assembly kernels, _ops.state_directions)
            _ops.state_directions(D, _ops.index_range(D.device, dim), 0)
        return PredictiveDistribution(self.model, x, D)

    def __call__(self, x, prior=False, **kwargs):
        if prior:
            raise NotImplementedError("prior=True is only used by the legacy un-whitened checkpoint path")
        if self.training:
            self._maybe_init()
        return self.forward(x, **kwargs)
