"""Whitened variational strategy for DSVGP -- HIP-backed mirror of the reference plugin
``directionalvi/DirectionalGradVariationalStrategy.py`` (constructor :65-69, ``forward`` :89-208,
``__call__`` :210-240).

Parameter / buffer names are the reference's (``inducing_points``, ``inducing_directions``,
``_variational_distribution.*``, ``updated_strategy``, ``variational_params_initialized``) so its
``state_dict`` checkpoints load.  ``forward`` returns a handle with ``.mean`` / ``.variance`` of length
``B(p+1)`` (interleaved); the arithmetic is ``_step.ElboEngine`` (Cholesky of K_ZZ + 1e-3 I in fp64,
panel solve on MFMA, variance diag), the lazy ``K_XX`` is never materialised (only its diagonal is used).
"""
import warnings

import torch

from .gp_shim import OldVersionWarning, PredictiveDistribution, PriorDistribution


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

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        self._init_known = False
        self._updated_known = None
        # reference ``_ensure_updated_strategy_flag_set`` (DGVS.py:17-29, registered at :68): a checkpoint without the flag was
        # written by the un-whitened VariationalStrategy of an older gpytorch -- mark it, the first call converts q(u)
        if ("updated_strategy" in self._buffers and prefix + "updated_strategy" not in state_dict
                and any(k.startswith(prefix) for k in state_dict)):
            device = state_dict[list(state_dict.keys())[0]].device
            state_dict[prefix + "updated_strategy"] = torch.tensor(False, device=device)
            warnings.warn(
                "You have loaded a variational GP model (using `VariationalStrategy`) from a previous version of "
                "GPyTorch. We have updated the parameters of your model to work with the new version of "
                "`VariationalStrategy` that uses whitened parameters.\nYour model will work as expected, but we "
                "recommend that you re-save your model.", OldVersionWarning)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _strategy_is_updated(self):
        # (device buffer for state_dict compatibility, mirrored on the host once read)
        if getattr(self, "_updated_known", None) is None:
            # (the CIQ strategy registers no such buffer, reference CiqDGVS.py:160-162: nothing to convert)
            self._updated_known = bool(self._buffers["updated_strategy"].item()) if "updated_strategy" in self._buffers else True
        return self._updated_known

    def _whiten_legacy_parameters(self):
        """reference ``__call__`` (DGVS.py:210-240): change the variational parameters of a legacy checkpoint to be whitened"""
        vd = self._variational_distribution
        if getattr(self, "shared_directions", False):
            raise NotImplementedError("legacy (un-whitened) checkpoints of the shared-directions variant are not converted")
        if not hasattr(vd, "chol_variational_covar"):
            raise NotImplementedError("legacy (un-whitened) checkpoints are converted for a CholeskyVariationalDistribution only")
        with torch.no_grad():
            params = self.model._param_dict(None)
            m_w, L_w = self.model.engine.whiten_legacy(params)
            vd.variational_mean.copy_(m_w)              # initialize_variational_distribution(whitened q(u)), mean_init_std = 0
            vd.chol_variational_covar.copy_(L_w)
            self.variational_params_initialized.fill_(1)
            self._init_known = True
            self.updated_strategy.fill_(True)           # mark that we have updated the variational strategy
            self._updated_known = True

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
            # _VariationalStrategy.__call__(prior=True) -> model.forward(x): only ever evaluated at the inducing points (the
            # legacy conversion above); the inducing directions stand in for the kwargs the reference does not pass
            loc, K = self.model.engine.prior_moments(self.model._param_dict(None))
            return PriorDistribution(loc, K)
        if not self._strategy_is_updated():
            self._whiten_legacy_parameters()
        if self.training:
            self._maybe_init()
        return self.forward(x, **kwargs)
