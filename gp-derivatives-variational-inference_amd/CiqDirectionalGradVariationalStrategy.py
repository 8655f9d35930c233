"""CIQ-whitened variational strategy for DSVGP -- HIP-backed mirror of the reference plugin
``directionalvi/CiqDirectionalGradVariationalStrategy.py`` (constructor :160-162, ``forward`` :197-295,
``kl_divergence`` :297-312, ``__call__`` :314-378).

Same kernels as ``DirectionalGradVariationalStrategy``; the whitening ``K_ZZ^{-1/2} K_ZX`` is evaluated by
contour-integral quadrature + msMINRES (``lazify(K_ZZ).sqrt_inv_matmul(K_ZX)``, :255-256) instead of a Cholesky
factor, and with a ``NaturalVariationalDistribution`` the mean / variance interpolation terms and the
expectation-parameter gradients come from ``_NgdInterpTerms`` (:19-123).  The arithmetic is
``_step.ElboEngine._ciq_step`` on ``csrc/ciq.hip``; this class keeps the reference's parameter names and the
natural-parameter initialisation of its ``__call__`` (:320-326).
"""
import torch

from .DirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy
from .gp_shim import NaturalVariationalDistribution


class CiqDirectionalGradVariationalStrategy(DirectionalGradVariationalStrategy):
    def __init__(self, model, inducing_points, inducing_directions, variational_distribution,
                 learn_inducing_locations=True):
        super().__init__(model, inducing_points, inducing_directions, variational_distribution,
                         learn_inducing_locations=learn_inducing_locations)
        del self._buffers["updated_strategy"]          # the reference's CIQ strategy registers no such buffer (:160-162)

    def _ngd(self):
        return isinstance(self._variational_distribution, NaturalVariationalDistribution)

    def _maybe_init(self):
        if self._init_known:
            return
        if not self.variational_params_initialized.item():
            vd = self._variational_distribution
            if self._ngd():                                # :320-326
                with torch.no_grad():
                    vd.natural_vec.copy_(torch.randn_like(vd.natural_vec).mul_(1e-3))
                    vd.natural_mat.copy_(torch.eye(vd.natural_vec.shape[0], device=vd.natural_vec.device).mul_(-0.5))
            else:
                vd.initialize_variational_distribution()
            self.variational_params_initialized.fill_(1)
        self._init_known = True

    def kl_divergence(self):
        """With NGD the KL value is not formed in the forward pass (the reference memoises zeros, :74,271-272); its
        gradient reaches the natural parameters inside the engine."""
        if self._ngd():
            return torch.zeros((), device=self.inducing_points.device)
        raise NotImplementedError("CIQ with a Cholesky variational distribution is not reachable from train_gp")
