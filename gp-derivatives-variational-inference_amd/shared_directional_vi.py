"""DSVGP with SHARED inducing directions -- drop-in mirror of the reference ``directionalvi/shared_directional_vi.py``
(``GPModel`` :25-63, ``train_gp`` as in ``directional_vi`` except that the canonical inducing directions are not tiled
when ``inducing_data_initialization=False``, :150-155).  Same engine, ``shared_directions=True``.

With ``inducing_data_initialization=True`` the reference still tiles the directions (:144-145), so its model takes
``num_directions = M p`` and its forward assertion fails; that behaviour is kept.

``use_ngd=True`` (:37-39,170-171,188-189) swaps q(u) over the M + p shared values to a NaturalVariationalDistribution + NGD.
``use_ciq=True`` (:48-50,166-169) hands the SHARED direction set [p, d] and the (M + p)-dimensional q(u) to the ordinary
``CiqDirectionalGradVariationalStrategy``, whose forward (CiqDGVS.py:197-231) derives ``num_directions = p / M`` from them and
builds an M(p / M + 1)-dimensional system: the shapes do not agree and the reference's own call fails.  Nothing to mirror
there; this module refuses the combination up front.
"""
import torch

from . import directional_vi as _dvi
from .SharedDirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy
from .directional_vi import eval_gp, select_cols_of_y  # noqa: F401
from .gp_shim import CholeskyVariationalDistribution, ConstantMean, NaturalVariationalDistribution, ScaleKernel
from .RBFKernelDirectionalGrad import RBFKernelDirectionalGrad


class GPModel(_dvi.GPModel):
    def __init__(self, inducing_points, inducing_directions, dim, learn_inducing_locations=True, **kwargs):
        torch.nn.Module.__init__(self)
        if kwargs.get("variational_strategy") == "CIQ":
            raise NotImplementedError("shared directions with the CIQ strategy: the reference's combination is shape-"
                                      "inconsistent (see the module docstring) and is not built")
        self.num_inducing = len(inducing_points)
        self.num_directions = len(inducing_directions)                  # shared set (:31-32)
        if kwargs.get("variational_distribution") == "NGD":             # :37-39
            variational_distribution = NaturalVariationalDistribution(self.num_inducing + self.num_directions)
        else:
            variational_distribution = CholeskyVariationalDistribution(self.num_inducing + self.num_directions)
        self._ciq = False
        self.variational_strategy = DirectionalGradVariationalStrategy(
            self, inducing_points, inducing_directions, variational_distribution,
            learn_inducing_locations=learn_inducing_locations)
        self._engine = None
        self.data_parallel = None
        self.mean_module = ConstantMean()
        self.covar_module = ScaleKernel(RBFKernelDirectionalGrad())

    @property
    def engine(self):
        eng = _dvi.GPModel.engine.fget(self)
        eng.shared_directions = True
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
             fixed_inducing_locations=None,
             **args):
    """Argument meaning identical to the reference (shared_directional_vi.py:93-130)."""
    if use_ciq:
        raise NotImplementedError("shared directions with the CIQ strategy: the reference's combination is shape-"
                                  "inconsistent (see the module docstring) and is not built")
    return _dvi.train_gp(train_dataset, num_inducing, num_directions, minibatch_size, minibatch_dim, num_epochs,
                         learning_rate_hypers, learning_rate_ngd, inducing_data_initialization, use_ngd, False, lr_sched,
                         mll_type, num_contour_quadrature, watch_model, gamma, verbose, fixed_inducing_locations,
                         _model_class=GPModel, _shared=True, **args)
