"""DSVGP with SHARED inducing directions -- drop-in mirror of the reference ``directionalvi/shared_directional_vi.py``
(``GPModel`` :25-63, ``train_gp`` as in ``directional_vi`` except that the canonical inducing directions are not tiled
when ``inducing_data_initialization=False``, :150-155).  Same engine, ``shared_directions=True``.

With ``inducing_data_initialization=True`` the reference still tiles the directions (:144-145), so its model takes
``num_directions = M p`` and its forward assertion fails; that behaviour is kept.
"""
import torch

from . import directional_vi as _dvi
from .SharedDirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy
from .directional_vi import eval_gp, select_cols_of_y  # noqa: F401
from .gp_shim import CholeskyVariationalDistribution, ConstantMean, ScaleKernel
from .RBFKernelDirectionalGrad import RBFKernelDirectionalGrad


class GPModel(_dvi.GPModel):
    def __init__(self, inducing_points, inducing_directions, dim, learn_inducing_locations=True, **kwargs):
        torch.nn.Module.__init__(self)
        if kwargs.get("variational_distribution") == "NGD" or kwargs.get("variational_strategy") == "CIQ":
            raise NotImplementedError("shared directions are built for the Cholesky strategy / distribution only")
        self.num_inducing = len(inducing_points)
        self.num_directions = len(inducing_directions)                  # shared set (:31-32)
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
    if use_ngd or use_ciq:
        raise NotImplementedError("shared directions are built for the Cholesky strategy / distribution only")
    return _dvi.train_gp(train_dataset, num_inducing, num_directions, minibatch_size, minibatch_dim, num_epochs,
                         learning_rate_hypers, learning_rate_ngd, inducing_data_initialization, False, False, lr_sched,
                         mll_type, num_contour_quadrature, watch_model, gamma, verbose, fixed_inducing_locations,
                         _model_class=GPModel, _shared=True, **args)
