"""MI355X-native DSVGP hot path (directional-derivative SVGP minibatch ELBO step).

Drop-in for the reference's ``directional_vi.train_gp`` / ``eval_gp`` / ``GPModel`` /
``DirectionalGradVariationalStrategy`` / ``RBFKernelDirectionalGrad`` on top of hand-written HIP
kernels for gfx950 behind the C ABI in ``include/dsvgp.h``.  Importing this package loads
``libdsvgp_hip.so`` and raises if it is missing -- there is no CPU or eager-PyTorch fallback.

The directory name contains hyphens; import it as ``import dsvgp_amd`` (alias module at the repo
root) or ``importlib.import_module("gp-derivatives-variational-inference_amd")``.
"""
from . import _lib, _ops, _step, _step64, gp_shim, optim, parallel  # noqa: F401
from . import RBFKernelDirectionalGrad as _rbf_mod
from . import DirectionalGradVariationalStrategy as _dgvs_mod
from . import directional_vi  # noqa: F401
from . import GradVariationalStrategy as _gvs_mod  # noqa: F401
from . import grad_svgp  # noqa: F401
from . import dfree_directional_vi  # noqa: F401
from . import shared_directional_vi  # noqa: F401
from . import traditional_vi  # noqa: F401
from ._step import ElboEngine, NotPSDError, NGD_PARAM_NAMES, PARAM_NAMES  # noqa: F401
from ._step64 import ElboEngine64  # noqa: F401
from .directional_vi import GPModel, TrainLoop, eval_gp, select_cols_of_y, setup_training, train_gp  # noqa: F401
from .gp_shim import (GaussianLikelihood, NaturalVariationalDistribution, PredictiveLogLikelihood,  # noqa: F401
                      VariationalELBO)
from .optim import NGD, FusedAdam  # noqa: F401
from .parallel import DataParallel  # noqa: F401

__version__ = "0.1.0"
