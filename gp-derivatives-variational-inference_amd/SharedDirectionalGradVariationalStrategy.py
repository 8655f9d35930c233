"""Shared-inducing-directions variant of the whitened DSVGP strategy -- HIP-backed mirror of the reference plugin
``directionalvi/SharedDirectionalGradVariationalStrategy.py`` (same class name as there: ``shared_directional_vi`` imports
it under the name ``DirectionalGradVariationalStrategy``, shared_directional_vi.py:13).

ONE set of p directions (``inducing_directions`` [p, d]) is tiled over the M inducing points (:95-98), q(u) covers the M
function values plus p shared derivative values (:99-107), and the middle term of the predictive covariance is zero as the
reference computes it (:210-212): ``Sigma = K_XX + 1e-4 I``.  Engine flag ``shared_directions``.
"""
from .DirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy as _Base
from .gp_shim import PredictiveDistribution


class DirectionalGradVariationalStrategy(_Base):
    shared_directions = True

    def forward(self, x, inducing_points=None, inducing_values=None, variational_inducing_covar=None, **kwargs):
        derivative_directions = kwargs["derivative_directions"]
        num_directions = self.inducing_directions.size(-2)            # shared: p rows in total (:94-96)
        num_data = x.size(-2)
        num_derivative_directions = int(derivative_directions.size(-2) / num_data)
        assert num_derivative_directions == num_directions, \
            "Need minibatch dim to be same as number of directions for kernel"
        self.model.covar_module.base_kernel.set_num_directions(num_directions)
        return PredictiveDistribution(self.model, x, derivative_directions.to(x.device))
