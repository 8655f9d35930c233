"""Derivative-free-data variant of the whitened DSVGP strategy -- HIP-backed mirror of the reference plugin
``directionalvi/DFreeDirectionalGradVariationalStrategy.py`` (same class name as there: ``dfree_directional_vi`` imports it
under the name ``DirectionalGradVariationalStrategy``, dfree_directional_vi.py:14).

The inducing side keeps its p directional derivatives per point; the data side carries function values only:
``K_ZX = K(Z, x)[:, ::p+1]`` (:119), ``K_XZ = K(x, Z)[::p+1, :]`` (:124), ``K_XX`` sliced the same way (:136), the
constant mean on ``B`` rows (:113).  ``forward`` returns a handle with ``.mean`` / ``.variance`` of length ``B``.
"""
from .DirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy as _Base


class DirectionalGradVariationalStrategy(_Base):
    data_outputs = "values"
