"""Generate tests/golden/kernel_*.npz by executing the REFERENCE kernel file in this container.

Runs only where /root/reference exists (the build container); the GPU box only sees the
committed .npz vectors.  The reference file ``directionalvi/RBFKernelDirectionalGrad.py`` imports
``gpytorch`` (absent here), but its arithmetic only needs three things from its base class:
``self.lengthscale``, ``self.covar_dist(x1, x2, square_dist=True, dist_postprocess_func=...)`` and
``super().forward(..., diag=True)`` (ones).  Those are provided by the throw-away module objects
below -- they are not a gpytorch re-implementation and nothing else in the repo uses them.  The
reference source is imported from where it lies and never copied.

Usage:  python oracle/make_golden.py   (writes tests/golden/)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/directionalvi/RBFKernelDirectionalGrad.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stand_in():
    def postprocess_rbf(dist_mat):
        return dist_mat.div(-2).exp()

    class RBFKernel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._ell = torch.tensor([[0.6931471805599453]], dtype=torch.float64)

        @property
        def lengthscale(self):
            return self._ell

        def covar_dist(self, x1, x2, square_dist=False, dist_postprocess_func=None, **params):
            assert square_dist
            diff = x1.unsqueeze(-2) - x2.unsqueeze(-3)
            return dist_postprocess_func((diff * diff).sum(-1))

        def forward(self, x1, x2, diag=False, **params):
            assert diag
            return torch.ones(x1.shape[-2], dtype=x1.dtype)

    gp = types.ModuleType("gpytorch")
    lazy = types.ModuleType("gpytorch.lazy")
    kron = types.ModuleType("gpytorch.lazy.kronecker_product_lazy_tensor")
    kron.KroneckerProductLazyTensor = object
    kernels = types.ModuleType("gpytorch.kernels")
    rbf = types.ModuleType("gpytorch.kernels.rbf_kernel")
    rbf.RBFKernel = RBFKernel
    rbf.postprocess_rbf = postprocess_rbf
    for name, mod in [("gpytorch", gp), ("gpytorch.lazy", lazy),
                      ("gpytorch.lazy.kronecker_product_lazy_tensor", kron),
                      ("gpytorch.kernels", kernels), ("gpytorch.kernels.rbf_kernel", rbf)]:
        sys.modules[name] = mod


def load_reference_kernel():
    _install_stand_in()
    spec = importlib.util.spec_from_file_location("_ref_rbf_dirgrad", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.RBFKernelDirectionalGrad


# (name, n1, n2, d, p, lengthscale, same_inputs, one_hot_v2)
CASES = [
    ("a", 4, 3, 5, 2, 0.6931471805599453, False, False),
    ("b_sym", 3, 3, 2, 2, 0.6931471805599453, True, False),
    ("c_onehot", 5, 7, 20, 5, 1.3, False, True),
    ("d_p1", 2, 2, 3, 1, 0.37, False, False),
    ("e_c2shape", 6, 9, 5, 2, 0.9, False, True),
    ("f_fullgrad", 3, 4, 4, 4, 0.55, False, False),
    # round 2: more than one output tile of every HIP assembly kernel (48 x 48 / 48 x 96 output entries = 8 x 8 / 8 x 16
    # points at q = 6, 16 x 16 / 16 x 32 points at q = 3), ragged last tiles, the C4 (d=20, p=5) and C2 (d=5, p=2) per-tile
    # geometry, symmetric (K_ZZ) cases with their diag=True vector, and the C3 (p = d = 10) and C5 (d = 50) geometries
    ("g_c4tiles", 110, 130, 20, 5, 0.6931471805599453, False, True),
    ("h_c2tiles", 120, 150, 5, 2, 0.8, False, True),
    ("i_c4sym", 100, 100, 20, 5, 0.9, True, False),
    ("j_c2sym", 128, 128, 5, 2, 0.45, True, False),
    ("k_c3tiles", 40, 52, 10, 10, 0.7, False, False),
    ("l_c5tiles", 60, 70, 50, 5, 2.0, False, True),
]
FULL_LIMIT = 400 * 400          # larger kernels are stored as a seeded sub-matrix + row / column sums of the whole matrix


def _reference_fp32_roundoff(Kern, x1, x2, v1, v2, ell, K64):
    """The reference's own fp32 round-off on this case: the kernel file run in fp32, with the value block computed the way
    gpytorch 1.4.0 ``covar_dist`` does (centred quadratic expansion through one matmul -- restated here, gpytorch is not
    installable), against the fp64 run.  Stored next to the vectors so that the HIP tolerance (2e-5) has a yardstick."""
    k = Kern()
    k._ell = torch.tensor([[ell]], dtype=torch.float32)

    def covar_dist(x1_, x2_, square_dist=False, dist_postprocess_func=None, **params):
        adj = x1_.mean(-2, keepdim=True)
        a, b = x1_ - adj, x2_ - adj
        an, bn = a.pow(2).sum(-1, keepdim=True), b.pow(2).sum(-1, keepdim=True)
        res = torch.cat([-2.0 * a, an, torch.ones_like(an)], -1) @ torch.cat([b, torch.ones_like(bn), bn], -1).t()
        return dist_postprocess_func(res.clamp_min_(0))

    k.covar_dist = covar_dist
    with torch.no_grad():
        K32 = k.forward(x1.float(), x2.float(), v1=v1.float(), v2=v2.float())
    return float((K32.double() - K64).abs().max() / K64.abs().max())


def main():
    os.makedirs(OUT, exist_ok=True)
    Kern = load_reference_kernel()
    g = torch.Generator().manual_seed(20240607)
    for name, n1, n2, d, p, ell, same, onehot in CASES:
        k = Kern()
        k._ell = torch.tensor([[ell]], dtype=torch.float64)
        x1 = torch.rand(n1, d, dtype=torch.float64, generator=g)
        v1 = torch.randn(n1 * p, d, dtype=torch.float64, generator=g)     # non-unit: exercises normalisation
        if same:
            x2, v2 = x1.clone(), v1.clone()
        else:
            x2 = torch.rand(n2, d, dtype=torch.float64, generator=g)
            if onehot:
                idx = torch.randperm(d, generator=g)[:p].sort().values
                v2 = torch.eye(d, dtype=torch.float64)[idx].repeat(n2, 1)
            else:
                v2 = torch.randn(n2 * p, d, dtype=torch.float64, generator=g)
        with torch.no_grad():
            K = k.forward(x1, x2, v1=v1, v2=v2)
            out = dict(x1=x1.numpy(), x2=x2.numpy(), v1=v1.numpy(), v2=v2.numpy(),
                       lengthscale=np.float64(ell), p=np.int64(p),
                       ref_fp32_relerr=np.float64(_reference_fp32_roundoff(Kern, x1, x2, v1, v2, ell, K)))
            if K.numel() <= FULL_LIMIT:
                out["K"] = K.numpy()
            else:
                # rows / columns around every 48- and 96-entry tile edge, the ragged tail, and a seeded random rest
                def pick(n):
                    edges = [e + o for e in range(48, n, 48) for o in (-1, 0)] + [0, n - 1]
                    rest = torch.randperm(n, generator=g)[:160].tolist()
                    return torch.tensor(sorted(set(edges + rest)))
                rows, cols = pick(K.shape[0]), pick(K.shape[1])
                out.update(K_rows=rows.numpy(), K_cols=cols.numpy(), K_sub=K[rows][:, cols].numpy(),
                           K_rowsum=K.sum(1).numpy(), K_colsum=K.sum(0).numpy(), K_shape=np.array(K.shape))
            if same:
                out["Kdiag"] = k.forward(x1, x2, diag=True, v1=v1, v2=v2).numpy()
        np.savez_compressed(os.path.join(OUT, "kernel_%s.npz" % name), **out)
        print("wrote kernel_%s.npz" % name, tuple(K.shape), "reference fp32 round-off %.2e" % out["ref_fp32_relerr"])


if __name__ == "__main__":
    main()
