"""Reference-faithful CPU training step for the ``cpu_baseline`` leg of bench.py.  TEST INFRASTRUCTURE.

Same op sequence as the reference loop (directionalvi/directional_vi.py:229-254) on top of the oracle
restatement: ``DataLoader(TensorDataset, shuffle=True)`` batching (per-index __getitem__ + collate),
``select_cols_of_y`` with Python's ``random.sample``, four kernel assemblies (K_ZX, K_XZ, K_ZZ, diag K_XX) EVALUATED
WITH THE REFERENCE'S OWN OP STRUCTURE (``dsvgp_oracle.kernel_matrix_refseq``: matmul / bmm projections, column-permutation
gathers, ``repeat`` broadcasts, perfect-shuffle gather -- RBFKernelDirectionalGrad.py:57-107; not the pair-wise einsum
form the parity tests use), fp64 Cholesky + two fp64 triangular solves, dense (S - I) products, torch autograd backward
and two ``torch.optim.Adam`` steps with per-iteration LambdaLR schedulers.
"""
import random
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

import dsvgp_oracle as O


class RefTrainer:
    def __init__(self, n, d, M, p, B, lr=0.01, num_data_override=None, seed=0, assembly="reference-sequence",
                 full_gradient=False, dtype=torch.float32):
        # dtype: torch.float64 = the reference under torch.set_default_dtype(torch.float64) (experiments/synthetic/exp_script.py:56)
        g = torch.Generator().manual_seed(seed)
        X = torch.rand(n, d, generator=g).to(dtype)
        Y = O.testfun(X)
        self.dtype = dtype
        self.d, self.p, self.full_gradient = d, p, full_gradient
        self.loader = DataLoader(TensorDataset(X, Y), batch_size=B, shuffle=True)
        self.it = iter(self.loader)
        self.num_data = num_data_override or ((d + 1) * n if not full_gradient else n)
        Z0 = torch.rand(M, d, generator=g).to(dtype) if full_gradient else X[:M].clone()
        P = O.init_params(Z0, torch.eye(d, dtype=dtype)[:p].repeat(M, 1), dtype, 1e-3, g)
        self.P = {k: v.requires_grad_(True) for k, v in P.items()}
        if full_gradient:
            self.P["inducing_directions"].requires_grad_(False)       # RBFKernelGrad: fixed canonical directions
        var = [self.P["variational_mean"], self.P["chol_variational_covar"]]
        hyp = [v for k, v in self.P.items() if k not in ("variational_mean", "chol_variational_covar") and v.requires_grad]
        self.opt_v = torch.optim.Adam([{"params": var}], lr=lr)
        self.opt_h = torch.optim.Adam([{"params": hyp}], lr=lr)
        self.sch_v = torch.optim.lr_scheduler.LambdaLR(self.opt_v, lr_lambda=lambda e: 1.0)
        self.sch_h = torch.optim.lr_scheduler.LambdaLR(self.opt_h, lr_lambda=lambda e: 1.0)
        self.assembly_seconds = 0.0           # forward time spent in the four kernel assemblies (accumulated)
        if assembly == "reference-sequence":
            def timed(*a):
                t0 = time.perf_counter()
                out = O.kernel_matrix_refseq(*a)
                self.assembly_seconds += time.perf_counter() - t0
                return out
            self.assembly = timed
        elif assembly == "pairwise":
            self.assembly = None
        else:
            raise ValueError(assembly)

    def step(self):
        try:
            xb, yb = next(self.it)
        except StopIteration:
            self.it = iter(self.loader)
            xb, yb = next(self.it)
        if self.full_gradient:                                                        # grad_svgp.py:143
            idx = list(range(self.d + 1))
        else:
            idx = sorted(random.sample(range(1, self.d + 1), self.p) + [0])           # select_cols_of_y, :68-90
        yb = yb[:, idx]
        D = torch.eye(self.d, dtype=self.dtype)[np.array(idx[1:]) - 1].repeat(yb.size(0), 1)   # :238
        yb = yb.reshape(torch.numel(yb))                                              # :241
        self.opt_v.zero_grad()
        self.opt_h.zero_grad()
        loss, _, _ = O.elbo_forward(self.P, xb, yb, D, self.num_data, assembly=self.assembly)
        loss.backward()
        self.opt_v.step(); self.sch_v.step()
        self.opt_h.step(); self.sch_h.step()
        return float(loss.detach())
