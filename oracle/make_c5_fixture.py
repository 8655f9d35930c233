"""Golden vectors of ONE full-size BASELINE-config-5 step: CIQ-whitened DSVGP with natural parameters (reference
CiqDirectionalGradVariationalStrategy.py:19-123,197-295; ``train_gp(use_ciq=True)``), d=50, M=1024, p=5 -> M'=6144,
B=512 -> B'=3072, Q=15 quadrature points, computed by the CPU oracle in fp32 (the reference's default dtype) in the build
container; the GPU box only reads the committed tests/golden/c5_step.npz.  Two states:
  init  exactly what ``train_gp(use_ciq=True)`` constructs (directional_vi.py:58-60,164-165): lengthscale = 1/num_inducing,
        natural_vec = 1e-3 randn, natural_mat = -I/2, raw hypers 0.  With ell = 1/1024 in d = 50 the kernel between
        distinct points underflows to 0, so the minibatch includes 8 of the inducing rows (as a shuffled DataLoader batch
        does now and then) to keep K_ZX from vanishing identically;
  mid   a mid-training-like state (ell ~ 2, generic SPD precision, perturbed directions) where K_ZZ is far from diagonal and
        msMINRES runs several convergence checks.
Usage: python oracle/make_c5_fixture.py [init|mid]   (several minutes per state on 8 cores, ~25 GB of memory)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dsvgp_oracle as O

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "c5_step_%s.npz")
N, d, M, p, B, Q = 100_000, 50, 1024, 5, 512, 15


def make_inputs(state):
    g = torch.Generator().manual_seed(5005 if state == "init" else 5006)
    X = torch.rand(M + B, d, generator=g)
    Y = O.testfun(X)
    Z = X[:M].clone()                                      # inducing_data_initialization=True (:138-145)
    V = torch.eye(d)[:p].repeat(M, 1)
    Mp = M * (p + 1)
    if state == "init":
        P = O.init_natural_params(Z, V, torch.float32, mean_init_std=1e-3, generator=g)
        ell = torch.tensor(1.0 / M)                        # "stable initialization of lengthscale for CIQ" (:58-60)
        P["raw_lengthscale"] = torch.log(torch.expm1(ell)).reshape(1, 1)
        lo = M - 8
    else:
        V = V + 0.05 * torch.randn(M * p, d, generator=g)
        P = O.init_natural_params(Z, V, torch.float32, mean_init_std=0.2, generator=g)
        R = 0.01 * torch.randn(Mp, 64, generator=g)
        P["natural_mat"] = -0.5 * (torch.eye(Mp) + R @ R.t())           # generic SPD precision (low-rank perturbation)
        P["constant"] = torch.tensor([0.05])
        P["raw_outputscale"] = torch.tensor(0.2)
        P["raw_lengthscale"] = torch.tensor([[1.9]])                    # ell ~ 2.04
        P["raw_noise"] = torch.tensor([-0.4])
        lo = M
    cols = sorted([0] + (torch.randperm(d, generator=g)[:p] + 1).tolist())
    x = X[lo:lo + B].contiguous()
    y = Y[lo:lo + B][:, cols].reshape(-1).contiguous()
    D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1)
    return P, x, y, D, (d + 1) * N


def main():
    torch.set_num_threads(os.cpu_count())
    states = sys.argv[1:] or ["init", "mid"]
    for state in states:
        # "init64": the init state evaluated by the oracle in FLOAT64 (the arithmetic of the reference's experiments under
        # torch.set_default_dtype(torch.float64), experiments/bunny/exp_bunny.py:66,78): the vector that the float64 MODEL's CIQ
        # step -- which runs on the fp32 CIQ kernels, DESIGN.md section 9 -- is measured against at C5 size
        f64 = state.endswith("64")
        P, x, y, D, nd = make_inputs(state[:-2] if f64 else state)
        if f64:
            P, x, y, D = {k: v.double() for k, v in P.items()}, x.double(), y.double(), D.double()
        st = {}
        t0 = time.time()
        loss, grads, mu, varn = O.ciq_loss_and_grads(P, x, y, D, nd, Q=Q, stats=st)
        print("oracle C5 %s step: %.1f s, loss %.8f, lmin %.4g lmax %.4g, %d msMINRES iterations"
              % (state, time.time() - t0, loss.item(), st["lmin"], st["lmax"], st["iterations"]), flush=True)
        out = dict(loss=np.float64(loss.item()), mu=mu.numpy(), varn=varn.numpy(), lmin=np.float64(st["lmin"]),
                   lmax=np.float64(st["lmax"]), iterations=np.int64(st["iterations"]))
        for k, gk in grads.items():
            if k == "natural_mat":
                out["g_nm_norm"] = np.float64(gk.double().norm().item())
                out["g_nm_block"] = gk[:96, :96].numpy()
                out["g_nm_diag"] = torch.diagonal(gk).numpy()
                out["g_nm_lastrows"] = gk[-8:, :].numpy()
            else:
                out["g_" + k] = gk.numpy()
        np.savez_compressed(OUT % state, **out)
        print("wrote", OUT % state, os.path.getsize(OUT % state) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()
