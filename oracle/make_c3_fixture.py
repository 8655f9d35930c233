"""Golden vector of ONE full-size BASELINE-config-3 step: full-gradient SVGP (reference GradVariationalStrategy.py:87-137,
grad_svgp.py:41-178), d=10, N=50k, M=300 -> M' = M(d+1) = 3300, B=512 -> B' = 5632, computed by the CPU oracle
(reference-mixed precision) in the build container; the GPU box only reads the committed tests/golden/c3_step.npz.
Full-gradient SVGP == the directional kernel with p = d and canonical directions at every inducing and data point
(RBFKernelDirectionalGrad.py:157-161), num_data = n_samples (grad_svgp.py:119), all d+1 target columns (:143),
Z ~ U[0,1]^{M x d} (:61).  Both objectives are stored: ELBO and PLL (the reference's tests/test_grad_svgp.py uses PLL).
Usage: python oracle/make_c3_fixture.py   (~1 min on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dsvgp_oracle as O

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "c3_step.npz")


def make_inputs():
    g = torch.Generator().manual_seed(3303)
    N, d, M, B = 50_000, 10, 300, 512
    X = torch.rand(2 * B, d, generator=g)                  # the rows of the dataset this step touches
    Y = O.testfun(X)
    Z = torch.rand(M, d, generator=g)                      # grad_svgp.py:61
    V = torch.eye(d).repeat(M, 1)                          # RBFKernelGrad: all d partials, fixed
    P = O.init_params(Z, V, torch.float32, mean_init_std=0.1, generator=g)
    Mp = M * (d + 1)
    P["chol_variational_covar"] = torch.eye(Mp) + 0.02 * torch.randn(Mp, Mp, generator=g).tril()
    P["constant"] = torch.tensor([-0.03])
    P["raw_outputscale"] = torch.tensor(0.15)
    P["raw_lengthscale"] = torch.tensor([[0.1]])
    P["raw_noise"] = torch.tensor([-0.2])
    x = X[:B].contiguous()
    y = Y[:B].reshape(-1).contiguous()                     # all d+1 columns, interleaved (:143)
    D = torch.eye(d).repeat(B, 1)
    return P, x, y, D, N                                   # num_data = n_samples (:119)


def pack(prefix, loss, grads, mu, varn, out):
    out[prefix + "loss"] = np.float64(loss.item())
    out[prefix + "mu_head"] = mu[:256].numpy()
    out[prefix + "varn_head"] = varn[:256].numpy()
    for k, g in grads.items():
        if k == "inducing_directions":
            continue                                       # fixed canonical directions: not a parameter of this model
        if k == "chol_variational_covar":
            out[prefix + "g_LS_norm"] = np.float64(g.double().norm().item())
            out[prefix + "g_LS_block"] = g[:96, :96].numpy()
            out[prefix + "g_LS_diag"] = torch.diagonal(g).numpy()
            out[prefix + "g_LS_lastrows"] = g[-8:, :].numpy()
        else:
            out[prefix + "g_" + k] = g.numpy()


def main():
    torch.set_num_threads(os.cpu_count())
    P, x, y, D, nd = make_inputs()
    out = {}
    for mll in ("ELBO", "PLL"):
        t0 = time.time()
        loss, grads, mu, varn = O.elbo_loss_and_grads(P, x, y, D, nd, mll)
        print("oracle C3 %s step: %.1f s, loss %.8f" % (mll, time.time() - t0, loss.item()))
        pack(mll + "_", loss, grads, mu, varn, out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
