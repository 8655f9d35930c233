"""Golden vector of ONE full-size BASELINE-config-4 step (d=20, M=500, p=5 -> M'=3000, B=4096 -> B'=24576)
computed by the CPU oracle (reference-mixed precision) in the build container; the GPU box only reads the
committed tests/golden/c4_step.npz.  Inputs are regenerated from the same seeded torch CPU generator.
Usage: python oracle/make_c4_fixture.py   (~1 min on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dsvgp_oracle as O

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "c4_step.npz")


def make_inputs():
    g = torch.Generator().manual_seed(4242)
    N, d, M, p, B = 20000, 20, 500, 5, 4096
    X = torch.rand(N, d, generator=g)
    Y = O.testfun(X)
    V = torch.eye(d)[:p].repeat(M, 1) + 0.05 * torch.randn(M * p, d, generator=g)
    P = O.init_params(X[:M].clone(), V, torch.float32, mean_init_std=0.1, generator=g)
    Mp = M * (p + 1)
    P["chol_variational_covar"] = torch.eye(Mp) + 0.02 * torch.randn(Mp, Mp, generator=g).tril()
    P["constant"] = torch.tensor([0.05])
    P["raw_outputscale"] = torch.tensor(0.1)
    P["raw_lengthscale"] = torch.tensor([[0.2]])
    P["raw_noise"] = torch.tensor([-0.3])
    cols = [0, 3, 7, 8, 15, 19]
    x = X[M:M + B].contiguous()
    y = Y[M:M + B][:, cols].reshape(-1).contiguous()
    D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1)
    return P, x, y, D, (d + 1) * 1_000_000


def main():
    torch.set_num_threads(os.cpu_count())
    P, x, y, D, nd = make_inputs()
    t0 = time.time()
    loss, grads, mu, varn = O.elbo_loss_and_grads(P, x, y, D, nd)
    print("oracle C4 step: %.1f s, loss %.8f" % (time.time() - t0, loss.item()))
    out = dict(loss=np.float64(loss.item()), mu_head=mu[:256].numpy(), varn_head=varn[:256].numpy())
    for k, g in grads.items():
        if k == "chol_variational_covar":
            out["g_LS_norm"] = np.float64(g.double().norm().item())
            out["g_LS_block"] = g[:96, :96].numpy()
            out["g_LS_diag"] = torch.diagonal(g).numpy()
            out["g_LS_lastrows"] = g[-8:, :].numpy()
        else:
            out["g_" + k] = g.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
