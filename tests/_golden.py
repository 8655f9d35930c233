"""Shared reader of the reference-generated kernel vectors (tests/golden/kernel_*.npz, oracle/make_golden.py).
Small kernels are stored whole (``K``); large ones as a seeded sub-matrix around every tile edge (``K_sub`` at
``K_rows`` x ``K_cols``) plus the row and column sums of the whole matrix, which pin every entry in aggregate."""
import glob
import os

import numpy as np
import torch

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "kernel_*.npz")))


def load(path):
    return np.load(path)


def kernel_error(K, g):
    """max |K - K_ref| / max |K_ref| over the stored entries, and the same for the row / column sums (None when whole)."""
    K = K.double().cpu()
    if "K" in g:
        ref = torch.from_numpy(g["K"])
        assert K.shape == ref.shape
        return ((K - ref).abs().max() / ref.abs().max()).item(), None
    assert tuple(K.shape) == tuple(int(v) for v in g["K_shape"])
    rows, cols = torch.from_numpy(g["K_rows"]), torch.from_numpy(g["K_cols"])
    ref = torch.from_numpy(g["K_sub"])
    e_sub = ((K[rows][:, cols] - ref).abs().max() / ref.abs().max()).item()
    rs, cs = torch.from_numpy(g["K_rowsum"]), torch.from_numpy(g["K_colsum"])
    e_sum = max(((K.sum(1) - rs).abs().max() / rs.abs().max()).item(), ((K.sum(0) - cs).abs().max() / cs.abs().max()).item())
    return e_sub, e_sum


STRATEGY = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_*.npz")))
            if not any(t in os.path.basename(p) for t in ("_ciq_", "_ngd_", "_grad_"))]      # (CIQ vectors: tests/test_ciq.py)
GRADIENT = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "strategy_grad_*.npz")))
PARAM_KEYS = ("inducing_points", "inducing_directions", "variational_mean", "chol_variational_covar", "constant",
              "raw_outputscale", "raw_lengthscale", "raw_noise")


def gradient_problem(path, dtype=torch.float64):
    """(params, x, y, D, num_data, flags, reference loss, reference gradients) of one ``strategy_grad_*.npz`` vector: the ELBO
    step differentiated by autograd THROUGH the reference's strategy forward and kernel file (oracle/make_strategy_fixtures.py)"""
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k]).to(dtype)
    P = dict(inducing_points=t("Z"), inducing_directions=t("V"), variational_mean=t("variational_mean"),
             chol_variational_covar=t("chol_variational_covar"), constant=t("constant"), raw_outputscale=t("raw_outputscale"),
             raw_lengthscale=t("raw_lengthscale"), raw_noise=t("raw_noise"))
    grads = {k: torch.from_numpy(g["d_" + k]) for k in PARAM_KEYS}
    grads["chol_variational_covar"] = torch.tril(grads["chol_variational_covar"])
    flags = dict(outputs=str(g["outputs"]), shared=bool(g["shared"]), p=int(g["p"]))
    return P, t("x"), t("y"), t("D"), float(g["num_data"]), flags, float(g["loss"]), grads


def strategy_problem(path, dtype=torch.float64):
    """(params dict with gpytorch-style raw hyper-parameters, x, D, flags, reference mean, reference covariance) of one
    reference-generated strategy vector (oracle/make_strategy_fixtures.py: the reference's own strategy ``forward`` text)."""
    g = np.load(path)
    t = lambda k: torch.from_numpy(g[k]).to(dtype)
    inv_softplus = lambda v: float(np.log(np.expm1(float(v))))
    P = dict(inducing_points=t("Z"), inducing_directions=t("V"), variational_mean=t("variational_mean"),
             chol_variational_covar=t("chol_variational_covar"),
             constant=torch.tensor([float(g["constant"])], dtype=dtype),
             raw_outputscale=torch.tensor(inv_softplus(g["outputscale"]), dtype=dtype),
             raw_lengthscale=torch.tensor([[inv_softplus(g["lengthscale"])]], dtype=dtype),
             raw_noise=torch.tensor([0.0], dtype=dtype))
    flags = dict(outputs=str(g["outputs"]), shared=bool(g["shared"]), p=int(g["p"]))
    return P, t("x"), t("D"), flags, torch.from_numpy(g["mean"]), torch.from_numpy(g["covariance"])
