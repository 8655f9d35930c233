"""Reference-TEXT step vectors at BASELINE geometry: tests/golden/reftext_{c2,c3,c4shard,c4}_step.npz.

What runs here is the reference's own code, executed from where it lies under /root/reference (never copied):
``DirectionalGradVariationalStrategy.forward`` (directionalvi/DirectionalGradVariationalStrategy.py:89-208) or
``GradVariationalStrategy.forward`` (directionalvi/GradVariationalStrategy.py:87-137) and the kernel file they call
(directionalvi/RBFKernelDirectionalGrad.py:41-108), in float64, behind the dense gpytorch stand-ins of
oracle/make_strategy_fixtures.py -- at the sizes the HIP kernels actually tile:

    c2       BASELINE config 2, full size: d=5,  M=200, p=2  -> M'=600,  B=512  -> B'=1536
    c3       BASELINE config 3, full size: d=10, M=300, p=d  -> M'=3300, B=512  -> B'=5632   (GradVariationalStrategy)
    c4shard  BASELINE config 4, one rank's share at 8 GPUs: d=20, M=500, p=5 -> M'=3000, B=512 -> B'=3072
    c4       BASELINE config 4, the whole global minibatch: B=4096 -> B'=24576
    c2pll / c3pll   the same C2 / C3 inputs under the PredictiveLogLikelihood objective (mll_type="PLL")

Gradients are torch autograd THROUGH that text (the stand-ins are plain torch).  What is NOT reference text, exactly as in
``make_strategy_fixtures.elbo_gradient_cases``: the two gpytorch-resident closed forms applied to the (mean, variance) the
reference's forward returns -- GaussianLikelihood.expected_log_prob with the noise counted twice, KL(N(m, S) || N(0, I)) --
and the softplus constraints.  They stay "parity unpinned" (DESIGN.md section 2).

The full C4 minibatch cannot go through the reference's forward in one call with autograd here: the forward materialises the
dense K_XX (B'=24576: 4.8 GB per temporary, ~15 temporaries alive under autograd; 64 GB container).  The data term of the
ELBO is a sum over minibatch rows and the reference's forward couples rows only through K_XX's off-diagonal, which the ELBO
never reads, so ``c4`` is assembled from EIGHT runs of the reference's forward + autograd on the eight 512-row shards (each
normalised by the global row count B'=24576) plus one KL term: loss, predictive head (shard 0 holds the first 512 rows) and
the SUM of the shard gradients.  Every number in it went through the reference's text; the summation is the only addition.

Each file also records how far the oracle restatement (oracle/dsvgp_oracle.py, float64) is from the reference text on the same
inputs (``oracle_err_*``: <= 1e-9 expected): the oracle is thereby pinned at BASELINE size, not only at toy size.

Usage:  python oracle/make_refsize_fixtures.py [c2 c3 c4shard c4]      (needs /root/reference; c4: ~10 min on 8 cores)
The GPU box only imports the ``*_inputs`` functions below (seeded input generation, no reference access).
"""
import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dsvgp_oracle as O

OUT = os.path.join(HERE, "..", "tests", "golden")


# ------------------------------------------------------------------ seeded inputs (importable without the reference)
def c2_inputs():
    """BASELINE config 2 at full size; a mid-training-like state (perturbed directions, dense L_S, non-default hypers)."""
    g = torch.Generator().manual_seed(2202)
    N, d, M, p, B = 10000, 5, 200, 2, 512
    X = torch.rand(N, d, generator=g)
    Y = O.testfun(X)
    V = torch.eye(d)[:p].repeat(M, 1) + 0.1 * torch.randn(M * p, d, generator=g)
    P = O.init_params(X[:M].clone(), V, torch.float32, mean_init_std=0.2, generator=g)
    Mp = M * (p + 1)
    P["chol_variational_covar"] = torch.eye(Mp) + 0.05 * torch.randn(Mp, Mp, generator=g).tril()
    P["constant"] = torch.tensor([0.1])
    P["raw_outputscale"] = torch.tensor(0.2)
    P["raw_lengthscale"] = torch.tensor([[0.3]])
    P["raw_noise"] = torch.tensor([-0.5])
    cols = [0, 2, 5]
    x = X[M:M + B].contiguous()
    y = Y[M:M + B][:, cols].reshape(-1).contiguous()
    D = torch.eye(d)[[c - 1 for c in cols[1:]]].repeat(B, 1) + 0.05 * torch.randn(B * p, d, generator=g)
    return P, x, y, D, (d + 1) * N


def c3_inputs():
    from make_c3_fixture import make_inputs
    return make_inputs()


def c4_inputs():
    from make_c4_fixture import make_inputs
    return make_inputs()


def c4shard_inputs():
    """rank 0's 512 rows of the C4 global minibatch, as a step of its own (normalised by its own row count)"""
    P, x, y, D, nd = c4_inputs()
    p = D.shape[0] // x.shape[0]
    return P, x[:512].contiguous(), y[:512 * (p + 1)].contiguous(), D[:512 * p].contiguous(), nd


# ------------------------------------------------------------------ the reference text, executed
_REF = {}


def _reference():
    if not _REF:
        import make_strategy_fixtures as S
        S._install_stand_ins()
        _REF["S"] = S
        _REF["Kern"] = S._load("RBFKernelDirectionalGrad.py", "_ref_rbf_dirgrad_full").RBFKernelDirectionalGrad
        _REF["DGVS"] = S._load("DirectionalGradVariationalStrategy.py", "_ref_dgvs_full").DirectionalGradVariationalStrategy
        _REF["GVS"] = S._load("GradVariationalStrategy.py", "_ref_gvs_full").GradVariationalStrategy
    return _REF


def _leaves(P):
    return {k: v.detach().double().clone().requires_grad_(True) for k, v in P.items()}


def reference_data_term(P, x, y, D, rows_glob, full_gradient=False, mll="ELBO"):
    """-(sum_j ll_j) / rows_glob and its gradients, through the reference's strategy forward + kernel file.
    mll: "ELBO" (VariationalELBO: GaussianLikelihood.expected_log_prob) or "PLL" (PredictiveLogLikelihood: log_marginal of the
    NOISED predictive -- the reference passes likelihood(model(x)) to the mll, so the noise enters twice, directional_vi.py:245)."""
    import torch.nn.functional as F
    R = _reference()
    S = R["S"]
    L = _leaves(P)
    x, y, D = x.double(), y.double(), D.double()
    kern = R["Kern"]()
    kern._ell = F.softplus(L["raw_lengthscale"])
    model = S._Model(kern, L["constant"].reshape(()), F.softplus(L["raw_outputscale"]))
    nq = L["variational_mean"].shape[0]
    if full_gradient:                                 # GradVariationalStrategy: ONE joint model call on [Z ; x] (:89-99)
        d = x.shape[1]
        eye = torch.eye(d, dtype=torch.float64)

        def forward(xx, _model=model, _eye=eye):
            v = _eye.repeat(xx.shape[0], 1)
            return S.MultivariateNormal(_model.mean_module(xx), _model.covar_module(xx, xx, v1=v, v2=v))

        model.forward = forward
        strat = R["GVS"](model, L["inducing_points"].detach(), S._VarDist(nq), learn_inducing_locations=True)
        out = strat.forward(x, strat.inducing_points, L["variational_mean"], S.CholLazyTensor(torch.tril(L["chol_variational_covar"])))
    else:
        strat = R["DGVS"](model, L["inducing_points"].detach(), L["inducing_directions"].detach(), S._VarDist(nq),
                          learn_inducing_locations=True)
        out = strat.forward(x, strat.inducing_points, L["variational_mean"], S.CholLazyTensor(torch.tril(L["chol_variational_covar"])),
                            derivative_directions=D)
    mu, var = out.mean, torch.diagonal(out.covariance_matrix)
    noise = F.softplus(L["raw_noise"]).reshape(()) + 1e-4                    # GaussianLikelihood: GreaterThan(1e-4)
    varn = (var + noise).clamp_min(1e-6)                                     # likelihood(q(f)).variance
    if mll == "ELBO":
        ll = -0.5 * (((y - mu) ** 2 + varn) / noise + torch.log(noise) + math.log(2 * math.pi))
    else:
        tot = (varn + noise).clamp_min(1e-8)
        ll = -0.5 * ((y - mu) ** 2 / tot + torch.log(tot) + math.log(2 * math.pi))
    part = -(ll.sum() / rows_glob)
    part.backward()
    grads = {k: (L[k].grad if L[k].grad is not None else torch.zeros_like(L[k])) for k in L}
    grads["inducing_points"] = strat.inducing_points.grad
    if not full_gradient:
        grads["inducing_directions"] = strat.inducing_directions.grad
    return part.detach(), grads, mu.detach(), varn.detach()


def kl_term(P, num_data):
    L = _leaves(P)
    Lt = torch.tril(L["chol_variational_covar"])
    m = L["variational_mean"]
    nq = m.shape[0]
    kl = 0.5 * ((m * m).sum() + (Lt * Lt).sum() - nq - torch.log(torch.diagonal(Lt) ** 2).sum())
    part = kl / num_data
    part.backward()
    return part.detach(), {"variational_mean": m.grad, "chol_variational_covar": L["chol_variational_covar"].grad}


def reference_step(P, x, y, D, num_data, full_gradient=False, shard_rows=None, mll="ELBO"):
    """(loss, grads, mu_head, varn_head) of one ELBO step through the reference text, row shards summed when asked"""
    B = x.shape[0]
    q = y.shape[0] // B
    p = D.shape[0] // B
    rows_glob = y.shape[0]
    step = shard_rows or B
    loss, grads, mu_head, varn_head = 0.0, None, None, None
    # the reference text runs under a float64 default dtype, like the reference's experiment scripts (exp_script.py:56);
    # the seeded input generators above run under the float32 default (main() restores it): the GPU box regenerates the
    # same inputs
    torch.set_default_dtype(torch.float64)
    for r0 in range(0, B, step):
        r1 = min(B, r0 + step)
        t0 = time.time()
        part, g, mu, varn = reference_data_term(P, x[r0:r1], y[r0 * q:r1 * q], D[r0 * p:r1 * p], rows_glob, full_gradient, mll)
        print("    rows %d..%d through the reference forward + autograd: %.1f s" % (r0, r1, time.time() - t0), flush=True)
        loss = loss + part
        grads = g if grads is None else {k: grads[k] + g[k] for k in grads}
        if r0 == 0:
            mu_head, varn_head = mu[:256].clone(), varn[:256].clone()
    klp, gk = kl_term(P, num_data)
    loss = loss + klp
    for k in gk:
        grads[k] = grads[k] + gk[k]
    grads["chol_variational_covar"] = torch.tril(grads["chol_variational_covar"])
    return loss, grads, mu_head, varn_head


def relmax(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300)).item()


def pack(loss, grads, mu_head, varn_head, skip=()):
    out = dict(loss=np.float64(loss.item()), mu_head=mu_head.numpy(), varn_head=varn_head.numpy())
    for k, g in grads.items():
        if k in skip:
            continue
        if k == "chol_variational_covar":
            out["g_LS_norm"] = np.float64(g.norm().item())
            out["g_LS_block"] = g[:96, :96].numpy()
            out["g_LS_diag"] = torch.diagonal(g).numpy()
            out["g_LS_lastrows"] = g[-8:, :].numpy()
            out["g_LS_rowsum"] = g.sum(1).numpy()                     # every entry in aggregate
            out["g_LS_colsum"] = g.sum(0).numpy()
        else:
            out["g_" + k] = g.numpy()
    return out


# name -> (inputs, GradVariationalStrategy?, row shards, objective).  The PLL cases: the reference's own tests train the
# full-gradient model with mll_type="PLL" (tests/test_grad_svgp.py) and offer it for DSVGP (directional_vi.py:218-219).
CASES = {"c2": (c2_inputs, False, None, "ELBO"), "c3": (c3_inputs, True, None, "ELBO"), "c4shard": (c4shard_inputs, False, None, "ELBO"),
         "c4": (c4_inputs, False, 512, "ELBO"), "c2pll": (c2_inputs, False, None, "PLL"), "c3pll": (c3_inputs, True, None, "PLL")}
INPUTS = {"c2pll": "c2", "c3pll": "c3"}           # (cases that share another case's inputs)


def main(names):
    torch.set_num_threads(os.cpu_count())
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        inputs, full_gradient, shard_rows, mll = CASES[name]
        torch.set_default_dtype(torch.float32)
        P, x, y, D, nd = inputs()
        assert x.dtype == torch.float32 and P["inducing_points"].dtype == torch.float32
        print("%s: M'=%d, B'=%d" % (name, P["variational_mean"].shape[0], y.shape[0]), flush=True)
        t0 = time.time()
        loss, grads, mu_head, varn_head = reference_step(P, x, y, D, nd, full_gradient, shard_rows, mll)
        print("  reference text: %.1f s, loss %.10f" % (time.time() - t0, loss.item()), flush=True)
        out = pack(loss, grads, mu_head, varn_head, skip=("inducing_directions",) if full_gradient else ())
        # the oracle restatement (float64) on the same inputs: pinned at this size by the numbers above
        t0 = time.time()
        P64 = {k: v.double() for k, v in P.items()}
        lo, go, muo, varo = O.elbo_loss_and_grads(P64, x.double(), y.double(), D.double(), nd, mll)
        errs = {"loss": abs(lo.item() - loss.item()) / abs(loss.item()), "mu": relmax(muo[:256], mu_head), "varn": relmax(varo[:256], varn_head)}
        for k in O.PARAM_NAMES:
            if full_gradient and k == "inducing_directions":
                continue
            errs[k] = relmax(go[k], grads[k])
        print("  oracle (fp64) vs reference text, %.1f s: %s" % (time.time() - t0, ", ".join("%s %.1e" % kv for kv in errs.items())), flush=True)
        torch.set_default_dtype(torch.float32)
        for k, v in errs.items():
            out["oracle_err_" + k] = np.float64(v)
        path = os.path.join(OUT, "reftext_%s_step.npz" % name)
        np.savez_compressed(path, **out)
        print("  wrote %s (%d KiB)" % (path, os.path.getsize(path) // 1024), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
