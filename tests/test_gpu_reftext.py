"""The HIP step at BASELINE geometry against the REFERENCE'S OWN TEXT.

tests/golden/reftext_{c2,c3,c4shard,c4}_step.npz hold loss, predictive head and every gradient of one ELBO step computed by
executing the reference's strategy forward (directionalvi/DirectionalGradVariationalStrategy.py:89-208 /
GradVariationalStrategy.py:87-137) and kernel file (RBFKernelDirectionalGrad.py:41-108) in float64 and differentiating
through them with torch autograd (oracle/make_refsize_fixtures.py; the full C4 minibatch as the sum of eight 512-row runs
of that text).  Sizes: C2 M'=600 x B'=1536, C3 M'=3300 x B'=5632, C4 per-rank shard M'=3000 x B'=3072, C4 M'=3000 x B'=24576:
every GEMM, solve, Cholesky launch and assembly tile of the HIP path is exercised across its tile boundaries against numbers
the builder did not write.  (Still restated, not reference text: the Gaussian expected log-likelihood, the KL closed form and
the softplus constraints -- gpytorch internals -- applied to the (mean, variance) that the reference's forward returns.)

Inputs are regenerated on the GPU box from the seeds (oracle/make_refsize_fixtures.py:*_inputs; no reference access).

Stated tolerances.  The vectors are float64 truth, the fp32 engine is the reference's default precision (fp32 model, fp64
Cholesky / solves): its distance to fp64 truth is that of the reference's own fp32 run (kernel entries 1e-7, amplified by
cond(L) ~ 1e2-1e3 on the K_ZZ path).  fp32 engine: loss 2e-6, predictive head 1e-4, gradients 1e-3 of the max entry;
fp64 engine: loss 1e-12, head 1e-10, gradients 1e-8.  Measured on MI355X (round 3; printed as ``[parity] reftext ...`` lines):
fp32 loss <= 3.7e-7, head <= 4.6e-5 (C2), gradients <= 2.9e-4 (last rows of L_S-bar at C2; <= 2.4e-6 at C4);
fp64 loss <= 2e-14, head <= 3.9e-11, gradients <= 9.2e-10 (C2; <= 7.3e-14 at C4)."""
import os
import sys

import numpy as np
import pytest
import torch

import dsvgp_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def _load(name):
    import make_refsize_fixtures as R
    g = np.load(os.path.join(GOLD, "reftext_%s_step.npz" % name))
    P, x, y, D, nd = getattr(R, R.INPUTS.get(name, name) + "_inputs")()
    return g, P, x, y, D, nd


def _errors(g, loss, grads, mu, varn, skip=()):
    t = lambda k: torch.from_numpy(g[k])
    errs = {"loss": abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), "mu": relmax(mu[:256], t("mu_head"))}
    if varn is not None and varn.numel():
        errs["varn"] = relmax(varn[:256], t("varn_head"))
    for k in O.PARAM_NAMES:
        if k in skip:
            continue
        if k == "chol_variational_covar":
            gl = torch.tril(grads[k]).double().cpu()
            errs["LS_norm"] = abs(gl.norm().item() - float(g["g_LS_norm"])) / float(g["g_LS_norm"])
            errs["LS_block"] = relmax(gl[:96, :96], t("g_LS_block"))
            errs["LS_diag"] = relmax(torch.diagonal(gl), t("g_LS_diag"))
            errs["LS_lastrows"] = relmax(gl[-8:, :], t("g_LS_lastrows"))
            errs["LS_rowsum"] = relmax(gl.sum(1), t("g_LS_rowsum"))          # every entry of the gradient, in aggregate
            errs["LS_colsum"] = relmax(gl.sum(0), t("g_LS_colsum"))
        else:
            errs[k] = relmax(grads[k], t("g_" + k))
    return errs


# per-case tolerances (loss, predictive head, gradients): a few times the errors measured on MI355X (round 4 GPUTEST log: fp32 C2
# 5.4e-8 / 4.6e-5 / 2.9e-4 -- the last rows of L_S-bar --, C3 5.6e-8 / 1.3e-5 / 1.5e-4, C4 and its shard 1.7e-7 / 9.3e-7 / 2.4e-6;
# fp64 C2 2.0e-14 / 3.9e-11 / 9.2e-10, C3 7.9e-16 / 3.7e-13 / 2.4e-11, C4 and its shard 3.2e-16 / 9.8e-15 / 7.2e-14)
TOL32 = {"c2": (1e-6, 1.5e-4, 1e-3), "c3": (1e-6, 5e-5, 5e-4), "c4shard": (1e-6, 1e-5, 2e-5), "c4": (1e-6, 1e-5, 2e-5)}
TOL64 = {"c2": (5e-12, 2e-10, 5e-9), "c3": (1e-13, 5e-12, 2e-10), "c4shard": (1e-13, 1e-12, 1e-12), "c4": (1e-13, 1e-12, 1e-12)}


def _check(tag, errs, tol_loss, tol_head, tol_grad):
    print("[parity] reftext %s: %s" % (tag, ", ".join("%s %.2e" % kv for kv in errs.items())))
    for k, v in errs.items():
        tol = tol_loss if k == "loss" else tol_head if k in ("mu", "varn") else tol_grad
        assert v < tol, (tag, k, v, tol)


@pytest.mark.parametrize("fast", [True, False], ids=["gram", "per-output"])
@pytest.mark.parametrize("name", ["c2", "c3", "c4shard", "c4"])
def test_fp32_step_against_reference_text_at_baseline_size(dsvgp, gpu_device, name, fast):
    g, P, x, y, D, nd = _load(name)
    eng = dsvgp.ElboEngine(gpu_device)
    if name == "c3":
        eng.chol_jitter = 1e-8                     # GradVariationalStrategy: psd_safe_cholesky's default jitter (:72)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO", fast=fast)
    torch.cuda.synchronize()
    assert grads["chol_variational_covar"].triu(1).abs().max().item() == 0.0
    errs = _errors(g, loss, grads, mu, varn, skip=("inducing_directions",) if name == "c3" else ())
    _check("%s fp32 %s" % (name, "gram" if fast else "per-output"), errs, *TOL32[name])


@pytest.mark.parametrize("fast", [True, False], ids=["gram", "per-output"])
@pytest.mark.parametrize("name", ["c2", "c3", "c4shard", "c4"])
def test_fp64_step_against_reference_text_at_baseline_size(dsvgp, gpu_device, name, fast):
    from dsvgp_amd._step64 import ElboEngine64
    g, P, x, y, D, nd = _load(name)
    eng = ElboEngine64(gpu_device)
    eng.fast_min_work = 0
    if name == "c3":
        eng.chol_jitter = 1e-8
    Pg = {k: v.double().to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.double().to(gpu_device), y.double().to(gpu_device), D.double().to(gpu_device),
                                               nd, "ELBO", fast=fast)
    torch.cuda.synchronize()
    errs = _errors(g, loss, grads, mu, varn, skip=("inducing_directions",) if name == "c3" else ())
    _check("%s fp64 %s" % (name, "gram" if fast else "per-output"), errs, *TOL64[name])


@pytest.mark.parametrize("name", ["c2pll", "c3pll"])
def test_pll_step_against_reference_text_at_baseline_size(dsvgp, gpu_device, name):
    """mll_type="PLL" (PredictiveLogLikelihood; what the reference's tests/test_grad_svgp.py trains with, and an option of
    directional_vi.train_gp, :218-219) at C2 / C3 size: the per-output path of the fp32 engine and the fp64 engine against the
    reference's strategy forward + kernel file + autograd (the log_marginal closed form of the noised predictive is restated)."""
    from dsvgp_amd._step64 import ElboEngine64
    g, P, x, y, D, nd = _load(name)
    skip = ("inducing_directions",) if name.startswith("c3") else ()
    eng = dsvgp.ElboEngine(gpu_device)
    if name.startswith("c3"):
        eng.chol_jitter = 1e-8
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "PLL")
    torch.cuda.synchronize()
    assert eng.c_step_used            # (round 5: the per-output step is ONE C call, dsvgp_elbo_step_po_f32)
    _check("%s fp32" % name, _errors(g, loss, grads, mu, varn, skip=skip), *TOL32[name[:2]])
    eng64 = ElboEngine64(gpu_device)
    if name.startswith("c3"):
        eng64.chol_jitter = 1e-8
    Pd = {k: v.double().to(gpu_device) for k, v in P.items()}
    loss, grads, mu, varn = eng64.loss_and_grads(Pd, x.double().to(gpu_device), y.double().to(gpu_device), D.double().to(gpu_device),
                                                 nd, "PLL")
    torch.cuda.synchronize()
    _check("%s fp64" % name, _errors(g, loss, grads, mu, varn, skip=skip), *TOL64[name[:2]])


@pytest.mark.parametrize("name", ["c2", "c3", "c4shard", "c4"])
def test_phi_argument_on_the_fp32_kernel_against_reference_text(dsvgp, gpu_device, name):
    """DEFAULT since round 4: tril(L^T L-bar) = -tril([S - I | m'][G ; b^T]) -- both operands fp32 data out of fp32 MFMA products -- runs
    on the fp32 LDS-DMA kernel and its result is widened for the fp64 Cholesky backward; ``ElboEngine.phi_arg_fp64`` (flag 64 of
    dsvgp_elbo_step_f32) keeps the fp64-accumulated product.  Both forms are held to the reference-text vectors at the same
    tolerances, and the default may not be further out than the fp64-accumulated form by more than two fp32-accurate evaluations of
    one step differ by anyway (their split-K sums meet in atomics: run to run a scalar gradient moves between 5e-8 and 4e-6).
    Measured: C4 inducing_points 1.8e-6 / 1.6e-6, inducing_directions 1.4e-6 / 1.5e-6, C3 inducing_points 2.5e-5 / 2.6e-5."""
    g, P, x, y, D, nd = _load(name)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    errs = {}
    for f64 in (False, True):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.phi_arg_fp64 = f64
        if name == "c3":
            eng.chol_jitter = 1e-8
        loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO", fast=True)
        torch.cuda.synchronize()
        assert eng.c_step_used
        errs[f64] = _errors(g, loss, grads, mu, varn, skip=("inducing_directions",) if name == "c3" else ())
        del eng
    _check("%s tril(L^T L-bar) on the fp32 kernel (default)" % name, errs[False], *TOL32[name])
    _check("%s tril(L^T L-bar) with fp64 accumulation" % name, errs[True], *TOL32[name])
    worst = lambda e: max(v for k, v in e.items() if k not in ("loss", "mu", "varn"))
    assert worst(errs[False]) <= 1.5 * worst(errs[True]) + 2e-7, (name, worst(errs[False]), worst(errs[True]))
    for k in errs[False]:
        assert errs[False][k] <= max(3.0 * errs[True][k], 3e-5), (name, k, errs[False][k], errs[True][k])


@pytest.mark.parametrize("name", ["c2", "c3", "c4shard", "c4"])
def test_split_bf16_step_against_reference_text_at_baseline_size(dsvgp, gpu_device, name):
    """OPT-IN mode (``ElboEngine.split_bf16`` / flag 32 of dsvgp_elbo_step_f32): the Gram product and the dense K_ZX-bar product as
    bf16 x 3 split products on the bf16 matrix pipe (six bf16 MFMA products per fp32 product, fp32 accumulation; csrc/gemm3b.hip).
    Held to the reference-text vectors at the fp32 path's tolerances, the default (fp32 MFMA) engine's errors printed beside it.
    Measured (round 4): the worst error per configuration is the same (C2 2.9e-4 / 2.9e-4, C3 1.0e-4 / 1.0e-4, C4 2.4e-6 / 2.7e-6),
    individual entries land on either side of the default's by the factors two fp32-accurate evaluations differ by; the widest
    gap is on the scalar hyper-parameter gradients (C3 raw_outputscale 2.6e-6 against 9.6e-8, C2 1.0e-5 ... 1.2e-5 against 2.5e-6 ... 3.2e-6).  So the
    mode does NOT meet "every error <= the fp32 path's"; asserted here: the fp32 tolerances, the worst error within 1.5x of the
    default's worst, and no entry beyond 3x the default's unless it is below 3e-5."""
    g, P, x, y, D, nd = _load(name)
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    errs = {}
    for split in (False, True):
        eng = dsvgp.ElboEngine(gpu_device)
        eng.split_bf16 = split
        if name == "c3":
            eng.chol_jitter = 1e-8
        loss, grads, mu, varn = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO", fast=True)
        torch.cuda.synchronize()
        assert eng.c_step_used
        errs[split] = _errors(g, loss, grads, mu, varn, skip=("inducing_directions",) if name == "c3" else ())
        del eng
    _check("%s fp32 MFMA (default)" % name, errs[False], *TOL32[name])
    _check("%s bf16 x 3 split products" % name, errs[True], *TOL32[name])
    worst = lambda e: max(v for k, v in e.items() if k not in ("loss", "mu", "varn"))
    assert worst(errs[True]) <= 1.5 * worst(errs[False]) + 2e-7, (name, worst(errs[True]), worst(errs[False]))
    for k in errs[True]:
        assert errs[True][k] <= max(3.0 * errs[False][k], 3e-5), (name, k, errs[True][k], errs[False][k])
