"""fp64 model mode of the DSVGP step.

The reference's experiment drivers switch the whole model to double precision (``torch.set_default_dtype(torch.float64)``,
experiments/synthetic/exp_script.py:56 and the other exp scripts): parameters, data, kernel matrices, the whitening solve
and the ELBO are then all fp64.  ``ElboEngine64`` is that mode of ``directional_vi.train_gp`` / ``DirectionalGradVariationalStrategy
.forward`` (DGVS.py:89-208) on the MI355X:

  * every O(M'^2 d), O(M' B' d), O(M'^3) and O(M'^2 B') term runs on the library's own HIP kernels -- the fp64 kernel assembly
    (csrc/assemble64.hip: T = P1 P2^T on the fp64 MFMA GEMM + per-pair transforms), the blocked MFMA Cholesky with fused inverse
    (csrc/potrf.hip), the triangular products (csrc/gemm64.hip / gemm.hip) and the column statistics (assemble64.hip);
  * the O(B') likelihood terms and the scalar tail are closed-form launches too (dsvgp_likelihood_terms_f64 for the general path,
    dsvgp_elbo_fast_tail_f64 for the Gram formulation); the O(M'^2) KL term and the softplus slopes are elementwise torch fp64
    expressions in closed form.  Nothing goes through torch.autograd (the reference differentiates the same expressions by
    autograd, directional_vi.py:245-249).

Two formulations, as in the fp32 engine: the general (variance-carrying) one covers ELBO and PLL; in ELBO mode, when the caller
does not read the per-output variances, the Gram-matrix formulation (``_elbo_fast64``) needs three [M', B'] products instead
of six.  q(u) may be a NaturalVariationalDistribution (``train_gp(use_ngd=True)``: natural parameters in, expectation-parameter gradients out).
It is not the benchmark path (the headline configs run the reference's default fp32 model; ``bench.py --fp64`` times it).  Under ``parallel.DataParallel`` the gradients of the row shards are
summed by one all-reduce at the end of the step (no early operand).  No CPU fallback: the inputs must be HIP tensors.
Shared inducing directions run in fp64 (``_shared_step64``), and so does the CIQ strategy of a float64 model (``_ciq_step64``:
fp64 msMINRES, the ``*_f64`` entry points of csrc/ciq.hip).
"""
import torch
import torch.nn.functional as F

from . import _lib, _ops
from ._step import CHOL_TRIES, NGD_PARAM_NAMES, PARAM_NAMES, _NGD_RENAME, ElboEngine, NotPSDError

KXX_JITTER = 1e-4       # data_data_covar.add_jitter(1e-4), DGVS.py:202
MIN_VARIANCE = 1e-6     # MultivariateNormal.variance clamp (gpytorch settings.min_variance)
NOISE_FLOOR = 1e-4      # GaussianLikelihood noise constraint GreaterThan(1e-4)

f64 = torch.float64
TRANS_A, TRANS_B = _lib.TRANS_A, _lib.TRANS_B
A_LOWER, A_UPPER, B_LOWER, OUT_LOWER = _lib.A_LOWER, _lib.A_UPPER, _lib.B_LOWER, _lib.OUT_LOWER


class ElboEngine64(ElboEngine):
    """Double-precision DSVGP step on one GPU (see module docstring).  Same call surface as ``ElboEngine``."""

    dtype = f64

    def __init__(self, device, trsm_nb=None):
        super().__init__(device, trsm_nb)
        self.elbo_fast = True
        self.fast_min_work = 4_000_000      # M' B' below which the per-output path is used anyway (C2: host-bound, 2.42 vs 2.58 ms)

    # ---- forward pieces -----------------------------------------------------------------------
    def _check(self, params, x, D):
        for k in PARAM_NAMES:
            if k not in params:
                raise NotImplementedError("fp64 model mode covers the Cholesky-whitened DSVGP parameterisation (%s missing)" % k)
            if params[k].dtype != f64:
                raise TypeError("fp64 model mode: parameter %s is %s" % (k, params[k].dtype))
            if not params[k].is_cuda:
                raise _lib.DsvgpError("%s must live on the GPU: the DSVGP hot path has no CPU fallback" % k)
        if x.dtype != f64 or (D is not None and D.numel() and D.dtype != f64):
            raise TypeError("fp64 model mode: inputs must be float64")
        if self.whitening != "cholesky":
            raise NotImplementedError("fp64 model mode: whitening must be 'cholesky' here (CIQ goes through _ciq_step64)")

    def _hyp64(self, params):
        """(raw values, hyp[4] = {lengthscale, outputscale, noise, 0}): gpytorch Positive / GreaterThan(1e-4) softplus constraints"""
        raw = [params[k].detach().reshape(()) for k in ("raw_lengthscale", "raw_outputscale", "raw_noise")]
        ell, s, noise = F.softplus(raw[0]), F.softplus(raw[1]), F.softplus(raw[2]) + NOISE_FLOOR
        return raw, (ell, s, noise), torch.stack([ell, s, noise, torch.zeros_like(ell)]).detach().contiguous()

    def _factor64(self, ctx, params, hyp, nrhs, sync=True):
        """``sync=False``: one plain attempt whose potrf status is NOT read here (``self._info64`` holds it): the training step
        queues everything behind it and looks at the status once at its end -- a host read in the middle of the step leaves the
        GPU idle while the host queues the ~100 launches that follow"""
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        Mp = M * (p + 1)
        self._problem_size(Mp)
        self.center = Z.mean(0).contiguous()
        packZ = _ops.pack_points_f64(ctx, Z.contiguous(), V.contiguous(), p, hyp, self.center)
        L = self._get("L", (Mp, Mp), f64)
        info = self._get("info", (1,), torch.int32)
        nrhs = max(int(nrhs), Mp + 1)
        ws = self._bytes("trsm_ws", _lib.lib.dsvgp_trsm_workspace_bytes(Mp, nrhs, self.trsm_nb))
        for t in range(-1, CHOL_TRIES):                         # psd_safe_cholesky: plain, then jitter * 10^t
            _ops.kernel_fwd_f64(ctx, packZ, M, packZ, M, d, p, hyp, jitter=self.kzz_jitter, out=L)
            if t >= 0:
                _ops.add_diag_(ctx, L, self.chol_jitter * (10 ** t))
            self._potrf_ws = self._potrf_and_inverse(ctx, L, info, ws, nrhs, "kzz")
            self._info64 = info
            if not sync or int(info.item()) == 0:
                break
        else:
            raise NotPSDError("Matrix not positive definite after repeatedly adding jitter up to %.1e."
                              % (self.chol_jitter * 10 ** (CHOL_TRIES - 1)))
        self._inverse_ws = ws
        return packZ, L, (M, d, p, Mp), ws

    def _interp64(self, ctx, params, hyp, packZ, L, dims, ws, x, D):
        """K_ZX, A = L^-1 K_ZX, W = L_S^T A, mu0 = A^T m, cs = colsum(W^2 - A^2): all fp64"""
        M, d, p, Mp = dims
        B = x.shape[0]
        pd = self._pd(p)
        Bp = B * (pd + 1)
        packX = _ops.pack_points_f64(ctx, x.contiguous(), D.contiguous() if p > 0 else None, p, hyp, self.center)
        if pd != p:                                             # derivative-free data: value columns of the full block matrix
            full = self._get("Kzx_full", (Mp, B * (p + 1)), f64)
            _ops.kernel_fwd_f64(ctx, packZ, M, packX, B, d, p, hyp, out=full)
            Kzx = self._get("Kzx", (Mp, Bp), f64)
            Kzx.copy_(full[:, ::p + 1])
        else:
            Kzx = self._get("Kzx", (Mp, Bp), f64)
            _ops.kernel_fwd_f64(ctx, packZ, M, packX, B, d, p, hyp, out=Kzx)
        A = self._get("A64", (Mp, Bp), f64)
        _ops.trsm(ctx, L, Kzx, False, A, None, self.trsm_nb, ws, reuse_inverse=True)
        if self._no_middle:                                     # shared directions: zero middle term, var = prior diagonal
            mu0, _ = _ops.colstats_f64(ctx, A, None, params["variational_mean"].contiguous())
            return packX, A, None, mu0, torch.zeros_like(mu0)
        W = self._get("W", (Mp, Bp), f64)
        _ops.gemm(ctx, TRANS_A | A_UPPER, params["chol_variational_covar"], A, W)      # tril(L_S)^T A
        mu0, cs = _ops.colstats_f64(ctx, A, W, params["variational_mean"].contiguous())
        return packX, A, W, mu0, cs

    @staticmethod
    def _kl64(m, LS, Mp, num_data, dm, dLS):
        """KL(q(u) || N(0, I)) / num_data of the whitened prior (a6) and its gradients added to (dm, dLS), closed form"""
        Lt = torch.tril(LS)
        dg = torch.diagonal(Lt)
        kl = 0.5 * ((m * m).sum() + (Lt * Lt).sum() - Mp - torch.log(dg * dg).sum())
        dm.add_(m, alpha=1.0 / float(num_data))
        dLS.add_(Lt - torch.diag(1.0 / dg), alpha=1.0 / float(num_data))
        return kl / float(num_data)

    @staticmethod
    def _raw_grads64(params, scal):
        """(d lengthscale, d outputscale, d noise) of the likelihood / prior-diagonal terms -> raw parameters through the softplus
        constraints (Positive: d raw = d * sigmoid(raw); GreaterThan(1e-4): the same slope)"""
        sg = [torch.sigmoid(params[k].reshape(())) for k in ("raw_lengthscale", "raw_outputscale", "raw_noise")]
        return [scal[4] * sg[0], scal[3] * sg[1], scal[1] * sg[2]]

    @staticmethod
    def _prior_diag(B, p, pd, ell, s, like):
        """s * diag K_XX (RBFKernelDirectionalGrad.py:110-119): 1 for value rows, 1/ell^2 for derivative rows"""
        row = torch.cat([torch.ones(1, dtype=f64, device=like.device), (1.0 / ell ** 2).expand(pd)]) if pd else \
            torch.ones(1, dtype=f64, device=like.device)
        return s * row.repeat(B)

    # ---- CIQ whitening under a float64 model (train_gp(use_ciq=True): the bunny / GNN drivers offer it under their fp64 default,
    #      reference experiments/bunny/exp_bunny.py:66,78) ----
    def _ciq_step64(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, want_grads):
        """The float64 form of ``ElboEngine._ciq_step`` (reference CiqDirectionalGradVariationalStrategy.py:197-295 and its
        _NgdInterpTerms, :19-123, on a float64 model): fp64 kernel assembly, fp64 msMINRES (the ``*_f64`` entry points of
        csrc/ciq.hip), every product on the fp64 MFMA GEMM.  Same schedule, same buffers' roles, all tensors float64."""
        if self.data_outputs != "all":
            raise NotImplementedError("derivative-free data is built for the Cholesky-whitened strategy only")
        if "natural_vec" not in params:
            raise NotImplementedError("the CIQ strategy is built for a NaturalVariationalDistribution (what "
                                      "train_gp(use_ciq=True) constructs, reference directional_vi.py:164-166)")
        for k, v in params.items():
            if v.dtype != f64:
                raise TypeError("fp64 model mode: parameter %s is %s" % (k, v.dtype))
            if not v.is_cuda:
                raise _lib.DsvgpError("%s must live on the GPU: the DSVGP hot path has no CPU fallback" % k)
        if x.dtype != f64 or (D is not None and D.numel() and D.dtype != f64) or (y is not None and y.dtype != f64):
            raise TypeError("fp64 model mode: inputs must be float64")
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        Mp = M * (p + 1)
        B = x.shape[0]
        Bp = B * (p + 1)
        dev = self.device
        rows = float(Bp if global_rows is None else global_rows)
        nat_vec, nat_mat = params["natural_vec"], params["natural_mat"]
        _, (ell, s, noise), hyp = self._hyp64(params)
        self.center = Z.mean(0).contiguous()
        packZ = _ops.pack_points_f64(ctx, Z.contiguous(), V.contiguous(), p, hyp, self.center)
        packX = _ops.pack_points_f64(ctx, x.contiguous(), D.contiguous() if p > 0 else None, p, hyp, self.center)
        K = self._get("ciq_K64", (Mp, Mp), f64)
        _ops.kernel_fwd_f64(ctx, packZ, M, packZ, M, d, p, hyp, jitter=self.kzz_jitter, out=K)       # :230-234
        Rrow = self._get("ciq_R64", (Bp, Mp), f64)                                                   # K_XZ, one right-hand side per row
        _ops.kernel_fwd_f64(ctx, packX, B, packZ, M, d, p, hyp, out=Rrow)
        sigma, omega, _ = self._ciq_quadrature(ctx, K, Rrow[0])
        Q = sigma.shape[0]
        Trow = self._get("ciq_T64", (Bp, Mp), f64)
        basisF, ycoefF, rnF, its = self._ciq_solve(ctx, "f", K, Rrow, sigma, omega, Trow)           # :255-256
        self.ciq_stats.update(iterations=its)
        S64, m64, info = self._natural_moments(ctx, nat_vec, nat_mat)                                # :51-61 as a direct fp64 solve
        if int(info[0].item()) != 0:
            raise NotPSDError("natural_mat does not define a positive definite precision")
        m = m64.reshape(Mp).contiguous()
        STrow = self._get("ciq_ST64", (Bp, Mp), f64)
        _ops.gemm(ctx, 0, Trow, S64, STrow)                                                          # (S T)^T = T^T S
        const = params["constant"].reshape(1).contiguous()
        imean, mu, var, live = _ops.ciq_rowstats(ctx, Trow, STrow, p, m, const, hyp, self.ciq_kxx_jitter)      # :65-69,265-266
        if not want_grads:
            return None, None, mu, (var + noise).clamp_min_(MIN_VARIANCE)
        # the likelihood launch of the fp64 mode takes (mean without the constant, variance beyond prior diagonal + 1e-4)
        cs = var - self._prior_diag(B, p, p, ell, s, x) - KXX_JITTER
        mu, varn, mu_bar, var_bar, scal = _ops.likelihood_terms_f64(ctx, imean, cs, y.contiguous(), const, p, hyp,
                                                                    0 if mll_type == "ELBO" else 1, rows)
        loss = -scal[0] / rows                                                                       # (the KL term stays out of the loss, :74)
        grads = {k: torch.zeros_like(params[k], memory_format=torch.contiguous_format) for k in NGD_PARAM_NAMES}
        Tbar = self._get("ciq_Tbar64", (Bp, Mp), f64)
        VT = self._get("ciq_VT64", (Bp, Mp), f64)
        cvec = _ops.ciq_tbar(ctx, Trow, STrow, m, mu_bar, var_bar, live, imean, Tbar, VT)           # :94-96
        kl_bar = (1.0 / float(num_data)) if include_kl else 0.0
        d1 = grads["natural_vec"].reshape(1, Mp)
        _ops.gemm(ctx, 0, cvec.reshape(1, Bp), Trow, d1)                                             # :102-106
        d1.add_(nat_vec.reshape(1, Mp), alpha=kl_bar)                                                # :107
        d2 = grads["natural_mat"]
        _ops.gemm(ctx, TRANS_A, VT, Trow, d2)                                                        # :115-116: T^T diag(vbar) T
        d2.add_(nat_mat, alpha=kl_bar)                                                               # kl/2 (I - prec), prec = -2 theta_2
        d2.diagonal().add_(0.5 * kl_bar)
        # backward of sqrt_inv_matmul: dR = K^-1/2 Tbar, dK = -sym sum_q omega_q Y_q^T X_q (same quadrature)
        Rbar = self._get("ciq_Rbar64", (Bp, Mp), f64)
        basisB, ycoefB, rnB, its_b = self._ciq_solve(ctx, "b", K, Tbar, sigma, omega, Rbar)
        self.ciq_stats.update(iterations_backward=its_b)
        dK = self._get("ciq_dK64", (Mp, Mp), f64)
        form = self.ciq_backward_form
        if form is None:
            form = "backward" if its_b <= min(its, Q) else ("forward" if its <= Q else "shifts")
        kmin = {"backward": its_b, "forward": its, "shifts": Q}[form]
        if form == "backward":
            ctab = _ops.ciq_cross(ctx, ycoefB, its_b, ycoefF, its, omega, rnB, rnF)
            Zs = _ops.ciq_mix(ctx, basisF, its, ctab, its_b, None, self._get("ciq_Z64", (its_b, Bp, Mp), f64))
            left, right = basisB[:its_b], Zs
        elif form == "forward":
            ctab = _ops.ciq_cross(ctx, ycoefF, its, ycoefB, its_b, omega, rnF, rnB)
            Zs = _ops.ciq_mix(ctx, basisB, its_b, ctab, its, None, self._get("ciq_Z64", (its, Bp, Mp), f64))
            left, right = Zs, basisF[:its]
        else:
            om = torch.zeros(_ops.ciq_qp(Q), dtype=f64, device=dev)
            om[:Q] = omega
            right = _ops.ciq_mix(ctx, basisF, its, ycoefF * om, Q, rnF, self._get("ciq_Z64", (Q, Bp, Mp), f64))
            left = _ops.ciq_mix(ctx, basisB, its_b, ycoefB, Q, rnB, self._get("ciq_Z264", (Q, Bp, Mp), f64))
        _ops.gemm(ctx, TRANS_A, left.reshape(kmin * Bp, Mp), right.reshape(kmin * Bp, Mp), dK, alpha=-1.0)
        self.ciq_stats.update(stacked_depth=int(kmin * Bp))
        Kzzbar = self._get("ciq_Kzzbar64", (Mp, Mp), f64)
        _ops.sym_average_f64(ctx, dK, Kzzbar)
        Kb = self._get("Kb64", (Mp, Bp), f64)
        _ops.transpose_f64(ctx, Rbar, Kb)
        dZ, dV = grads["inducing_points"], grads["inducing_directions"]
        d_hyp = torch.zeros(4, dtype=f64, device=dev)
        scratch = self._get("T_zx", (Mp, Bp), f64)
        _ops.kernel_bwd_f64(ctx, Kb, packZ, M, packX, B, d, p, hyp, False, dZ, dV, d_hyp, scratch)
        scratch = self._get("T_zz", (Mp, Mp), f64)
        _ops.kernel_bwd_f64(ctx, Kzzbar, packZ, M, packZ, M, d, p, hyp, True, dZ, dV, d_hyp, scratch)
        # likelihood / prior-diagonal parts (scal) + kernel parts (d_hyp) through the softplus constraints
        d_raw = self._raw_grads64(params, scal)
        sig = [torch.sigmoid(params[k].reshape(())) for k in ("raw_lengthscale", "raw_outputscale")]
        grads["raw_lengthscale"].add_((d_raw[0] + d_hyp[0] * sig[0]).reshape(grads["raw_lengthscale"].shape))
        grads["raw_outputscale"].add_((d_raw[1] + d_hyp[1] * sig[1]).reshape(grads["raw_outputscale"].shape))
        grads["raw_noise"].add_(d_raw[2].reshape(grads["raw_noise"].shape))
        grads["constant"].add_(scal[2].reshape(grads["constant"].shape))
        return loss, grads, mu, varn

    # ---- NaturalVariationalDistribution (train_gp(use_ngd=True), reference directional_vi.py:35-37,186-187) in fp64 ----
    def _from_natural(self, ctx, params):
        """(theta_1, theta_2) -> (m, tril L_S) through the fp64 factorisations of the base engine; returns the parameter dict
        of the Cholesky parameterisation and what ``_natural_grads64`` needs, or (params, None)"""
        if "natural_vec" not in params:
            return params, None
        nv, nm = params["natural_vec"], params["natural_mat"]
        if nv.dtype != f64 or nm.dtype != f64:
            raise TypeError("fp64 model mode: natural parameters must be float64")
        Mp = nv.shape[0]
        LS64, m64, info = self._natural_moments(ctx, nv, nm)
        wsS = self._bytes("ngd_wsS", _lib.lib.dsvgp_trsm_workspace_bytes(Mp, Mp, self.trsm_nb))
        self._potrf_and_inverse(ctx, LS64, info[1:2], wsS, Mp, "ngd_LS")                # L_S (lower triangle) and L_S^-1
        bad = info.tolist()
        if bad[0] or bad[1]:
            raise NotPSDError("natural_mat does not define a positive definite precision (potrf info %s)" % bad)
        m = m64.reshape(Mp).clone()
        out = {k: v for k, v in params.items() if not k.startswith("natural_")}
        out["variational_mean"], out["chol_variational_covar"] = m, torch.tril(LS64)
        return out, (m, LS64, wsS)

    def _natural_grads64(self, ctx, grads, m, LS64, wsS):
        """(dm, dL_S) -> gradients w.r.t. the expectation parameters (``_NaturalToMuVarSqrt.backward``): d eta_2 = dS through
        the Cholesky factor of S, d eta_1 = dm - 2 dS mu; in place, all fp64"""
        Mp = m.shape[0]
        self._problem_size(Mp)
        dm, dLS = grads["variational_mean"], grads["chol_variational_covar"]
        Lbar = self._get("Lbar", (Mp, Mp), f64)
        Lbar.copy_(dLS)
        dS = self._chol_backward(ctx, LS64, Lbar, wsS, Mp)
        dLS.copy_(dS)
        t = torch.empty(Mp, 1, dtype=f64, device=self.device)
        _ops.gemm(ctx, 0, dS, m.reshape(Mp, 1).contiguous(), t)
        dm.add_(t.reshape(Mp), alpha=-2.0)

    # ---- shared inducing directions (SharedDirectionalGradVariationalStrategy.py:95-107,210-212) in fp64 ----
    def _shared_expand64(self, params):
        """tile the p shared directions over the M points, interleave the M + p variational values; the covariance of q(u) does
        not reach the predictive (zero middle term), so a unit factor stands in for it"""
        Z, Vs, ms = params["inducing_points"], params["inducing_directions"], params["variational_mean"]
        M, p = Z.shape[0], Vs.shape[0]
        if ms.shape[0] != M + p:
            raise ValueError("shared directions: q(u) has M + p = %d values, got %d" % (M + p, ms.shape[0]))
        idx = torch.cat([torch.arange(M, device=self.device).reshape(M, 1),
                         torch.arange(M, M + p, device=self.device).reshape(1, p).expand(M, p)], dim=1).reshape(-1)
        full = dict(params)
        full["inducing_directions"] = Vs.repeat(M, 1).contiguous()
        full["variational_mean"] = ms[idx].contiguous()
        full["chol_variational_covar"] = torch.zeros(1, 1, dtype=f64, device=self.device)      # never read
        return full, idx

    def _shared_step64(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl):
        full, idx = self._shared_expand64(params)
        M, p = params["inducing_points"].shape[0], params["inducing_directions"].shape[0]
        self._no_middle = True
        try:
            loss, g, mu, varn = self._loss_and_grads64(ctx, full, x, y, D, num_data, mll_type, global_rows, False, False)
        finally:
            self._no_middle = False
        ms, LS = params["variational_mean"], params["chol_variational_covar"]
        grads = {k: g[k] for k in PARAM_NAMES if k not in ("inducing_directions", "variational_mean", "chol_variational_covar")}
        grads["inducing_directions"] = g["inducing_directions"].reshape(M, p, -1).sum(0)
        dm = torch.zeros_like(ms)
        dm.index_add_(0, idx, g["variational_mean"])
        dLS = torch.zeros_like(LS, memory_format=torch.contiguous_format)
        if include_kl:                                             # KL of the (M + p)-dimensional q(u)
            Lt = torch.tril(LS)
            dg = torch.diagonal(Lt)
            kl = 0.5 * ((ms * ms).sum() + (Lt * Lt).sum() - (M + p) - torch.log(dg * dg).sum())
            loss = loss + kl / float(num_data)
            dm.add_(ms, alpha=1.0 / float(num_data))
            dLS.add_(Lt - torch.diag(1.0 / dg), alpha=1.0 / float(num_data))
        grads["variational_mean"], grads["chol_variational_covar"] = dm, dLS
        return loss, {k: grads[k] for k in PARAM_NAMES}, mu, varn

    def _shared_predict_params(self, params):
        full, _ = self._shared_expand64(params)
        return full

    # ---- public API -----------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, params, x, D, cache=False):
        ctx = _ops.Context.get(self.device)
        if self.whitening == "ciq":
            _, _, mu, varn = self._ciq_step64(ctx, params, x, None, D, 1.0, "ELBO", None, False, False)
            return mu, varn
        params, _ = self._from_natural(ctx, params)
        if self.shared_directions:
            params = self._shared_predict_params(params)
        self._check(params, x, D)
        _, (ell, s, noise), hyp = self._hyp64(params)
        Mz = params["inducing_points"].shape[0]
        pz = params["inducing_directions"].shape[0] // Mz if Mz else 0
        packZ, L, dims, ws = self._factor64(ctx, params, hyp, x.shape[0] * (self._pd(pz) + 1))
        self._no_middle = bool(self.shared_directions)
        try:
            _, _, _, mu0, cs = self._interp64(ctx, params, hyp, packZ, L, dims, ws, x, D)
        finally:
            self._no_middle = False
        p = dims[2]
        var = self._prior_diag(x.shape[0], p, self._pd(p), ell, s, x) + KXX_JITTER + cs
        return mu0 + params["constant"].reshape(()), (var + noise).clamp_min(MIN_VARIANCE)

    @torch.no_grad()
    def predict_joint(self, params, x, D, cache=False):
        """Mean [B'] and the full predictive covariance [B', B'] (fp64, likelihood noise on the diagonal):
        Sigma = s K_XX + 1e-4 I + W^T W - A^T A + noise I  (DGVS.py:199-208 + likelihood)"""
        ctx = _ops.Context.get(self.device)
        if self.whitening == "ciq":
            # NGD-CIQ: the reference's q(f) carries a DIAGONAL covariance (CiqDGVS.py:264-267), see ElboEngine.predict_joint
            _, _, mu, varn = self._ciq_step64(ctx, params, x, None, D, 1.0, "ELBO", None, False, False)
            return mu, torch.diag(varn)
        params, _ = self._from_natural(ctx, params)
        if self.shared_directions:
            params = self._shared_predict_params(params)
        self._check(params, x, D)
        _, (ell, s, noise), hyp = self._hyp64(params)
        Mz = params["inducing_points"].shape[0]
        pz = params["inducing_directions"].shape[0] // Mz if Mz else 0
        packZ, L, dims, ws = self._factor64(ctx, params, hyp, x.shape[0] * (self._pd(pz) + 1))
        self._no_middle = bool(self.shared_directions)
        try:
            packX, A, W, mu0, _ = self._interp64(ctx, params, hyp, packZ, L, dims, ws, x, D)
        finally:
            self._no_middle = False
        M, d, p, Mp = dims
        B = x.shape[0]
        Sigma = _ops.kernel_fwd_f64(ctx, packX, B, packX, B, d, p, hyp)
        if self._pd(p) != p:
            Sigma = Sigma[::p + 1, ::p + 1].contiguous()
        if W is not None:
            _ops.gemm(ctx, TRANS_A, W, W, Sigma, beta=1.0, Cin=Sigma)
            _ops.gemm(ctx, TRANS_A, A, A, Sigma, alpha=-1.0, beta=1.0, Cin=Sigma)
        Sigma.diagonal().add_(noise + KXX_JITTER)
        return mu0 + params["constant"].reshape(()), Sigma

    @torch.no_grad()
    def loss_and_grads(self, params, x, y, D, num_data, mll_type="ELBO", global_rows=None, include_kl=True, fast=None):
        """(loss, grads dict, mu, varn), all fp64; see ``ElboEngine.loss_and_grads`` for the arguments.  With natural parameters
        (``natural_vec``, ``natural_mat``) the gradients of those two slots are the expectation-parameter gradients NGD steps along."""
        if self.whitening == "ciq":
            if mll_type not in ("ELBO", "PLL"):
                raise ValueError("mll_type must be 'ELBO' or 'PLL'")
            return self._ciq_step64(_ops.Context.get(self.device), params, x, y, D, num_data, mll_type, global_rows, include_kl, True)
        if getattr(self, "deterministic", False):
            # the bitwise-reproducible mode (fixed-order split-K slabs, one stream) is built for the fp32 engine only: the fp64
            # engine's transposed gemv and split-K products sum through fp64 atomics
            raise NotImplementedError("deterministic mode covers the float32 engine (ElboEngine); the float64 model mode "
                                      "(ElboEngine64) sums split-K slices and the transposed gemv with fp64 atomics")
        ctx = _ops.Context.get(self.device)
        params, nat = self._from_natural(ctx, params)
        if self.shared_directions:
            out = self._shared_step64(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl)
        else:
            out = self._loss_and_grads64(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast)
        if nat is not None:
            loss, grads, mu, varn = out
            self._natural_grads64(ctx, grads, *nat)
            out = (loss, {_NGD_RENAME.get(k, k): v for k, v in grads.items()}, mu, varn)
        return out

    def _loss_and_grads64(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast):
        """first attempt without a host read of the potrf status inside the step; the status is read once at the end and a failed
        factorisation (rare: psd_safe_cholesky's jitter ladder) repeats the step synchronously"""
        out = self._loss_and_grads64_once(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast, False)
        if int(self._info64.item()) != 0:
            out = self._loss_and_grads64_once(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast, True)
        return out

    def _loss_and_grads64_once(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast, sync):
        self._check(params, x, D)
        self._eval_cache = None
        dev = self.device
        m = params["variational_mean"].contiguous()
        LS = params["chol_variational_covar"]
        Mz = params["inducing_points"].shape[0]
        pz = params["inducing_directions"].shape[0] // Mz if Mz else 0
        B = x.shape[0]
        pd = self._pd(pz)
        Bp = B * (pd + 1)
        if y.shape != (Bp,) or y.dtype != f64:
            raise ValueError("y must be the interleaved float64 target vector of length B*(p+1)=%d" % Bp)
        rows = float(Bp if global_rows is None else global_rows)

        if mll_type not in ("ELBO", "PLL"):
            raise ValueError("mll_type must be 'ELBO' or 'PLL'")
        if fast is None:
            fast = self.elbo_fast
        raw, _, hyp = self._hyp64(params)
        packZ, L, dims, ws = self._factor64(ctx, params, hyp, Bp, sync=sync)
        M, d, p, Mp = dims
        if fast and mll_type == "ELBO" and Mp * Bp >= self.fast_min_work:
            return self._elbo_fast64(ctx, params, hyp, packZ, L, dims, ws, x, y, D, rows, num_data, include_kl)
        packX, A, W, mu0, cs = self._interp64(ctx, params, hyp, packZ, L, dims, ws, x, D)

        # ---- O(B') likelihood terms in one launch (dsvgp_likelihood_terms_f64), O(M'^2) KL and the softplus constraints in closed
        #      form (directional_vi.py:245-249 differentiates the same expressions by autograd) ----
        mu, varn, mu_bar, var_bar, scal = _ops.likelihood_terms_f64(ctx, mu0, cs, y, params["constant"].reshape(1).contiguous(), pd,
                                                                    hyp, 0 if mll_type == "ELBO" else 1, rows)
        loss = -scal[0] / rows
        grads = {k: torch.zeros_like(params[k]) for k in PARAM_NAMES}
        dm, dLS = grads["variational_mean"], grads["chol_variational_covar"]
        if include_kl:
            loss = loss + self._kl64(m, LS, Mp, num_data, dm, dLS)
        d_con = self._raw_grads64(params, scal)

        # ---- variational parameters: m-bar += A mu_bar, L_S-bar += tril(2 A diag(var_bar) W^T) ----
        Abar = self._get("Abar", (Mp, Bp), f64)
        if W is None:                                                               # zero middle term: A-bar = m mu_bar^T
            _ops.abar_f64(ctx, A, None, m, mu_bar, var_bar, Abar, None)
        else:
            U = self._get("U", (Mp, Bp), f64)
            _ops.gemm(ctx, A_LOWER, LS, W, U)                                       # U = tril(L_S) W
            Av = self._get("Av", (Mp, Bp), f64)
            _ops.abar_f64(ctx, A, U, m, mu_bar, var_bar, Abar, Av)
            tmp = self._get("dLS_data", (Mp, Mp), f64)
            _ops.gemm(ctx, TRANS_B | OUT_LOWER, Av, W, tmp)
            dLS.add_(torch.tril(tmp))
        dmd = torch.empty(Mp, 1, dtype=f64, device=dev)
        _ops.gemm(ctx, 0, A, mu_bar.reshape(Bp, 1), dmd)
        dm.add_(dmd.reshape(-1))

        # ---- through the solve and the factorisation ----
        Kb = self._get("Kb64", (Mp, Bp), f64)
        _ops.trsm(ctx, L, Abar, True, Kb, None, self.trsm_nb, ws, reuse_inverse=True)     # K_ZX-bar = L^-T Abar
        Lbar = self._get("Lbar", (Mp, Mp), f64)
        _ops.gemm(ctx, TRANS_B | OUT_LOWER, Kb, A, Lbar, alpha=-1.0)                      # L-bar = -tril(K_ZX-bar A^T)
        self._kernel_part64(ctx, params, hyp, packZ, packX, L, Lbar, Kb, dims, ws, B, pd, grads, scal[2], d_con)
        return loss, grads, mu, varn

    def _kernel_part64(self, ctx, params, hyp, packZ, packX, L, Lbar, Kb, dims, ws, B, pd, grads, dc, d_raw, phi_arg=False):
        """K_ZX-bar and L-bar -> inducing points / directions and the kernel hyper-parameters (both formulations)"""
        M, d, p, Mp = dims
        dev = self.device
        zero = torch.zeros((), dtype=f64, device=dev)
        d_raw = [t if t is not None else zero for t in d_raw]
        dZ, dV = grads["inducing_points"], grads["inducing_directions"]
        d_hyp = torch.zeros(4, dtype=f64, device=dev)
        if pd != p:
            full = self._get("Kzx_full", (Mp, B * (p + 1)), f64)
            full.zero_()
            full[:, ::p + 1] = Kb
            Kb = full
        scratch = self._get("T_zx", (Mp, B * (p + 1)), f64)
        _ops.kernel_bwd_f64(ctx, Kb, packZ, M, packX, B, d, p, hyp, False, dZ, dV, d_hyp, scratch)
        Kzzbar = self._chol_backward(ctx, L, Lbar, ws, Mp, phi_arg=phi_arg)
        scratch = self._get("T_zz", (Mp, Mp), f64)
        _ops.kernel_bwd_f64(ctx, Kzzbar, packZ, M, packZ, M, d, p, hyp, True, dZ, dV, d_hyp, scratch)
        # softplus chain rule of the kernel hyper-parameters; d_raw already holds the likelihood / prior-diagonal parts
        sig = [torch.sigmoid(params[k].reshape(())) for k in ("raw_lengthscale", "raw_outputscale")]
        grads["raw_lengthscale"].add_((d_raw[0] + d_hyp[0] * sig[0]).reshape(grads["raw_lengthscale"].shape))
        grads["raw_outputscale"].add_((d_raw[1] + d_hyp[1] * sig[1]).reshape(grads["raw_outputscale"].shape))
        grads["raw_noise"].add_(d_raw[2].reshape(grads["raw_noise"].shape))
        grads["constant"].add_(dc.reshape(grads["constant"].shape))

    def _elbo_fast64(self, ctx, params, hyp, packZ, L, dims, ws, x, y, D, rows, num_data, include_kl):
        """ELBO through the Gram matrix G = A A^T (the fp64 form of ``ElboEngine._elbo_fast``): d loss / d var_j = vbar is the
        same for every output, so  sum_j var_j = prior + tr(L_S^T G L_S) - tr G,  L_S-bar = 2 vbar tril(G L_S),
        K_ZX-bar = [2 vbar Q' | a] [A ; mu_bar^T],  L-bar = -tril([2 vbar Q' | a] [G ; b^T])  with  Q' = L^-T (S - I), a = L^-T m,
        b = A mu_bar.  [M', B'] products: the solve, the Gram product and one dense product (the general path has six).
        The min-variance clamp of the per-output path (1e-6 on var + noise, noise >= 1e-4) is taken as inactive, as there.
        Returns an empty ``varn``."""
        M, d, p, Mp = dims
        B = x.shape[0]
        pd = self._pd(p)
        Bp = B * (pd + 1)
        dev = self.device
        m = params["variational_mean"].contiguous()
        LS = params["chol_variational_covar"]
        packX = _ops.pack_points_f64(ctx, x.contiguous(), D.contiguous() if p > 0 else None, p, hyp, self.center)
        Ae = self._get("Ae64", (Mp + 1, Bp), f64)                   # [A ; mu_bar^T]
        A = Ae[:Mp]
        Kzx = self._get("Kzx", (Mp, Bp), f64)
        if pd != p:                                                 # derivative-free data: value columns of the full block matrix
            full = self._get("Kzx_full", (Mp, B * (p + 1)), f64)
            _ops.kernel_fwd_f64(ctx, packZ, M, packX, B, d, p, hyp, out=full)
            Kzx.copy_(full[:, ::p + 1])
        else:
            _ops.kernel_fwd_f64(ctx, packZ, M, packX, B, d, p, hyp, out=Kzx)
        ev = self._event_pair()                         # (bench.py --fp64: the forward solve is the roofline entry of this mode too)
        _ops.trsm(ctx, L, Kzx, False, A, None, self.trsm_nb, ws, reuse_inverse=True)      # A = L^-1 K_ZX
        self._event_done("solve_fwd", ev)
        mu0 = torch.empty(Bp, dtype=f64, device=dev)
        _ops.gemv_f64(ctx, A, m, mu0, trans=True)                                         # A^T m
        # G = A A^T needs no gradient information, but b = A mu_bar does: the likelihood terms first
        Ge = self._get("Ge64", (Mp + 1, Mp), f64)                   # [tril(G) ; b^T] -> [G ; b^T]
        G = Ge[:Mp]
        _ops.gemm(ctx, TRANS_B | OUT_LOWER, A, A, G)
        H = self._get("H64", (Mp, Mp), f64)
        LSl = torch.tril(LS).contiguous()
        Gs = torch.tril(G)
        Gs = Gs + torch.tril(G, -1).t()                             # the symmetric G
        G.copy_(Gs)
        _ops.gemm(ctx, B_LOWER | OUT_LOWER, G, LSl, H)                                    # tril(G L_S): nothing else of it is read
        tvar = (H * LSl).sum() - torch.diagonal(G).sum()                                # tr(L_S^T G L_S) - tr G
        # scalar tail in closed form (dsvgp_elbo_fast_tail_f64: two launches; the KL term and the softplus slopes as in the general path)
        const = params["constant"].reshape(1).contiguous()
        mu, mu_bar, scal = _ops.elbo_fast_tail_f64(ctx, mu0, y.contiguous(), const, B, pd, hyp, tvar, rows)
        loss = -scal[0] / rows
        vbar = scal[5]                                              # d loss / d (sum of variances) = 1 / (2 noise rows)
        grads = {k: torch.zeros_like(params[k]) for k in PARAM_NAMES}
        dm, dLS = grads["variational_mean"], grads["chol_variational_covar"]
        if include_kl:
            loss = loss + self._kl64(m, LS, Mp, num_data, dm, dLS)
        d_raw = self._raw_grads64(params, scal)
        dLS.add_(torch.tril(H) * (2.0 * vbar))                                            # 2 vbar tril(G L_S)
        Ae[Mp].copy_(mu_bar)
        b = torch.empty(Mp, dtype=f64, device=dev)
        _ops.gemv_f64(ctx, A, mu_bar, b)                                                  # b = A mu_bar
        dm.add_(b)
        Ge[Mp].copy_(b)
        # [Q' | a] = L^-T [S - I | m], then the 2 vbar of the variance terms on the Q' block
        Se = self._get("Se64", (Mp, Mp + 2), f64)[:, :Mp + 1]       # (even leading dimension: 16-byte rows for the lean fp64 kernel)
        _ops.gemm(ctx, TRANS_B | A_LOWER | OUT_LOWER, LSl, LSl, Se[:, :Mp])               # S = L_S L_S^T: lower triangle,
        _ops.phi_symmetrize_(ctx, Se[:, :Mp])                                             # mirrored
        Se[:, :Mp].diagonal().sub_(1.0)
        Se[:, Mp].copy_(m)
        Kb = self._get("Kb64", (Mp, Bp), f64)
        Lbar = self._get("Lbar", (Mp, Mp), f64)
        if self.trsm_nb >= Mp:
            # explicit inverse in the workspace: the TRANSPOSE [Q' | a]^T = [S - I | m]^T L^-1 in one product, so that the two products
            # that follow read it as an mn-contiguous operand (the lean fp64 kernel: 62 instead of 54 TF on the dense one at C4)
            Linv = ws[:Mp * Mp * 8].view(f64).view(Mp, Mp)
            QeT = self._get("QeT64", (Mp + 1, Mp), f64)
            _ops.gemm(ctx, TRANS_A | B_LOWER, Se, Linv, QeT)
            QeT[:Mp].mul_(2.0 * vbar)
            _ops.gemm(ctx, TRANS_A, QeT, Ae, Kb)                                          # K_ZX-bar (the one dense [M', B'] product)
        else:
            Qe = self._get("Qe64", (Mp, Mp + 1), f64)
            _ops.trsm(ctx, L, Se, True, Qe, None, self.trsm_nb, ws, reuse_inverse=True)
            Qe[:, :Mp].mul_(2.0 * vbar)
            _ops.gemm(ctx, 0, Qe, Ae, Kb)
        # L-bar = -tril([2 vbar Q' | a][G ; b^T]) is not formed: the Cholesky backward reads only tril(L^T L-bar), where the tril() of
        # L-bar does not matter, and L^T Q' = S - I, L^T a = m (see ElboEngine._elbo_fast):
        #     tril(L^T L-bar) = -tril([2 vbar (S - I) | m][G ; b^T])            -- one product instead of two
        Se[:, :Mp].mul_(2.0 * vbar)
        _ops.gemm(ctx, OUT_LOWER, Se, Ge, Lbar, alpha=-1.0)
        self._kernel_part64(ctx, params, hyp, packZ, packX, L, Lbar, Kb, dims, ws, B, pd, grads, scal[2], d_raw, phi_arg=True)
        return loss, grads, mu, torch.empty(0, dtype=f64, device=dev)
