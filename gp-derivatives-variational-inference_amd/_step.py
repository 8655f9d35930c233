"""One DSVGP minibatch ELBO evaluation (forward + analytic backward) on the HIP kernels.

Follows the reference step ``loss = -mll(likelihood(model(x, derivative_directions=D)), y)``
(directionalvi/directional_vi.py:245-249) with the composition of
DirectionalGradVariationalStrategy.forward (directionalvi/DirectionalGradVariationalStrategy.py:89-208):

  forward   K_ZZ(+1e-3 I, fp32 values widened to fp64) -> rocSOLVER potrf (fp64)        :140-144,172
            K_ZX (fp32)  ->  A = L^-1 K_ZX (fp64 panel solve on MFMA, cast to fp32)     :128-137,181-183
            W = L_S^T A ; mu = A^T m + c ; var = s*diag + 1e-4 + colsum(W^2 - A^2)       :188-205
            Gaussian expected-log-lik (noise counted twice) / PLL, KL / num_data         directional_vi.py:217-246
  backward  analytic: Abar, L_S-bar, K_ZX-bar = L^-T Abar, L-bar,
            Cholesky backward, kernel backward (dZ, dV, d ell, d s), softplus chain rule.

K_XZ is not assembled (it is K_ZX^T) and the second triangular solve of the reference is not
repeated (A_t == A); everything else keeps the reference's arithmetic precision: fp32 model, fp64
Cholesky / triangular solves.
"""
import os
import time

import torch

from . import _lib, _ops
from ._lib import A_LOWER, A_UPPER, B_LOWER, OUT_LOWER, TRANS_A, TRANS_B

KZZ_JITTER = 1e-3     # LazyTensor.add_jitter() default (DGVS.py:144)
CHOL_JITTER = 1e-6    # settings.cholesky_jitter.value() (DGVS.py:74)
CHOL_TRIES = 3        # psd_safe_cholesky max_tries

PARAM_NAMES = ("inducing_points", "inducing_directions", "variational_mean", "chol_variational_covar",
               "constant", "raw_outputscale", "raw_lengthscale", "raw_noise")

# q(u) as a NaturalVariationalDistribution (reference directional_vi.py:35-37, use_ngd / use_ciq): same slots, the
# gradients returned for (natural_vec, natural_mat) are those w.r.t. the expectation parameters (natural gradient)
NGD_PARAM_NAMES = ("inducing_points", "inducing_directions", "natural_vec", "natural_mat",
                   "constant", "raw_outputscale", "raw_lengthscale", "raw_noise")
_NGD_RENAME = {"variational_mean": "natural_vec", "chol_variational_covar": "natural_mat"}

f32, f64 = torch.float32, torch.float64


class NotPSDError(RuntimeError):
    """Same failure the reference surfaces from psd_safe_cholesky (gpytorch.utils.errors.NotPSDError)."""


class _Refactored(Exception):
    """The deferred potrf status turned out non-zero: the jitter ladder must run and the step be redone."""


class ElboEngine:
    """Owns the HBM workspaces of one (M', B', d, p) configuration on one GPU."""

    def __init__(self, device, trsm_nb=None):
        self.device = torch.device(device)
        # block size of the panel triangular solve.  None = automatic: the explicit-inverse regime (one triangular MFMA
        # product per solve, inverse fused into the Cholesky launches) whenever M' <= 8192, 512-wide panels beyond
        self._trsm_nb = None if trsm_nb is None else int(trsm_nb)
        self._auto_nb = 512
        # data-parallel hook (parallel.DataParallel while a sharded step runs): .rank, .world and
        # .all_reduce_async(tensor) -> handle with .wait()
        self.collective = None
        self.global_gram = True       # ELBO fast path on > 1 rank: reduce [G ; b^T] early, split the Cholesky backward by columns
        # global-Gram schedule: the M' x M' x M' products that every rank used to repeat -- Q' = L^-T (S - I) and
        # L-bar = -tril([Q' | a][G ; b^T]) -- are SHARDED: rank g forms its column block of [Q' | a] (fp64 product, balanced: every
        # column sees the whole triangle of L^-T) and its row block of L-bar, and two fp32 all-gathers (36 MB each at M' = 3000)
        # put the whole matrices on every rank.  DSVGP_SHARD_REPLICATED=0 restores the replicated products (A/B on a node).
        self.shard_replicated = os.environ.get("DSVGP_SHARD_REPLICATED", "1") == "1"
        self.shard_min_mp = 1024      # below this M' the two extra collectives cost more than the products they shard
        self._early_handle = None
        self._allow_early = False
        self._global_gram = False
        # the early all-reduce operand ([G ; b^T] or [m-bar, L_S-bar]) travels as its packed lower triangle (SURVEY 8e:
        # 18 MB instead of 36 MB at M' = 3000); early_wire_numel = floats actually handed to the collective last step
        self.pack_reduce = True
        self.early_wire_numel = 0
        self._ctx = None
        self.variational_grads_global = False     # the last step returned m-bar / L_S-bar already summed over the ranks
        self._buf = {}
        self.chol_jitter = CHOL_JITTER  # base of the psd_safe_cholesky retry ladder (1e-8 for GradVariationalStrategy)
        self.potrf_algo = 1             # 1: blocked Cholesky on the MFMA GEMM (csrc/potrf.hip, 5.7 ms at M'=3000),
                                        # 0: rocSOLVER dpotrf (8.1 ms)
        self.elbo_fast = True           # ELBO mode: Gram-matrix formulation (see _elbo_fast)
        self._hyp_host = None
        self._pending = None            # (hyp, packZ, L, dims, info) of a factorisation whose status is not read yet
        self._host_status = self._host_info = self._status_ready = None
        self._potrf_ws = None
        self._inverse_ws = None         # the trsm workspace that holds the inverse of the current factor
        self._side = None               # second HIP stream (work overlapped with the Cholesky chain)
        # S = L_S L_S^T on the side stream as a one-workgroup-per-CU launch (DSVGP_GEMM_BACKGROUND).  Measured alternatives that
        # did not help: least-priority side stream (hipStreamCreateWithPriority), CU-masked side stream (hipExtStreamCreateWithCUMask)
        self.side_background = os.environ.get("DSVGP_SIDE_BACKGROUND", "1") == "1"
        # K_ZX-bar's kernel backward on the side stream next to the L-bar / Cholesky-backward products.  Measured: C3 8.11 -> 8.00
        # ms/step, C4 14.04-14.18 -> 14.13-14.21, the 8-rank share 5.66 -> 5.68: automatic = one GPU and B' <= 2 M' only
        _bo = os.environ.get("DSVGP_BWD_OVERLAP")
        self.bwd_overlap = None if _bo is None else _bo == "1"
        # one GPU: the L_S / m gradient kernels (need only G) on the side stream next to the Q' solve and the dense product.
        # Measured: C3 8.26 -> 8.18 ms/step, C4 unchanged (14.01 / 13.97-14.02)
        self.var_overlap = os.environ.get("DSVGP_VAR_OVERLAP", "1") == "1"
        # one-call step, probe (flag 16 of dsvgp_elbo_step_f32): L-bar and the Cholesky backward (they need [Q' | a] and G only)
        # behind the variational block on the side stream, under the dense K_ZX-bar product.  Measured (round 4, same box, twice):
        # C4 13.57-13.65 -> 13.65-13.66 ms, C3 7.79 -> 7.80-7.84, C2 0.60 -> 0.61: the dense product stretches from 3.63 to 5.23 ms
        # while the 1.6 ms tail runs beside it -- matrix-pipe time is conserved, nothing is recovered.  Off by default.
        self.tail_side = os.environ.get("DSVGP_TAIL_SIDE", "0") == "1"
        # OPT-IN (flag 32 of the one-call step, `bench.py --split-bf16`): the Gram product and the dense K_ZX-bar product as bf16 x 3
        # split products on the bf16 matrix pipe (six bf16 MFMA products per fp32 product, fp32 accumulation; csrc/gemm3b.hip).
        # The default computes them on v_mfma_f32_32x32x2_f32.
        self.split_bf16 = os.environ.get("DSVGP_SPLIT_BF16", "0") == "1"
        # flag 64 of the one-call step: tril(L^T L-bar) = -tril([S - I | m'][G ; b^T]) with fp64 accumulation.  Default off: both
        # operands are fp32 data and G carries sqrt(M') times the error fp32 accumulation adds, so the product runs on the fp32
        # LDS-DMA kernel and only its result is widened (csrc/step.hip, chol_tail)
        self.phi_arg_fp64 = os.environ.get("DSVGP_PHI_ARG_FP64", "0") == "1"
        # flag 128 of the one-call step: the forward solve A = L^-1 K_ZX in row ranges, the first two on the side stream under the
        # Cholesky chain's later launches (csrc/step.hip, step_front); same arithmetic per output element
        self.solve_pipe = os.environ.get("DSVGP_SOLVE_PIPE", "0") == "1"
        self.dp_host_trace = None           # list: host time stamps of every data-parallel rank step's phases (_c_step_dp_phases)
        self._side_done = None
        # K_ZX assembly + S = L_S L_S^T on a second stream under the Cholesky chain.  None = automatic: only from M' = 2048 up
        # (at M' = 600 the fork / join costs more than the overlap returns: 0.88 vs 0.76 ms per step; +0.05 ms gain at M' = 3000)
        self.overlap = None
        # whitening: "cholesky" (DirectionalGradVariationalStrategy) or "ciq" (CiqDirectionalGradVariationalStrategy:
        # K_ZZ^{-1/2} by contour-integral quadrature + msMINRES; needs natural parameters)
        self.whitening = "cholesky"
        # "all": every data point carries p directional derivatives (DSVGP); "values": derivative-free data, only
        # function values on the data side (reference DFreeDirectionalGradVariationalStrategy.py:113-136)
        self.data_outputs = "all"
        # True: ONE set of p inducing directions shared by all inducing points, q(u) over M + p values, zero middle term
        # (reference SharedDirectionalGradVariationalStrategy.py:95-107,210-212)
        self.shared_directions = False
        self._no_middle = False
        self.fused_inverse = True       # L^-1 by forward elimination inside the Cholesky launches (csrc/potrf.hip)
        self.lib_dense_gemm = False     # True (diagnostics / test comparator only): the dense K_ZX-bar product through rocBLAS
        # K_ZZ jitter: LazyTensor.add_jitter() default 1e-3 (DGVS.py:144, CiqDGVS.py:233); gpytorch's plain CiqVariationalStrategy
        # (grad_svgp.py:25-27, traditional_vi.py:22-24; its forward is quoted in CiqDGVS.py:243-251) adds 1e-2 to K_ZZ and 1e-4 to
        # diag K_XX instead
        self.kzz_jitter = KZZ_JITTER
        self.ciq_kxx_jitter = 0.0
        self.ciq_num_quadrature = 15        # train_gp(num_contour_quadrature=15)
        self.ciq_tolerance = 1e-4           # gpytorch settings.minres_tolerance
        self.ciq_max_iter = 1000            # gpytorch settings.max_cg_iterations
        self.ciq_capacity = None            # Lanczos rows kept per solve (csrc/ciq.hip): None = as many as fit 4 GiB (at least
                                            # 20, at most ciq_max_iter); doubled (and the solve repeated) when one needs more
        self.ciq_backward_form = None       # stacking of the backward's sum over shifts: None = the shortest of "backward" /
                                            # "forward" (rows of that solve's basis) / "shifts" (the Q materialised solves)
        self.ciq_stats = {}                 # lmin / lmax / iterations of the last CIQ forward + backward
        self.ciq_eig_bounds = None          # (lmin, lmax) to build the quadrature on instead of the 20-step Lanczos estimate
        self._eval_cache = None
        # True while a step is being captured into / replayed from a HIP graph (directional_vi.TrainLoop): nothing may read
        # device results on the host -- the potrf status is checked by the caller one step later, and the Adam kernels of the
        # captured step are guarded by the status word instead
        self.capture_mode = False
        self.record_events = False      # bench.py: HIP-event timing of the dominant kernel on the launch stream
        self.events = []
        # True: bitwise reproducible steps -- every split-K product adds its K slices from scratch slabs in a fixed order instead of
        # meeting in floating-point atomics, scalar sums go through per-workgroup partials (dsvgp_set_deterministic), and the step
        # runs on ONE stream (the slab serves one stream at a time).  The CPU reference is deterministic for a fixed seed; the
        # default HIP step is not (order of atomic adds).  Measured cost: DESIGN.md section 5.
        self.deterministic = os.environ.get("DSVGP_DETERMINISTIC", "0") == "1"
        # The ELBO fast path of one rank from ONE host call (dsvgp_elbo_step_f32, csrc/step.hip) instead of ~100 ctypes calls:
        # same library entry points in the same order, queued from C.  Anything outside its scope (PLL, per-output variances,
        # CIQ, shared directions, derivative-free data, data-parallel schedules, graph capture, the jitter ladder) keeps the
        # piecewise path below.  DSVGP_C_STEP=0 switches it off (A/B runs).
        self.c_step = os.environ.get("DSVGP_C_STEP", "1") == "1"
        self._plans = {}
        # deferred status (round 6, TrainLoop._eager_step): the one-call step returns WITHOUT waiting for the factorisation's status; the caller
        # queues its parameter update guarded on the device by the status word (deferred_guard) and reads the status before the next step
        # (deferred_check): the ~100 us the host waited per step at M' = 600 leave its critical path
        self.defer_status = False
        self._deferred = None                   # (plan, workspace) of a step whose status has not been read yet
        self._zx_dirs = None                    # (idx, base) of the batch in flight when its directions were stated one-hot (_ops.state_directions)
        self.c_step_used = False        # whether the last step ran through the one-call path
        self.c_step_timed = []          # plans of the steps queued with record_events on, in order
        self.record_every = 1           # bench.py: HIP events on every record_every-th one-call step only (an event record costs the
        self._rec_count = 0             # stream ~5 us: six of them are 5 % of a 0.6 ms step, nothing of a 14 ms one)
        self.host_trace = None

    @property
    def trsm_nb(self):
        return self._trsm_nb if self._trsm_nb is not None else self._auto_nb

    @trsm_nb.setter
    def trsm_nb(self, value):
        self._trsm_nb = None if value is None else int(value)

    def _problem_size(self, Mp):
        """resolve the automatic solve regime for an M' x M' inducing system"""
        self._auto_nb = (1 << max(6, (int(Mp) - 1).bit_length())) if Mp <= 8192 else 512

    # ---- workspace management ---------------------------------------------------------------
    def _get(self, name, shape, dtype):
        t = self._buf.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    def _get_zeroed(self, name, shape, dtype):
        """like ``_get`` but zero-filled when (re)allocated: buffers whose padding must read as zero"""
        t = self._buf.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    def _bytes(self, name, nbytes):
        t = self._buf.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)
            self._buf[name] = t
        return t

    # ---- shared forward pieces ----------------------------------------------------------------
    def _factor(self, ctx, params, sync=True, side_job=None, nrhs=0):
        """hyp, packs of (Z,V), L = chol(K_ZZ + 1e-3 I) with psd_safe_cholesky retries.
        ``side_job(ctx, hyp)``: work that needs only the hyper-parameters / the centre (K_ZX assembly, S = L_S L_S^T):
        queued on a second HIP stream so that it fills the CUs the latency-bound Cholesky chain leaves idle; the
        caller waits on ``self._side_done`` before consuming its results.
        ``sync=False`` only enqueues the first factorisation; the status word (and the host copy of the
        hyper-parameters) is read later by ``_finish_factor`` once more GPU work has been queued behind it, so the
        host round trip does not drain the device."""
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        Mp = M * (p + 1)
        self._problem_size(Mp)
        hyp = _ops.hyp_forward(ctx, params["raw_lengthscale"], params["raw_outputscale"], params["raw_noise"])
        # common shift of Z and x (gpytorch covar_dist centres on x1.mean): keeps the fp32 quadratic expansion accurate
        self.center = _ops.column_mean(ctx, Z.contiguous())
        packZ = _ops.pack_points(ctx, Z.contiguous(), V.contiguous(), p, hyp, self.center)
        self._pending_packZ = packZ
        if side_job is not None:
            main = torch.cuda.current_stream(self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                ctx.bind()
                side_job(ctx, hyp)
                self._side_done = torch.cuda.Event()
                self._side_done.record(self._side)
            ctx.bind()                                  # back on the caller's stream
        L = self._get("L", (Mp, Mp), f64)
        info = self._get("info", (1,), torch.int32)
        _ops.kernel_fwd(ctx, packZ, M, packZ, M, d, p, hyp, jitter=self.kzz_jitter, out=L, dtype=f64)
        nrhs = max(int(nrhs), Mp + 1)
        ws = self._bytes("trsm_ws", _lib.lib.dsvgp_trsm_workspace_bytes(Mp, nrhs, self.trsm_nb))
        self._potrf_ws = self._potrf_and_inverse(ctx, L, info, ws, nrhs, "kzz")   # L and the inverted blocks of L
        self._inverse_ws = ws
        self._pending = (hyp, packZ, L, (M, d, p, Mp), info)
        # status word + hyper-parameters go to pinned host memory right behind the factorisation: the later read waits for
        # THIS point of the stream only, not for the solve / Gram product queued after it (no idle device around the read)
        if self._host_status is None:
            self._host_status = torch.empty(16, dtype=f32).pin_memory()
            self._host_info = torch.empty(1, dtype=torch.int32).pin_memory()
        self._host_status[:hyp.numel()].copy_(hyp.reshape(-1), non_blocking=True)
        self._host_info.copy_(info, non_blocking=True)
        self._status_ready = torch.cuda.Event()
        self._status_ready.record(torch.cuda.current_stream(self.device))
        if sync:
            self._finish_factor(ctx, ladder=True)
        return hyp, packZ, L, (M, d, p, Mp)

    def _potrf_scratch(self, name, n):
        """Cholesky scratch of ONE factor of THIS engine (``name``: kzz / ngd_P / ngd_LS / root).  It keeps the inverted
        64 x 64 diagonal blocks that a later ``trtri_blocks`` of the same factor copies verbatim, so it is never shared
        between factors, engines or models."""
        return self._bytes("potrf_ws_" + name, _lib.lib.dsvgp_potrf_workspace_bytes(int(n), 1))

    def _potrf_and_inverse(self, ctx, L, info, ws, nrhs, name):
        """L <- chol(L) and the inverted blocks of L into the trsm workspace ``ws``: ONE fused chain of launches when the
        explicit-inverse regime applies (blocked algorithm, nb >= n), else potrf followed by the trtri recursion."""
        n = L.shape[0]
        scratch = self._potrf_scratch(name, n) if self.potrf_algo == 1 else None
        if self.potrf_algo == 1 and self.trsm_nb >= n and self.fused_inverse:
            return _ops.potrf_inverse_(ctx, L, info, self.trsm_nb, ws, scratch)
        pws = _ops.potrf_(ctx, L, info, self.potrf_algo, scratch)
        _ops.trtri_blocks(ctx, L, nrhs, self.trsm_nb, ws, pws)
        return pws

    def _finish_factor(self, ctx, ladder=False):
        """Read the potrf status (host sync).  Non-zero: run psd_safe_cholesky's jitter ladder here (``ladder``) or
        tell the caller to start over synchronously (``_Refactored``)."""
        if self._pending is None:
            return
        hyp, packZ, L, (M, d, p, Mp), info = self._pending
        self._pending = None
        self._status_ready.synchronize()
        self._hyp_host = self._host_status[:hyp.numel()].tolist()      # host copy of (ell, s, noise)
        if int(self._host_info[0]) == 0:
            return
        if not ladder:
            raise _Refactored()
        for t in range(CHOL_TRIES):                     # rare path: psd_safe_cholesky jitter ladder
            _ops.kernel_fwd(ctx, packZ, M, packZ, M, d, p, hyp, jitter=self.kzz_jitter, out=L, dtype=f64)
            _ops.add_diag_(ctx, L, self.chol_jitter * (10 ** t))
            self._potrf_ws = self._potrf_and_inverse(ctx, L, info, self._inverse_ws, Mp + 1, "kzz")
            if int(info.item()) == 0:
                return
        raise NotPSDError("Matrix not positive definite after repeatedly adding jitter up to %.1e."
                          % (self.chol_jitter * 10 ** (CHOL_TRIES - 1)))

    def _pd(self, p):
        """directional derivatives per DATA point"""
        return 0 if self.data_outputs == "values" else p

    def _assemble_kzx(self, ctx, packZ, M, packX, B, d, p, hyp, Mp):
        """K_ZX [M', B(pd+1)].  Derivative-free data: the value columns (every (p+1)-th) of the full block matrix."""
        if self.data_outputs == "values" and p > 0:
            full = self._get("Kzx_full", (Mp, B * (p + 1)), f32)
            _ops.kernel_fwd(ctx, packZ, M, packX, B, d, p, hyp, out=full)
            Kzx = self._get("Kzx", (Mp, B), f32)
            Kzx.copy_(full[:, ::p + 1])
            return Kzx
        Kzx = self._get("Kzx", (Mp, B * (p + 1)), f32)
        ev = self._event_pair()                          # (on whichever stream the assembly is queued: main or side)
        st = self._zx_dirs
        if st is not None:                               # one-hot shared directions stated by the caller (_ops.state_directions)
            _ops.kernel_fwd_canon(ctx, packZ, M, packX, B, d, p, st[0], st[1], hyp, out=Kzx)
        else:
            _ops.kernel_fwd(ctx, packZ, M, packX, B, d, p, hyp, out=Kzx)
        self._event_done("assemble_fwd", ev)
        return Kzx

    def event_durations(self, name):
        """durations (seconds) recorded for ``name`` (solve_fwd / assemble_fwd / assemble_bwd / ...) since ``events`` /
        ``c_step_timed`` were last cleared: torch events of the piecewise path and the plan's HIP events of the one-call path"""
        out = [s.elapsed_time(e) * 1e-3 for (nm, s, e) in self.events if nm == name]
        slot = {"solve_fwd": 0, "assemble_fwd": 1, "assemble_bwd": 2, "gram": 3, "dense": 4}.get(name)
        if slot is not None and self.c_step_timed:
            for plan, idx in reversed(self.c_step_timed):      # (absolute index of the timed step in its plan's ring)
                b = plan.timed_count() - 1 - idx
                if 0 <= b < 128:
                    out.append(plan.timings(b)[slot] * 1e-3)
        return out

    def _event_pair(self):
        if not self.record_events or self.capture_mode:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0, e1

    def _event_done(self, name, ev):
        if ev is not None:
            ev[1].record()
            self.events.append((name, ev[0], ev[1]))

    def _kernel_bwd_zx(self, ctx, Kb32, packZ, M, packX, B, d, p, hyp, dZ, dV, d_hyp, kws):
        if self.data_outputs == "values" and p > 0:
            full = self._get("Kzx_full", (Kb32.shape[0], B * (p + 1)), f32)
            full.zero_()
            full[:, ::p + 1] = Kb32                      # derivative columns carry no gradient
            Kb32 = full
        ev = self._event_pair()
        st = self._zx_dirs if self.data_outputs != "values" else None
        if st is not None:
            _ops.kernel_bwd_canon(ctx, Kb32, packZ, M, packX, B, d, p, st[0], st[1], hyp, dZ, dV, d_hyp, kws)
        else:
            _ops.kernel_bwd(ctx, Kb32, packZ, M, packX, B, d, p, hyp, False, dZ, dV, d_hyp, kws)
        self._event_done("assemble_bwd", ev)

    def _interp(self, ctx, params, hyp, packZ, L, dims, x, D, reuse_inverse=False):
        """K_ZX, A = L^-1 K_ZX (fp64 + fp32 copy), W = L_S^T A, mu, var."""
        M, d, p, Mp = dims
        B = x.shape[0]
        pd = self._pd(p)
        Bp = B * (pd + 1)
        packX = _ops.pack_points(ctx, x.contiguous(), D.contiguous() if p > 0 else None, p, hyp, self.center)
        self._zx_dirs = _ops.stated_directions(D, d, p) if p > 0 else None
        Kzx = self._assemble_kzx(ctx, packZ, M, packX, B, d, p, hyp, Mp)
        A64 = self._get("A64", (Mp, Bp), f64)
        A32 = self._get("A32", (Mp, Bp), f32)
        need = _lib.lib.dsvgp_trsm_workspace_bytes(Mp, max(Bp, Mp), self.trsm_nb)
        old = self._buf.get("trsm_ws")
        ws = self._bytes("trsm_ws", need)
        if not ((reuse_inverse and ws is old) or ws is self._inverse_ws):   # a re-allocated workspace has no inverse in it
            _ops.trtri_blocks(ctx, L, max(Bp, Mp), self.trsm_nb, ws, self._potrf_ws)
            self._inverse_ws = ws
        ev = self._event_pair()
        _ops.trsm(ctx, L, Kzx, False, A64, A32, self.trsm_nb, ws, reuse_inverse=True)
        self._event_done("solve_fwd", ev)
        LS = params["chol_variational_covar"]
        if self._no_middle:
            W = A32                         # zero middle term: colsum(W^2 - A^2) vanishes
        else:
            W = self._get("W", (Mp, Bp), f32)
            # W = tril(L_S)^T A : op(A) = L_S^T is upper triangular -> the strict upper part of the
            # parameter is never read (CholeskyVariationalDistribution masks it with tril)
            _ops.gemm(ctx, TRANS_A | A_UPPER, LS, A32, W)
        mu = torch.empty(Bp, dtype=f32, device=self.device)
        var = torch.empty(Bp, dtype=f32, device=self.device)
        sws = self._bytes("stats_ws", _lib.lib.dsvgp_stats_workspace_bytes(Mp, Bp))
        _ops.predictive_stats(ctx, A32, W, pd, params["variational_mean"], params["constant"].reshape(-1), hyp, mu, var,
                              sws)
        return packX, A64, A32, W, mu, var

    # ---- public API -----------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, params, x, D, cache=False):
        """q(f) mean / variance plus likelihood noise: what ``likelihood(model(x)).mean/.variance`` returns.
        ``cache=True`` (eval mode) keeps the Cholesky factor and its inverted blocks across calls while the
        parameters are unchanged, like the reference's ``@cached`` ``_cholesky_factor`` (DGVS.py:72)."""
        ctx = _ops.Context.get(self.device)
        if self.whitening == "ciq":
            _, _, mu, varn = self._ciq_step(ctx, params, x, None, D, 1.0, "ELBO", None, False, False)
            return mu, varn
        if self.shared_directions:
            if "natural_vec" in params:                 # natural q(u) over the M + p shared values
                m32, LS32, _, _ = self._natural_to_mu_chol(ctx, params["natural_vec"], params["natural_mat"])
                params = {k: v for k, v in params.items() if not k.startswith("natural_")}
                params["variational_mean"], params["chol_variational_covar"] = m32, LS32
            params, _ = self._shared_expand(params)
            self._no_middle = True
            try:
                return self._predict_chol(ctx, params, x, D, cache)
            finally:
                self._no_middle = False
        return self._predict_chol(ctx, params, x, D, cache)

    @torch.no_grad()
    def value_variances(self, params):
        """Predictive variances WITH likelihood noise of the function-value rows ([::p+1]) of the ELBO fast-path step queued
        last -- what the reference's every-50th-step report reads (``output.variance.sqrt()[::num_directions + 1]``,
        directional_vi.py:256-258) -- from the A = L^-1 K_ZX that step left in its workspace: W = L_S^T A[:, ::p+1] is one
        triangular [M', M'] x [M', B] fp32 product (1/(p+1) of the per-output path's W, and none of its U / per-output backward).
        Valid until the parameters change or the next step is queued: the training loop calls it BEFORE the optimizers step."""
        lf = getattr(self, "_last_fast", None)
        if lf is None or "chol_variational_covar" not in params:
            raise RuntimeError("value_variances needs an ELBO fast-path step of the Cholesky-whitened strategy queued last")
        ctx = _ops.Context.get(self.device)
        if lf[0] == "plan":
            _, plan, ws, p = lf
            A32 = plan.locate(ws, 0)[:-1]
            hyp = plan.locate(ws, 4).reshape(-1)
            q = p + 1
        else:
            _, name, hyp, pd = lf
            A32 = self._buf[name][:-1]
            q = pd + 1
        Mp, Bp = A32.shape
        B = Bp // q
        Asub = self._get("vv_A", (Mp, B), f32)
        Asub.copy_(A32[:, ::q])
        W = self._get("vv_W", (Mp, B), f32)
        _ops.gemm(ctx, TRANS_A | A_UPPER, params["chol_variational_covar"], Asub, W)          # W = tril(L_S)^T A[:, ::q]
        mu = torch.empty(B, dtype=f32, device=self.device)
        var = torch.empty(B, dtype=f32, device=self.device)
        sws = self._bytes("vv_stats_ws", _lib.lib.dsvgp_stats_workspace_bytes(Mp, B))
        _ops.predictive_stats(ctx, Asub, W, 0, params["variational_mean"], params["constant"].reshape(-1), hyp, mu, var, sws)
        return (var + hyp[2]).clamp_min_(1e-6)

    @torch.no_grad()
    def predict_joint(self, params, x, D, cache=False):
        """Mean [B'] and the FULL predictive covariance [B', B'] (fp32, likelihood noise on the diagonal): the
        MultivariateNormal behind ``likelihood(model(x, derivative_directions=D))`` (reference DGVS.py:199-208), which the
        BO drivers sample jointly (experiments/GNN_bo/gcn_turbo.py:238-239).
        Sigma = s K_XX + 1e-4 I + W^T W - A^T A + noise I: one symmetric kernel assembly and two MFMA Gram products."""
        ctx = _ops.Context.get(self.device)
        if self.whitening == "ciq":
            # NGD-CIQ: the reference's q(f) carries a DIAGONAL covariance, DiagLazyTensor(predictive_var) (CiqDGVS.py:264-267);
            # the likelihood adds its noise on that diagonal -- joint samples are independent draws
            _, _, mu, varn = self._ciq_step(ctx, params, x, None, D, 1.0, "ELBO", None, False, False)
            return mu, torch.diag(varn)
        if self.shared_directions:
            if "natural_vec" in params:
                m32, LS32, _, _ = self._natural_to_mu_chol(ctx, params["natural_vec"], params["natural_mat"])
                params = {k: v for k, v in params.items() if not k.startswith("natural_")}
                params["variational_mean"], params["chol_variational_covar"] = m32, LS32
            params, _ = self._shared_expand(params)
            self._no_middle = True
            try:
                return self._predict_chol(ctx, params, x, D, cache, joint=True)
            finally:
                self._no_middle = False
        return self._predict_chol(ctx, params, x, D, cache, joint=True)

    @torch.no_grad()
    def whiten_legacy(self, params):
        """Un-whitened q(u) = N(m_u, L_u L_u^T) of a checkpoint written before gpytorch's whitened VariationalStrategy ->
        the whitened parameters this strategy works with (reference DGVS.py:210-240):
            L = chol(K_ZZ + 1e-3 I),  m_w = L^-1 (m_u - c),  L_w = chol(L^-1 S_u L^-T)
        (fp64 throughout, like the reference's ``.double()`` solves).  The reference evaluates the prior p(u) through
        ``self(inducing_points, prior=True)`` WITHOUT direction kwargs, which its directional kernel cannot take: the
        model's own inducing directions are used here, the evident intent.  Returns (m_w fp32, L_w fp32 lower)."""
        ctx = _ops.Context.get(self.device)
        self._eval_cache = None
        m_u, L_u = params["variational_mean"], params["chol_variational_covar"]
        Mp = m_u.shape[0]
        hyp, packZ, L, dims = self._factor(ctx, params, sync=True, nrhs=Mp + 1)
        rhs = torch.empty(Mp, Mp + 1, dtype=f64, device=self.device)
        rhs[:, 0] = (m_u - params["constant"].reshape(())).to(f64)
        rhs[:, 1:] = torch.tril(L_u).to(f64)
        X = torch.empty_like(rhs)
        _ops.trsm(ctx, L, rhs, False, X, None, self.trsm_nb, self._inverse_ws, reuse_inverse=True)     # L^-1 [m_u - c | L_u]
        R = X[:, 1:].contiguous()
        Sw = torch.empty(Mp, Mp, dtype=f64, device=self.device)
        _ops.gemm(ctx, TRANS_B, R, R, Sw)                                                             # L^-1 S_u L^-T
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        _ops.potrf_(ctx, Sw, info, self.potrf_algo, self._potrf_scratch("legacy", Mp) if self.potrf_algo == 1 else None)
        if int(info.item()) != 0:
            raise NotPSDError("the un-whitened variational covariance of the checkpoint is not positive definite")
        return X[:, 0].to(f32).contiguous(), torch.tril(Sw).to(f32).contiguous()

    @torch.no_grad()
    def prior_moments(self, params):
        """p(u) = N(c 1, s K(Z, Z; V, V)) at the inducing points, un-jittered, fp32: ``strategy(Z, prior=True)``"""
        ctx = _ops.Context.get(self.device)
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        hyp = _ops.hyp_forward(ctx, params["raw_lengthscale"], params["raw_outputscale"], params["raw_noise"])
        center = _ops.column_mean(ctx, Z.contiguous())
        packZ = _ops.pack_points(ctx, Z.contiguous(), V.contiguous(), p, hyp, center)
        K = _ops.kernel_fwd(ctx, packZ, M, packZ, M, d, p, hyp)
        return params["constant"].reshape(()).expand(M * (p + 1)).clone(), K

    @torch.no_grad()
    def covariance_root(self, Sigma):
        """Lower Cholesky factor (fp64) of a predictive covariance, with the psd_safe_cholesky jitter ladder."""
        ctx = _ops.Context.get(self.device)
        n = Sigma.shape[0]
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        jit = 0.0
        for t in range(-1, 3):
            R = Sigma.to(f64, copy=True)        # (factorised in place: never the caller's matrix)
            if t >= 0:
                jit = self.chol_jitter * 10.0 ** t
                _ops.add_diag_(ctx, R, jit)
            info.zero_()
            _ops.potrf_(ctx, R, info, self.potrf_algo, self._potrf_scratch("root", n) if self.potrf_algo == 1 else None)
            if int(info.item()) == 0:
                return R
        raise NotPSDError("Matrix not positive definite after repeatedly adding jitter up to %g" % jit)

    @torch.no_grad()
    def draw(self, mu, root, eps):
        """mu + tril(root) eps_i for every row eps_i of eps [n, B']: one triangular fp64 MFMA product.  Returns [n, B'] in mu's dtype."""
        ctx = _ops.Context.get(self.device)
        out = torch.empty(mu.shape[0], eps.shape[0], dtype=f64, device=self.device)
        _ops.gemm(ctx, A_LOWER, root, eps.t().contiguous(), out)
        return out.t().to(mu.dtype) + mu

    def _predict_chol(self, ctx, params, x, D, cache, joint=False):
        key = tuple((t.data_ptr(), t._version) for t in params.values()) if cache else None
        hit = cache and self._eval_cache is not None and self._eval_cache[0] == key
        if "natural_vec" in params:
            if hit:
                params = self._eval_cache[5]
            else:
                m32, LS32, _, _ = self._natural_to_mu_chol(ctx, params["natural_vec"], params["natural_mat"])
                params = {k: v for k, v in params.items() if not k.startswith("natural_")}
                params["variational_mean"], params["chol_variational_covar"] = m32, LS32
        if hit:
            _, hyp, packZ, L, dims, _ = self._eval_cache
        else:
            Mz = params["inducing_points"].shape[0]
            pz = params["inducing_directions"].shape[0] // Mz if Mz else 0
            hyp, packZ, L, dims = self._factor(ctx, params, nrhs=x.shape[0] * (self._pd(pz) + 1))
            self._eval_cache = (key, hyp, packZ, L, dims, params) if cache else None
        packX, _, A32, W, mu, var = self._interp(ctx, params, hyp, packZ, L, dims, x, D, reuse_inverse=hit)
        if joint:
            M, d, p, Mp = dims
            B = x.shape[0]
            Sigma = _ops.kernel_fwd(ctx, packX, B, packX, B, d, p, hyp)            # s K(X, X; D, D), all (p+1)^2 blocks
            if self.data_outputs == "values" and p > 0:
                Sigma = Sigma[::p + 1, ::p + 1].contiguous()
            if not self._no_middle:
                _ops.gemm(ctx, TRANS_A, W, W, Sigma, beta=1.0, Cin=Sigma)          # + W^T W
                _ops.gemm(ctx, TRANS_A, A32, A32, Sigma, alpha=-1.0, beta=1.0, Cin=Sigma)   # - A^T A
            Sigma.diagonal().add_(hyp[2] + 1e-4)                                   # add_jitter(1e-4) + likelihood noise
            return mu, Sigma
        varn = (var + hyp[2]).clamp_min_(1e-6)
        return mu, varn

    @torch.no_grad()
    def loss_and_grads(self, params, x, y, D, num_data, mll_type="ELBO", global_rows=None, include_kl=True,
                       fast=None):
        """Returns (loss, grads dict, mu, varn).  ``global_rows`` = B'(global) for data-parallel ranks;
        ``include_kl=False`` leaves the (replicated) KL term out so exactly one rank adds it.
        ``fast`` (default ``self.elbo_fast``): in ELBO mode use the Gram-matrix formulation, which does not
        produce per-output variances (``varn`` is then an empty tensor; ``predict`` gives them on demand)."""
        ctx = _ops.Context.get(self.device)
        self._eval_cache = None
        if fast is None:
            fast = self.elbo_fast
        if self.deterministic:
            Mq = params["natural_vec" if "natural_vec" in params else "variational_mean"].shape[0]
            Mq = max(Mq, params["inducing_points"].shape[0] * (params["inducing_directions"].shape[0] // max(params["inducing_points"].shape[0], 1) + 1))
            # room for 5 fp64 slabs of an [M', M' + 1] product (the fp32 Gram product's 5-6 slices need half of that)
            ctx.set_deterministic(self._bytes("det_slab", max(1 << 20, 5 * 8 * Mq * (Mq + 4))))
            try:
                return self._loss_and_grads_entry(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast)
            finally:
                ctx.set_deterministic(None)
        return self._loss_and_grads_entry(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast)

    def _loss_and_grads_entry(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast):
        self._allow_early = False
        self._early_handle = None
        self.variational_grads_global = False
        if self.whitening == "ciq":
            return self._ciq_step(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, True)
        nat = None
        if "natural_vec" in params:
            # NaturalVariationalDistribution.forward: (theta_1, theta_2) -> (mu, chol S); the step itself is unchanged
            m32, LS32, LS64, wsS = self._natural_to_mu_chol(ctx, params["natural_vec"], params["natural_mat"])
            nat = (m32, LS64, wsS)
            params = {k: v for k, v in params.items() if not k.startswith("natural_")}
            params["variational_mean"], params["chol_variational_covar"] = m32, LS32
        self._allow_early = nat is None and not self.shared_directions     # (natural / shared parameterisations post-process m-bar and L_S-bar)
        if self.shared_directions:
            out = self._shared_step(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl)
        else:
            try:    # first attempt: potrf status read only after the forward solve has been queued
                out = self._loss_and_grads(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast, False)
            except _Refactored:
                out = self._loss_and_grads(ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast, True)
        if nat is not None:
            loss, grads, mu, varn = out
            self._natural_grads(ctx, grads, *nat)
            out = (loss, {_NGD_RENAME.get(k, k): v for k, v in grads.items()}, mu, varn)
        return out

    def _alloc_grads(self, params, names, zero=True):
        """All gradients + the loss in ONE flat buffer (one fill; the data-parallel all-reduce needs no packing), the
        variational parameters (names[2:4], 99.8 % of the bytes) first: they are final half-way through the backward, so
        the data-parallel layer reduces ``flat_early`` while the rest of the step runs and ``flat_late`` at the end.
        Returns (dict of views, loss slot, d_hyp[4])."""
        order = list(names[2:4]) + list(names[:2]) + list(names[4:])
        pad = lambda nk: (nk + 15) // 16 * 16                       # every segment starts 64-byte aligned
        total = sum(pad(params[k].numel()) for k in order)
        # (zero=False: the callee clears it; the length is rounded up to 64 floats so that one fill kernel does it)
        flat = (torch.zeros if zero else torch.empty)((total + 1 + 4 + 63) // 64 * 64, dtype=f32, device=self.device)
        self._flat_full = flat
        views, off = {}, 0
        for k in order:
            nk = params[k].numel()
            views[k] = flat[off:off + nk].view(params[k].shape)
            off += pad(nk)
        early = sum(pad(params[k].numel()) for k in order[:2])
        self.flat = flat[:off + 1]                      # [grads (padded)..., loss]
        self.flat_early, self.flat_late = flat[:early], flat[early:off + 1]
        self._early_handle = None
        v0, v1 = views[order[0]], views[order[1]]
        self._early_views = (v0, v1) if (v1.dim() == 2 and v1.shape[0] == v1.shape[1] == v0.numel()) else None
        return {k: views[k] for k in names}, flat[off:off + 1], flat[off + 1:off + 5]

    def _variational_grads_final(self):
        """called once m-bar and L_S-bar are complete: hand them to the data-parallel layer (asynchronous all-reduce)"""
        if self.collective is not None and self._allow_early and self.collective.world > 1:
            views = getattr(self, "_early_views", None)
            if self.pack_reduce and views is not None:
                dm, dLS = views                  # L_S-bar is lower triangular: send [tril(L_S-bar) | m-bar]
                self._early_handle = self._reduce_tril_async(self._ctx, self.collective, dLS, dm.reshape(-1))
            else:
                self.early_wire_numel = self.flat_early.numel()
                self._early_handle = self.collective.all_reduce_async(self.flat_early)

    # ---- packed-triangle collective operand ----
    def _tril_pack(self, ctx, src, extra, dst):
        _ops.tril_pack_f32(ctx, src, extra, dst)

    def _tril_unpack(self, ctx, src, dst, extra):
        _ops.tril_unpack_f32(ctx, src, dst, extra)

    def _reduce_tril_async(self, ctx, coll, mat, extra):
        """all-reduce(sum) of the lower triangle of ``mat`` [n, n] and of ``extra`` [k] through ONE packed buffer
        (n(n+1)/2 + k floats, rounded up to a multiple of 2048 so that any world size <= 8 can reduce-scatter it).
        Returns a handle whose ``wait()`` orders the current stream after the collective and unpacks in place."""
        n, k = mat.shape[0], extra.numel()
        used = _ops.tril_packed_numel(n, k)
        total = (used + 2047) // 2048 * 2048
        pk = self._buf.get("pk_early")
        if pk is None or pk.numel() != total or pk.device != mat.device:
            pk = self._buf["pk_early"] = torch.zeros(total, dtype=f32, device=mat.device)    # (the tail stays zero)
        self._tril_pack(ctx, mat, extra, pk)
        self.early_wire_numel = total
        inner = coll.all_reduce_async(pk)
        engine = self

        class _Handle:
            def wait(self_inner):
                inner.wait()
                engine._tril_unpack(ctx, pk, mat, extra)

        return _Handle()

    # ---- q(u) in natural parameters (gpytorch 1.4.0 NaturalVariationalDistribution / _NaturalToMuVarSqrt) ----
    def _natural_moments(self, ctx, nat_vec, nat_mat):
        """P = -2 theta_2 = L_P L_P^T (fp64), S = L_P^-T L_P^-1, mu = S theta_1.  Returns (S fp64 [M',M'], mu fp64 [M',1],
        device status word of the factorisation)."""
        Mp = nat_vec.shape[0]
        self._problem_size(Mp)
        nb = self.trsm_nb
        P = self._get("ngd_P", (Mp, Mp), f64)
        P.copy_(nat_mat)
        P.mul_(-2.0)
        info = self._get("ngd_info", (2,), torch.int32)
        info.zero_()
        wsP = self._bytes("ngd_wsP", _lib.lib.dsvgp_trsm_workspace_bytes(Mp, Mp, nb))
        self._potrf_and_inverse(ctx, P, info[0:1], wsP, Mp, "ngd_P")
        if nb >= Mp:
            X = wsP[:Mp * Mp * 8].view(f64).view(Mp, Mp)                               # the explicit inverse is already there
        else:                                                                          # (only its lower triangle is read below)
            eye = self._buf.get("ngd_eye")
            if eye is None or eye.shape[0] != Mp:
                eye = self._buf["ngd_eye"] = torch.eye(Mp, dtype=f64, device=self.device)
            X = self._get("ngd_X", (Mp, Mp), f64)
            _ops.trsm(ctx, P, eye, False, X, None, nb, wsP, reuse_inverse=True)        # X = L_P^-1 (lower)
        S64 = self._get("ngd_LS64", (Mp, Mp), f64)
        _ops.gemm(ctx, TRANS_A | A_UPPER | B_LOWER | OUT_LOWER, X, X, S64)              # tril(S), S = X^T X (symmetric:
        _ops.phi_symmetrize_(ctx, S64)                                                  # n^3/6 multiply-adds + a mirror pass)
        m64 = self._get("ngd_m64", (Mp, 1), f64)
        t64 = self._get("ngd_t64", (Mp, 1), f64)
        t64.copy_(nat_vec.reshape(Mp, 1))
        _ops.gemm(ctx, 0, S64, t64, m64)                                                # mu = S theta_1
        return S64, m64, info

    def _natural_to_mu_chol(self, ctx, nat_vec, nat_mat):
        """NaturalVariationalDistribution.forward: (theta_1, theta_2) -> mu and L_S = chol(S).
        Returns (mu fp32, L_S fp32, L_S fp64, trsm workspace holding the inverted blocks of L_S)."""
        Mp = nat_vec.shape[0]
        self._problem_size(Mp)          # (before reading trsm_nb: the regime of THIS factor, not of the previous problem)
        nb = self.trsm_nb
        LS64, m64, info = self._natural_moments(ctx, nat_vec, nat_mat)
        wsS = self._bytes("ngd_wsS", _lib.lib.dsvgp_trsm_workspace_bytes(Mp, Mp, nb))
        self._potrf_and_inverse(ctx, LS64, info[1:2], wsS, Mp, "ngd_LS")                # L_S (lower triangle) and L_S^-1
        bad = info.tolist()
        if bad[0] or bad[1]:
            raise NotPSDError("natural_mat does not define a positive definite precision (potrf info %s)" % bad)
        m32 = m64.reshape(Mp).to(f32)
        LS32 = torch.tril(LS64).to(f32)
        return m32, LS32, LS64, wsS

    def _natural_grads(self, ctx, grads, m32, LS64, wsS):
        """(dm, dL_S) -> gradients w.r.t. the expectation parameters eta_1 = mu, eta_2 = S + mu mu^T
        (``_NaturalToMuVarSqrt.backward``): dS through the Cholesky factor, d eta_2 = dS, d eta_1 = dm - 2 dS mu.
        In place in the slots of (dm, dL_S)."""
        Mp = m32.shape[0]
        self._problem_size(Mp)          # (the shared strategy factors an M(p+1) system in between: back to q(u)'s size)
        dm, dLS = grads["variational_mean"], grads["chol_variational_covar"]
        Lbar = self._get("Lbar", (Mp, Mp), f64)
        Lbar.copy_(dLS)
        dS = self._chol_backward(ctx, LS64, Lbar, wsS, Mp)
        dLS.copy_(dS)
        m64 = self._get("ngd_m64", (Mp, 1), f64)
        t64 = self._get("ngd_t64", (Mp, 1), f64)
        m64.copy_(m32.reshape(Mp, 1))
        _ops.gemm(ctx, 0, dS, m64, t64)
        dm.add_(t64.reshape(Mp).to(f32), alpha=-2.0)

    # ---- shared inducing directions (SharedDirectionalGradVariationalStrategy) ----
    def _shared_expand(self, params):
        """(:95-107): tile the p shared directions over the M points, interleave the M + p variational values; the
        covariance of q(u) does not reach the predictive (zero middle term, :210-212), so a unit factor stands in."""
        Z, Vs, ms = params["inducing_points"], params["inducing_directions"], params["variational_mean"]
        M, p = Z.shape[0], Vs.shape[0]
        if ms.shape[0] != M + p:
            raise ValueError("shared directions: q(u) has M + p = %d values, got %d" % (M + p, ms.shape[0]))
        idx = torch.cat([torch.arange(M, device=self.device).reshape(M, 1),
                         torch.arange(M, M + p, device=self.device).reshape(1, p).expand(M, p)], dim=1).reshape(-1)
        full = dict(params)
        full["inducing_directions"] = Vs.repeat(M, 1).contiguous()
        full["variational_mean"] = ms[idx].contiguous()
        full["chol_variational_covar"] = torch.zeros(1, 1, dtype=f32, device=self.device)      # never read
        return full, idx

    def _shared_step(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl):
        full, idx = self._shared_expand(params)
        M, p = params["inducing_points"].shape[0], params["inducing_directions"].shape[0]
        self._no_middle = True
        try:
            try:
                out = self._loss_and_grads(ctx, full, x, y, D, num_data, mll_type, global_rows, False, False, False)
            except _Refactored:
                out = self._loss_and_grads(ctx, full, x, y, D, num_data, mll_type, global_rows, False, False, True)
        finally:
            self._no_middle = False
        loss, g, mu, varn = out
        ms, LS = params["variational_mean"], params["chol_variational_covar"]
        grads = {k: g[k] for k in PARAM_NAMES if k not in ("inducing_directions", "variational_mean",
                                                            "chol_variational_covar")}
        grads["inducing_directions"] = g["inducing_directions"].reshape(M, p, -1).sum(0)
        dm = torch.zeros_like(ms)
        dm.index_add_(0, idx, g["variational_mean"])
        dLS = torch.zeros_like(LS, memory_format=torch.contiguous_format)
        if include_kl:                                             # KL of the (M + p)-dimensional q(u)
            kl_buf = torch.zeros(M + p + 1, dtype=f32, device=self.device)
            _ops.kl_terms(ctx, ms, LS, num_data, kl_buf, dm, dLS)
            loss = loss + kl_buf[0] / float(num_data)
        grads["variational_mean"], grads["chol_variational_covar"] = dm, dLS
        self.flat = None                                           # gradients are not views of one buffer here
        return loss, {k: grads[k] for k in PARAM_NAMES}, mu, varn

    # ---- CIQ whitening (CiqDirectionalGradVariationalStrategy.forward with a NaturalVariationalDistribution) ----
    def _ciq_quadrature(self, ctx, K32, v0):
        """Eigenvalue bounds from 20 Lanczos steps (device) and the elliptic-function quadrature (host, scipy) exactly
        as gpytorch's contour_integral_quad: K^-1/2 ~ sum_q omega_q (K + sigma_q I)^-1.  ``K32``: float32, or float64 for the
        fp64 model mode (the shifts and weights come back in its type)."""
        n = K32.shape[0]
        iters = min(20, n)
        if self.ciq_eig_bounds is not None:                 # (tests: the quadrature of a given spectrum interval)
            return self._ciq_quadrature_from(K32, float(self.ciq_eig_bounds[0]), float(self.ciq_eig_bounds[1]))
        alpha, beta = _ops.ciq_lanczos(ctx, K32, v0.contiguous(), iters)
        a, b = alpha.double().cpu(), beta.double().cpu()              # host sync (40 floats)
        # an invariant subspace was reached at step k (beta_k ~ 0: e.g. a start vector that is an eigenvector of an almost
        # diagonal K_ZZ, the reference's CIQ initialisation lengthscale = 1 / M): the recurrence stops there and the
        # tridiagonal is its leading (k + 1) x (k + 1) block, as in the oracle's ``lanczos_eig_bounds``
        for k in range(iters - 1):
            if not (float(b[k]) >= 1e-12 * abs(float(a[k]))):
                iters = k + 1
                a, b = a[:iters], b[:iters]
                break
        Tm = torch.diag(a)
        if iters > 1:
            Tm = Tm + torch.diag(b[:iters - 1], 1) + torch.diag(b[:iters - 1], -1)
        eigs = torch.linalg.eigvalsh(Tm)
        if not torch.isfinite(eigs).all() or eigs.min() <= 0:
            eigs = torch.diagonal(K32).double().cpu()
        lmin, lmax = float(eigs.min()), float(eigs.max())
        coll = self.collective
        if coll is not None and coll.world > 1:
            # data parallel: K_ZZ is replicated but the Lanczos start (first row of the LOCAL K_XZ shard) is not; rank 0's
            # start is the single-process one (first row of the global batch), so its bounds define the quadrature for all
            lmin, lmax = coll.broadcast_floats([lmin, lmax], self.device)
        return self._ciq_quadrature_from(K32, lmin, lmax)

    def _ciq_quadrature_from(self, K32, lmin, lmax):
        import math
        import numpy as np
        import scipy.special
        Q = int(self.ciq_num_quadrature)
        k2 = lmin / lmax
        Kp = scipy.special.ellipk(1.0 - k2)
        u = (np.arange(1, Q + 1) - 0.5) * Kp / Q
        sn, cn, dn, _ = scipy.special.ellipj(u, 1.0 - k2)
        sigma = lmin * (sn / cn) ** 2
        omega = 2.0 * Kp * math.sqrt(lmin) / (math.pi * Q) * dn / cn ** 2
        self.ciq_stats.update(lmin=lmin, lmax=lmax)
        dev = self.device
        return (torch.tensor(sigma, dtype=K32.dtype, device=dev), torch.tensor(omega, dtype=K32.dtype, device=dev),
                [float(w) for w in omega])

    def _ciq_solve(self, ctx, tag, K32, R, sigma, omega, out):
        """out = sum_q omega_q (K + sigma_q)^-1 R by basis-resident msMINRES (csrc/ciq.hip); returns the Lanczos basis, the
        per-shift coefficient table, the row norms and the iteration count.  The basis is sized for ``ciq_capacity``
        iterations; a solve that needs more is run again with twice the room (and the engine keeps the larger size)."""
        t, n = R.shape
        Q = sigma.shape[0]
        dt = K32.dtype                                      # float32, or float64 (fp64 model mode: the *_f64 entry points)
        esz = 8 if dt == f64 else 4
        names = ("ciq_basis_" + tag, "ciq_ycoef_" + tag)
        if self.ciq_capacity is None:
            self.ciq_capacity = max(20, (4 << 30) // (esz * t * n))
        while True:
            cap = min(int(self.ciq_capacity), int(self.ciq_max_iter))
            have = self._buf.get(names[0])
            if have is not None and (have.shape != torch.Size((cap + 1, t, n)) or have.dtype != dt):
                for nm in names:
                    self._buf.pop(nm, None)
                del have
                free, _ = torch.cuda.mem_get_info(self.device)
                cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
                if esz * (cap + 1) * t * n > free + cached:
                    raise RuntimeError("msMINRES did not converge within the %d Lanczos rows [%d, %d] that fit this GPU's "
                                       "free memory" % (cap, t, n))
            basis = self._get(names[0], (cap + 1, t, n), dt)
            ycoef = self._get(names[1], (t, cap, _ops.ciq_qp(Q)), dt)
            rnorm = self._get("ciq_rnorm_" + tag, (t,), dt)
            ws = self._bytes("ciq_ws", _ops.ciq_workspace_bytes(Q, t, n, cap, dt))
            its = _ops.ciq_solve(ctx, K32, R, sigma, omega, basis, ycoef, rnorm, out, ws, self.ciq_tolerance,
                                 self.ciq_max_iter)
            if its is not None:
                return basis, ycoef, rnorm, its
            self.ciq_capacity = min(2 * cap, int(self.ciq_max_iter))

    def _ciq_step(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, want_grads):
        """One step with K_ZZ^{-1/2} whitening by CIQ (reference CiqDirectionalGradVariationalStrategy.py:197-295) and the
        NGD interpolation terms (:19-123).  Everything Krylov lives in the row layout [B', M'] (csrc/ciq.hip).
        The KL term is NOT part of the returned loss (the reference's forward leaves it at zero, :74) but its gradient
        reaches (natural_vec, natural_mat) (:107,117)."""
        if self.data_outputs != "all":
            raise NotImplementedError("derivative-free data is built for the Cholesky-whitened strategy only")
        if "natural_vec" not in params:
            raise NotImplementedError("the CIQ strategy is built for a NaturalVariationalDistribution (what "
                                      "train_gp(use_ciq=True) constructs, reference directional_vi.py:164-166)")
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        Mp = M * (p + 1)
        B = x.shape[0]
        Bp = B * (p + 1)
        dev = self.device
        rows = float(Bp if global_rows is None else global_rows)
        nat_vec, nat_mat = params["natural_vec"], params["natural_mat"]
        hyp = _ops.hyp_forward(ctx, params["raw_lengthscale"], params["raw_outputscale"], params["raw_noise"])
        self.center = _ops.column_mean(ctx, Z.contiguous())
        packZ = _ops.pack_points(ctx, Z.contiguous(), V.contiguous(), p, hyp, self.center)
        packX = _ops.pack_points(ctx, x.contiguous(), D.contiguous() if p > 0 else None, p, hyp, self.center)
        K32 = self._get("ciq_K", (Mp, Mp), f32)
        _ops.kernel_fwd(ctx, packZ, M, packZ, M, d, p, hyp, jitter=self.kzz_jitter, out=K32)     # :230-234
        Rrow = self._get("ciq_R", (Bp, Mp), f32)                                                 # K_XZ = K_ZX^T, one RHS per row
        _ops.kernel_fwd(ctx, packX, B, packZ, M, d, p, hyp, out=Rrow)
        sigma, omega, omega_host = self._ciq_quadrature(ctx, K32, Rrow[0])
        Q = sigma.shape[0]
        Trow = self._get("ciq_T", (Bp, Mp), f32)
        basisF, ycoefF, rnF, its = self._ciq_solve(ctx, "f", K32, Rrow, sigma, omega, Trow)        # :255-256
        self.ciq_stats.update(iterations=its)
        # natural parameters -> S, m (the reference's preconditioned CG on the precision, :51-61, as a direct fp64 solve)
        S64, m64, info = self._natural_moments(ctx, nat_vec, nat_mat)
        if int(info[0].item()) != 0:
            raise NotPSDError("natural_mat does not define a positive definite precision")
        S32 = self._get("ciq_S32", (Mp, Mp), f32)
        S32.copy_(S64)
        m32 = m64.reshape(Mp).to(f32)
        STrow = self._get("ciq_ST", (Bp, Mp), f32)
        _ops.gemm(ctx, 0, Trow, S32, STrow)                                                      # (S T)^T = T^T S
        imean, mu, var, live = _ops.ciq_rowstats(ctx, Trow, STrow, p, m32, params["constant"].reshape(-1), hyp,
                                                 self.ciq_kxx_jitter)                              # :65-69,265-266
        mu_bar = torch.empty(Bp, dtype=f32, device=dev)
        var_bar = torch.empty(Bp, dtype=f32, device=dev)
        varn = torch.empty(Bp, dtype=f32, device=dev)
        scal = torch.empty(8, dtype=f32, device=dev)
        if not want_grads:
            noise = hyp[2]
            return None, None, mu, (var + noise).clamp_min_(1e-6)
        _ops.likelihood_terms(ctx, mu, var, y.contiguous(), p, hyp, 0 if mll_type == "ELBO" else 1, rows, mu_bar, var_bar,
                              varn, scal)
        grads, loss_out, d_hyp = self._alloc_grads(params, NGD_PARAM_NAMES)
        Tbar = self._get("ciq_Tbar", (Bp, Mp), f32)
        VT = self._get("ciq_VT", (Bp, Mp), f32)
        cvec = _ops.ciq_tbar(ctx, Trow, STrow, m32, mu_bar, var_bar, live, imean, Tbar, VT)       # :94-96
        kl_bar = (1.0 / float(num_data)) if include_kl else 0.0
        d1 = grads["natural_vec"].reshape(1, Mp)
        _ops.gemm(ctx, 0, cvec.reshape(1, Bp), Trow, d1)                                          # :102-106
        d1.add_(nat_vec.reshape(1, Mp), alpha=kl_bar)                                             # :107
        d2 = grads["natural_mat"]
        _ops.gemm(ctx, TRANS_A | OUT_LOWER, VT, Trow, d2)                                         # :115-116: T^T diag(vbar) T is
        _ops.mirror_lower_f32_(ctx, d2, Mp)                                                       # symmetric: lower triangle + mirror
        d2.add_(nat_mat, alpha=kl_bar)                                                            # kl/2 (I - prec), prec = -2 theta_2
        d2.diagonal().add_(0.5 * kl_bar)
        # backward of sqrt_inv_matmul: dR = K^-1/2 Tbar, dK = -sym sum_q omega_q Y_q^T X_q (same quadrature)
        Rbar = self._get("ciq_Rbar", (Bp, Mp), f32)
        basisB, ycoefB, rnB, its_b = self._ciq_solve(ctx, "b", K32, Tbar, sigma, omega, Rbar)
        self.ciq_stats.update(iterations_backward=its_b)
        # the solves X_q = rnF mix(basisF, ycoefF)_q, Y_q = rnB mix(basisB, ycoefB)_q are never formed: with per-row cross
        # coefficients the sum over shifts is ONE product over the stacked rows of the shorter basis (or, when both solves took
        # more iterations than there are shifts, over the Q materialised solves)
        dK = self._get("ciq_dK", (Mp, Mp), f32)
        form = self.ciq_backward_form
        if form is None:
            form = "backward" if its_b <= min(its, Q) else ("forward" if its <= Q else "shifts")
        kmin = {"backward": its_b, "forward": its, "shifts": Q}[form]
        if form == "backward":
            ctab = _ops.ciq_cross(ctx, ycoefB, its_b, ycoefF, its, omega, rnB, rnF)
            Zs = _ops.ciq_mix(ctx, basisF, its, ctab, its_b, None, self._get("ciq_Z", (its_b, Bp, Mp), f32))
            left, right = basisB[:its_b], Zs
        elif form == "forward":
            ctab = _ops.ciq_cross(ctx, ycoefF, its, ycoefB, its_b, omega, rnF, rnB)
            Zs = _ops.ciq_mix(ctx, basisB, its_b, ctab, its, None, self._get("ciq_Z", (its, Bp, Mp), f32))
            left, right = Zs, basisF[:its]
        else:
            om = torch.zeros(_ops.ciq_qp(Q), dtype=f32, device=dev)
            om[:Q] = omega
            right = _ops.ciq_mix(ctx, basisF, its, ycoefF * om, Q, rnF, self._get("ciq_Z", (Q, Bp, Mp), f32))
            left = _ops.ciq_mix(ctx, basisB, its_b, ycoefB, Q, rnB, self._get("ciq_Z2", (Q, Bp, Mp), f32))
        ev = self._event_pair()                           # (bench.py --config c5: the largest single launch of the CIQ step)
        _ops.gemm(ctx, TRANS_A, left.reshape(kmin * Bp, Mp), right.reshape(kmin * Bp, Mp), dK, alpha=-1.0)
        self._event_done("ciq_stacked_backward", ev)
        self.ciq_stats.update(stacked_depth=int(kmin * Bp))
        Kzzbar = self._get("ciq_Kzzbar", (Mp, Mp), f32)
        _ops.sym_average_f32(ctx, dK, Kzzbar)
        Kb32 = self._get("Kb32", (Mp, Bp), f32)
        _ops.transpose_f32(ctx, Rbar, Kb32)
        dZ, dV = grads["inducing_points"], grads["inducing_directions"]
        kws = self._bytes("kbwd_ws", max(_lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, B, d, p),
                                         _lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, M, d, p)))
        _ops.kernel_bwd(ctx, Kb32, packZ, M, packX, B, d, p, hyp, False, dZ, dV, d_hyp, kws)
        _ops.kernel_bwd(ctx, Kzzbar, packZ, M, packZ, M, d, p, hyp, True, dZ, dV, d_hyp, kws)
        kl0 = torch.zeros(1, dtype=f32, device=dev)                                               # :74
        _ops.step_epilogue(ctx, scal, kl0, rows, num_data, params["raw_lengthscale"].reshape(-1),
                           params["raw_outputscale"].reshape(-1), params["raw_noise"].reshape(-1), d_hyp,
                           grads["raw_lengthscale"].reshape(-1), grads["raw_outputscale"].reshape(-1),
                           grads["raw_noise"].reshape(-1), grads["constant"].reshape(-1), loss_out)
        return loss_out[0], grads, mu, varn

    def _chol_backward(self, ctx, L, Lbar, ws, Mp, phi_arg=False):
        """K-bar = 1/2 L^-T (Phi(L^T L-bar) + Phi(.)^T) L^-1 (symmetric, fp64) for the lower factor L whose inverted
        blocks are in ``ws``; L-bar (lower) is destroyed.  ``phi_arg``: ``Lbar`` holds tril(L^T L-bar) already (the ELBO fast
        path forms it as -tril([S - I | m / (2 vbar)][G ; b^T]) without L-bar, see ``_elbo_fast``)."""
        G1 = self._get("G1", (Mp, Mp), f64)
        if phi_arg:
            G1.copy_(Lbar)
        else:
            _ops.gemm(ctx, TRANS_A | A_UPPER | B_LOWER | OUT_LOWER, L, Lbar, G1)   # tril(L^T L-bar): Phi reads nothing else
        _ops.phi_symmetrize_(ctx, G1)                                       # Phi(.) + Phi(.)^T (mirror of the lower part)
        Kbar = G1                                                           # reuse (after the first product has read it)
        if self.trsm_nb >= Mp:
            # explicit inverse in the workspace: Yt = S L^-1 directly (S symmetric, read through its transpose: the
            # mn-contiguous operand path), then only the lower half of the symmetric result, mirrored
            Linv = ws[:Mp * Mp * 8].view(f64).view(Mp, Mp)
            Yt = Lbar                                                       # reuse
            # only the lower triangle of S L^-1 is read by the next product (rows k >= i >= j of column j): n^3/3 instead of n^3/2
            _ops.gemm(ctx, TRANS_A | B_LOWER | OUT_LOWER, G1, Linv, Yt)     # tril(S L^-1), S L^-1 = (L^-T S)^T
            _ops.gemm(ctx, TRANS_A | A_UPPER | OUT_LOWER, Linv, Yt, Kbar, alpha=0.5)        # 1/2 tril(L^-T S L^-1)
            _ops.phi_symmetrize_(ctx, Kbar)
        else:
            Y = self._get("Y", (Mp, Mp), f64)
            _ops.trsm(ctx, L, G1, True, Y, None, self.trsm_nb, ws, reuse_inverse=True)      # L^-T S
            Yt = Lbar                                                       # reuse
            _ops.transpose_f64(ctx, Y, Yt)
            _ops.trsm(ctx, L, Yt, True, Kbar, None, self.trsm_nb, ws, reuse_inverse=True)   # L^-T S L^-1 (symmetric)
            Kbar.mul_(0.5)
        return Kbar

    def _chol_backward_cols(self, ctx, L, Lbar, ws, Mp, c0, c1, phi_arg=False):
        """Columns [c0, c1) of K-bar = 1/2 L^-T (Phi(L^T L-bar) + Phi(.)^T) L^-1 from the explicit inverse in ``ws``
        (one rank's share of the Cholesky backward under the global-Gram schedule): 2 M'^2 w flops instead of 3 M'^3.
        ``phi_arg``: ``Lbar`` (float32 or float64) holds tril(L^T L-bar) already."""
        G1 = self._get("G1", (Mp, Mp), f64)
        if phi_arg:
            G1.copy_(Lbar)
        else:
            _ops.gemm(ctx, TRANS_A | A_UPPER | B_LOWER | OUT_LOWER, L, Lbar, G1)   # tril(L^T L-bar)
        _ops.phi_symmetrize_(ctx, G1)                                       # S = Phi(.) + Phi(.)^T
        Linv = ws[:Mp * Mp * 8].view(f64).view(Mp, Mp)
        w = c1 - c0
        T = self._get("cbT", (Mp, w), f64)
        # S L^-1[:, c0:c1]: rows < c0 of that column block of the lower-triangular inverse are zero, the rest is lower
        # triangular in its own coordinates (nothing above the diagonal of the workspace is read)
        _ops.gemm(ctx, TRANS_A | B_LOWER, G1[c0:, :], Linv[c0:, c0:c1], T)       # (S symmetric: read through its transpose)
        Kc = self._get("cbK", (Mp, w), f64)
        _ops.gemm(ctx, TRANS_A | A_UPPER, Linv, T, Kc, alpha=0.5)
        return Kc

    def _c_step_eligible(self, params, x, use_fast, sync):
        if not (self.c_step and use_fast and not sync and not self.capture_mode and self.whitening == "cholesky"
                and self.data_outputs == "all" and not self.shared_directions and not self._no_middle and self.potrf_algo == 1
                and self.fused_inverse and self._trsm_nb is None and not self.lib_dense_gemm):
            return False
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        Mp = M * (p + 1)
        coll = self.collective
        world = coll.world if coll is not None else 1
        if world > 1:
            # a data-parallel rank: dsvgp_elbo_step_dp_f32 covers the global-Gram schedule with the sharded replicated stage and
            # the packed early operand; every other multi-rank schedule keeps the piecewise path
            if not (self.global_gram and self._allow_early and self.shard_replicated and self.pack_reduce and M >= world
                    and hasattr(coll, "all_gather_async") and Mp >= self.shard_min_mp and Mp >= 4 * world
                    and not self.deterministic):
                return False
        rows_local = x.shape[0]
        if world > 1:
            # the choice of path must be the SAME on every rank (the two paths issue differently sized collectives): decide it
            # from global quantities only -- the largest shard of the global minibatch (a shape the smaller shards then fit too),
            # never from this rank's own row count
            gb = getattr(coll, "global_batch", None)
            if gb is not None:
                if gb < world:
                    return False
                rows_local = -(-gb // world)
        return M > 0 and rows_local > 0 and Mp <= 8192 and _ops.step_supported(M, d, p, rows_local, world)

    def _c_step_po_eligible(self, params, x, sync):
        """the per-output step (PLL objective, or ELBO with the per-output variances wanted) as ONE C call (dsvgp_elbo_step_po_f32):
        one rank, Cholesky whitening, every data point with its derivatives, explicit-inverse regime"""
        if not (self.c_step and not sync and not self.capture_mode and self.whitening == "cholesky" and self.data_outputs == "all"
                and not self.shared_directions and not self._no_middle and self.potrf_algo == 1 and self.fused_inverse
                and self._trsm_nb is None and not self.lib_dense_gemm and not self.deterministic):
            return False
        coll = self.collective
        if coll is not None and coll.world > 1:
            return False
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        return M > 0 and x.shape[0] > 0 and M * (p + 1) <= 8192 and _ops.step_supported(M, d, p, x.shape[0], 1, per_output=True)

    def _c_step(self, ctx, params, x, y, D, num_data, rows, include_kl, per_output=None):
        """the whole fast-path step queued by dsvgp_elbo_step_f32 (one ctypes call); raises _Refactored when the
        factorisation failed (the caller then runs the jitter ladder on the piecewise path).
        ``per_output``: None = the ELBO fast path; "ELBO" / "PLL" = the per-output step (dsvgp_elbo_step_po_f32), which also
        returns the per-output variances"""
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M
        B = x.shape[0]
        Mp, Bp = M * (p + 1), B * (p + 1)
        self._problem_size(Mp)
        coll = self.collective
        world = coll.world if coll is not None else 1
        po = per_output is not None
        pkey = (M, d, p, B, world, po)
        plan = self._plans.get(pkey)
        if plan is None:
            plan = self._plans[pkey] = _ops.StepPlan(ctx, M, d, p, B, world, per_output=po)
        # one workspace PER plan: a plan clears the pad columns of its fp32 [Q' | a] once per workspace and assumes nobody else
        # writes there (a ragged tail batch has its own plan, layout and buffer)
        ws = self._bytes("cstep_ws_%d_%d_%d_%d_%d%s" % (M, d, p, B, world, "_po" if po else ""), plan.bytes)
        grads, loss_out, d_hyp = self._alloc_grads(params, PARAM_NAMES, zero=False)
        mu = torch.empty(Bp, dtype=f32, device=self.device)
        LS, dLS = params["chol_variational_covar"], grads["chol_variational_covar"]
        for name, t in (("inducing_points", Z), ("inducing_directions", V), ("x", x), ("y", y), ("D", D),
                        ("variational_mean", params["variational_mean"]), ("chol_variational_covar", LS)):
            if t.numel() and (t.dtype != f32 or not t.is_cuda or (t.dim() == 2 and t.stride(1) != 1) or (t.dim() != 2 and not t.is_contiguous())):
                raise _lib.DsvgpError("%s must be a float32 GPU tensor with unit inner stride" % name)
        if not (Z.is_contiguous() and x.is_contiguous() and (p == 0 or (V.is_contiguous() and D.is_contiguous()))):
            raise ValueError("points and directions must be contiguous")
        for name in ("constant", "raw_lengthscale", "raw_outputscale", "raw_noise"):      # (handed over as bare device pointers)
            t = params[name]
            if t.dtype != f32 or not t.is_cuda or t.numel() != 1 or not t.is_contiguous():
                raise _lib.DsvgpError("%s must be a contiguous float32 GPU tensor with one element" % name)
        P = lambda t: t.data_ptr() if t is not None and t.numel() else None
        io = plan.io
        io.Z, io.V, io.m, io.LS, io.ldls = P(Z), P(V), P(params["variational_mean"]), P(LS), _ops._ld(LS)
        io.constant, io.raw_lengthscale = P(params["constant"]), P(params["raw_lengthscale"])
        io.raw_outputscale, io.raw_noise = P(params["raw_outputscale"]), P(params["raw_noise"])
        io.x, io.y, io.D = P(x), P(y), P(D)
        # (one-hot shared directions stated by the caller: the canonical assembly kernels; the same statement on the inducing directions --
        #  the full-gradient SVGP -- the both-sides kernels)
        st = _ops.stated_directions(D, d, p, lambda d_, p_: _ops.canon_supported(d_, p_) or _ops.canon2_supported(d_, p_))
        io.dir_idx, io.dir_idx_base = (st[0].data_ptr(), st[1]) if st is not None else (None, 0)
        io.v_one_hot = 1 if (st is not None and _ops.canon2_supported(d, p)
                             and _ops.same_statement(st, _ops.stated_directions(V, d, p, _ops.canon2_supported))) else 0
        full = self._flat_full
        io.flat, io.flat_floats = full.data_ptr(), full.numel()
        io.dZ, io.dV, io.dm = P(grads["inducing_points"]), P(grads["inducing_directions"]), P(grads["variational_mean"])
        io.dLS, io.lddls = P(dLS), _ops._ld(dLS)
        io.d_hyp, io.d_constant = P(d_hyp), P(grads["constant"])
        io.d_raw_lengthscale, io.d_raw_outputscale = P(grads["raw_lengthscale"]), P(grads["raw_outputscale"])
        io.d_raw_noise, io.loss, io.mu = P(grads["raw_noise"]), P(loss_out), P(mu)
        io.num_data, io.global_rows, io.kzz_jitter = float(num_data), float(rows), float(self.kzz_jitter)
        # the second stream from M' = 512 up: queued from C (two event records + two waits per fork / join) it pays at C2 already
        # (M' = 600: 0.632 -> 0.615 ms/step); the Python-orchestrated path keeps its M' >= 2048 rule
        overlap = self.overlap if self.overlap is not None else Mp >= 512
        timed = self.record_events and self._rec_count % max(1, self.record_every) == 0
        if self.record_events:
            self._rec_count += 1
        flags = (1 if overlap and not self.deterministic else 0) | (2 if include_kl else 0) | (4 if timed else 0) \
            | (16 if self.tail_side else 0) | (64 if self.phi_arg_fp64 else 0) | (128 if self.solve_pipe else 0)
        io.split_ws, io.split_ws_bytes = None, 0
        if self.split_bf16 and Mp >= 256 and Bp >= 256:
            sws = self._bytes("cstep_split_ws", int(_lib.lib.dsvgp_elbo_step_split_bytes(M, d, p, B)))
            io.split_ws, io.split_ws_bytes = sws.data_ptr(), sws.numel()
            flags |= 32
        tr = self.host_trace                  # (tools/host_trace.py: where the host's time goes; None in production)
        if tr is not None:
            import time as _t
            t0 = _t.perf_counter()
        varn = None
        if po:
            varn = torch.empty(Bp, dtype=f32, device=self.device)
            plan.run_po(ctx, ws, (flags & 3) | (256 if per_output == "PLL" else 0), varn)
            timed = False
        elif world > 1:
            self._c_step_dp_phases(ctx, plan, ws, flags, coll, Mp)
        else:
            plan.run(ctx, ws, flags)
        timed_idx = plan.timed_count() - 1 if timed else None
        if tr is not None:
            t1 = _t.perf_counter()
        if self.defer_status and world == 1 and not timed:
            self._deferred = (plan, ws)          # (status unread: deferred_check / deferred_guard)
            if tr is not None:
                tr.append((t0, t1, t1))
        else:
            info, hyp = plan.status()            # waits for the factorisation only; the rest of the step stays queued
            if tr is not None:
                tr.append((t0, t1, _t.perf_counter()))
            self._hyp_host = hyp[:3]
            if info != 0:
                raise _Refactored()
        if timed:
            self.c_step_timed.append((plan, timed_idx))  # bench.py reads the plan's HIP events after its timed region (a step whose
                                                         # factorisation failed is not listed: its events time garbage)
        self.c_step_used = True
        self._pending = None
        if po:
            self._last_fast = None
            return loss_out[0], grads, mu, varn
        self._last_fast = ("plan", plan, ws, p)
        return loss_out[0], grads, mu, torch.empty(0, dtype=f32, device=self.device)

    def deferred_guard(self):
        """int32 device tensor [1]: the status word of the deferred step (0 = the factorisation went through); None when nothing is deferred"""
        if self._deferred is None:
            return None
        plan, ws = self._deferred
        return plan.locate(ws, 5).reshape(-1)

    def deferred_check(self):
        """reads the status of the deferred step (waits for its factorisation only); returns the status word, 0 when nothing was deferred"""
        if self._deferred is None:
            return 0
        plan, _ = self._deferred
        self._deferred = None
        info, hyp = plan.status()
        self._hyp_host = hyp[:3]
        return info

    def _c_step_dp_phases(self, ctx, plan, ws, flags, coll, Mp):
        """one data-parallel rank: the five pieces of dsvgp_elbo_step_dp_f32 with this rank's collectives between them
        (include/dsvgp.h; same schedule as the piecewise global-Gram path with the sharded replicated stage below)"""
        world, dev = coll.world, self.device
        wq = ((Mp + 1 + world - 1) // world + 3) // 4 * 4
        wr = ((Mp + world - 1) // world + 1) // 2 * 2
        used = _ops.tril_packed_numel(Mp, Mp)
        total = (used + 2047) // 2048 * 2048                # (any world size <= 8 can reduce-scatter it)
        key = ("dp_bufs", Mp, world)
        bufs = self._buf.get(key)
        if bufs is None:
            bufs = self._buf[key] = dict(wire=torch.zeros(total, dtype=f32, device=dev),
                                         q_local=torch.zeros(Mp, wq, dtype=f32, device=dev),
                                         q_all=torch.empty(world, Mp, wq, dtype=f32, device=dev),
                                         lbar_local=torch.zeros(wr, Mp, dtype=f32, device=dev),
                                         lbar_all=torch.empty(world * wr, Mp, dtype=f32, device=dev))
        dp = plan.dp
        dp.rank, dp.world = coll.rank, world
        dp.wire, dp.wire_floats = bufs["wire"].data_ptr(), total
        dp.q_local, dp.q_all = bufs["q_local"].data_ptr(), bufs["q_all"].data_ptr()
        dp.lbar_local, dp.lbar_all = bufs["lbar_local"].data_ptr(), bufs["lbar_all"].data_ptr()
        self.early_wire_numel = total
        # dp_host_trace (bench.py --gpus N / tools/host_trace.py; None in production): host time stamps around the five C calls and
        # the collectives issued / waited for between them -- what the HOST spends per data-parallel rank step, phase by phase
        tr = self.dp_host_trace
        now = time.perf_counter if tr is not None else (lambda: 0.0)
        t = [now()]
        plan.run_dp(ctx, ws, flags, 0); t.append(now())
        h_g = coll.all_reduce_async(bufs["wire"]); t.append(now())
        plan.run_dp(ctx, ws, flags, 1); t.append(now())
        h_q = coll.all_gather_async(bufs["q_all"], bufs["q_local"]); t.append(now())
        ev_w = self._event_pair()                        # (bench.py: how long the main stream stalls for [G ; b^T])
        h_g.wait()
        self._event_done("early_reduce_wait", ev_w); t.append(now())
        plan.run_dp(ctx, ws, flags, 2); t.append(now())
        h_l = coll.all_gather_async(bufs["lbar_all"], bufs["lbar_local"]); t.append(now())
        h_q.wait(); t.append(now())
        plan.run_dp(ctx, ws, flags, 3); t.append(now())
        h_l.wait(); t.append(now())
        plan.run_dp(ctx, ws, flags, 4); t.append(now())
        if tr is not None:
            tr.append(t)
        self._global_gram = True
        self.sharded_stage_used = True
        self.variational_grads_global = True

    def _loss_and_grads(self, ctx, params, x, y, D, num_data, mll_type, global_rows, include_kl, fast, sync):
        use_fast = mll_type == "ELBO" and fast
        self._ctx = ctx
        self.c_step_used = False
        self._last_fast = None
        if self._c_step_eligible(params, x, use_fast, sync):
            pz = params["inducing_directions"].shape[0] // params["inducing_points"].shape[0]
            Bq = x.shape[0] * (pz + 1)
            if y.shape != (Bq,):
                raise ValueError("y must be the interleaved target vector of length B*(p+1)=%d" % Bq)
            return self._c_step(ctx, params, x, y.contiguous(), D, num_data, float(Bq if global_rows is None else global_rows),
                                include_kl)
        if not use_fast and mll_type in ("ELBO", "PLL") and self._c_step_po_eligible(params, x, sync):
            pz = params["inducing_directions"].shape[0] // params["inducing_points"].shape[0]
            Bq = x.shape[0] * (pz + 1)
            if y.shape != (Bq,):
                raise ValueError("y must be the interleaved target vector of length B*(p+1)=%d" % Bq)
            return self._c_step(ctx, params, x, y.contiguous(), D, num_data, float(Bq if global_rows is None else global_rows),
                                include_kl, per_output=mll_type)
        Mz = params["inducing_points"].shape[0]
        p = params["inducing_directions"].shape[0] // Mz if Mz else 0
        B = x.shape[0]
        pd = self._pd(p)
        Bp = B * (pd + 1)
        if y.shape != (Bp,):
            raise ValueError("y must be the interleaved target vector of length B*(p+1)=%d" % Bp)
        rows = float(Bp if global_rows is None else global_rows)
        side = {}
        side_job = None
        overlap = self.overlap if self.overlap is not None else Mz * (p + 1) >= 2048
        if self.deterministic:
            overlap = False                                 # one stream: the split-K scratch serves one stream at a time
        if use_fast and overlap:
            def side_job(c, hyp_):
                side.update(self._fast_prologue(c, params, hyp_, x, D, rows, background=self.side_background))
        if self.capture_mode and not use_fast:
            raise RuntimeError("only the ELBO fast path can be captured into a HIP graph")
        hyp, packZ, L, dims = self._factor(ctx, params, sync=(sync or not use_fast) and not self.capture_mode,
                                           side_job=side_job, nrhs=Bp)
        M, d, p, Mp = dims
        m = params["variational_mean"]
        LS = params["chol_variational_covar"]
        dev = self.device
        grads, loss_out, d_hyp = self._alloc_grads(params, PARAM_NAMES)
        dLS, dm = grads["chol_variational_covar"], grads["variational_mean"]
        scal = torch.empty(8, dtype=f32, device=dev)
        kl_buf = torch.zeros(2 * Mp + 1, dtype=f32, device=dev)      # [KL | per-row KL partials | per-row trace partials]
        Lbar = self._get("Lbar", (Mp, Mp), f64)
        Kb32 = self._get("Kb32", (Mp, Bp), f32)
        y = y.contiguous()

        coll = self.collective
        self._global_gram = bool(use_fast and self.global_gram and self._allow_early and coll is not None
                                 and coll.world > 1 and self.trsm_nb >= Mp and M >= coll.world)
        self._dev_scale = False
        if use_fast:
            if not side:        # no overlap: the prologue runs in line
                side.update(self._fast_prologue(ctx, params, hyp, x, D, rows))
            else:
                torch.cuda.current_stream(self.device).wait_event(self._side_done)
            packX, mu = self._elbo_fast(ctx, params, hyp, packZ, L, dims, x, y, D, rows, num_data, include_kl,
                                        scal, kl_buf, dm, dLS, Kb32, Lbar, side)
            varn = torch.empty(0, dtype=f32, device=dev)
        else:
            packX, A64, A32, W, mu, var = self._interp(ctx, params, hyp, packZ, L, dims, x, D)
            mu_bar = torch.empty(Bp, dtype=f32, device=dev)
            var_bar = torch.empty(Bp, dtype=f32, device=dev)
            varn = torch.empty(Bp, dtype=f32, device=dev)
            _ops.likelihood_terms(ctx, mu, var, y, pd, hyp, 0 if mll_type == "ELBO" else 1, rows, mu_bar, var_bar,
                                  varn, scal)
            # ---- variational parameters ----
            Abar = self._get("Abar", (Mp, Bp), f32)
            if self._no_middle:
                _ops.abar(ctx, A32, A32, m, mu_bar, var_bar, Abar)              # m mu_bar^T (U == A: no variance path)
            else:
                U = self._get("U", (Mp, Bp), f32)
                _ops.gemm(ctx, A_LOWER, LS, W, U)                               # U = L_S W
                _ops.abar(ctx, A32, U, m, mu_bar, var_bar, Abar)                # m mu_bar^T + 2 (U - A) diag(var_bar)
                _ops.gemm(ctx, TRANS_B | OUT_LOWER, A32, W, dLS, alpha=2.0, kscale=var_bar)   # tril(2 A diag(vbar) W^T)
            _ops.rowdot_accum(ctx, A32, mu_bar, dm)                             # A mu_bar
            if include_kl:
                _ops.kl_terms(ctx, m, LS, num_data, kl_buf, dm, dLS)
            self._variational_grads_final()
            # ---- through the triangular solve (fp64) ----
            Kb64 = self._get("Kb64", (Mp, Bp), f64)
            ws = self._buf["trsm_ws"]
            _ops.trsm(ctx, L, Abar, True, Kb64, Kb32, self.trsm_nb, ws, reuse_inverse=True)   # K_ZX-bar = L^-T Abar
            _ops.gemm(ctx, TRANS_B | OUT_LOWER, Kb64, A64, Lbar, alpha=-1.0)    # L-bar = -tril(K_ZX-bar A^T)

        dZ, dV = grads["inducing_points"], grads["inducing_directions"]
        kws = self._bytes("kbwd_ws", max(_lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, B, d, p),
                                         _lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, M, d, p)))
        # K_ZX-bar's kernel backward (HBM-bound read of 4 M' B' bytes) next to the fp64 products of L-bar and the Cholesky backward
        # (matrix-pipe bound): on the side stream from the moment the dense product is done, joined before the K_ZZ kernel backward
        # (both accumulate into Z-bar, V-bar and the hyper-parameter slots)
        zx_done = None
        dense_done, self._dense_done = getattr(self, "_dense_done", None), None
        bwd_overlap = self.bwd_overlap if self.bwd_overlap is not None else (Bp <= 2 * Mp and self.collective is None)
        bwd_overlap = bwd_overlap and not self.deterministic
        if bwd_overlap and use_fast and dense_done is not None and self._side is not None and not self.capture_mode:
            kws2 = self._bytes("kbwd_ws_zx", _lib.lib.dsvgp_kernel_bwd_workspace_bytes(M, B, d, p))
            with torch.cuda.stream(self._side):
                self._side.wait_event(dense_done)
                ctx.bind()
                self._kernel_bwd_zx(ctx, Kb32, packZ, M, packX, B, d, p, hyp, dZ, dV, d_hyp, kws2)
                zx_done = torch.cuda.Event()
                zx_done.record(self._side)
            ctx.bind()
        else:
            self._kernel_bwd_zx(ctx, Kb32, packZ, M, packX, B, d, p, hyp, dZ, dV, d_hyp, kws)     # data side: no gradient
        if self._global_gram:
            # L-bar is the same on every rank: each takes the inducing points [m0, m1) = columns [m0 q, m1 q) of the
            # symmetric K_ZZ-bar, whose contributions sum in the final all-reduce of (Z-bar, V-bar, hyper-parameters)
            q = p + 1
            base, rem = divmod(M, coll.world)
            m0 = coll.rank * base + min(coll.rank, rem)
            m1 = m0 + base + (1 if coll.rank < rem else 0)
            Lb = self._lbar_f32 if getattr(self, "_lbar_f32", None) is not None else Lbar
            Kcols = self._chol_backward_cols(ctx, L, Lb, self._buf["trsm_ws"], Mp, m0 * q, m1 * q, phi_arg=use_fast)
            if zx_done is not None:
                torch.cuda.current_stream(self.device).wait_event(zx_done)
            sub = (packZ[0][m0 * q:m1 * q], packZ[1][m0 * q:m1 * q])
            _ops.kernel_bwd(ctx, Kcols, packZ, M, sub, m1 - m0, d, p, hyp, True, dZ, dV, d_hyp, kws)
            self.variational_grads_global = True
        else:
            # ---- Cholesky backward (fp64): K_ZZ-bar = 1/2 L^-T (Phi(L^T L-bar) + Phi(.)^T) L^-1, symmetric kernel backward ----
            Kzzbar = self._chol_backward(ctx, L, Lbar, self._buf["trsm_ws"], Mp, phi_arg=use_fast)
            if zx_done is not None:
                torch.cuda.current_stream(self.device).wait_event(zx_done)
            _ops.kernel_bwd(ctx, Kzzbar, packZ, M, packZ, M, d, p, hyp, True, dZ, dV, d_hyp, kws)

        if self._dev_scale:
            # the kernel gradients above came from the UNSCALED K_ZX-bar / K_ZZ-bar: apply 2 vbar = 1 / (noise rows) now
            _ops.scale_by_vbar_(ctx, [dZ, dV if dV.numel() else None, d_hyp[0:2]], hyp, rows)
        # ---- scalars: d_hyp += data-term scalars, softplus chain rule, d constant, loss (one launch) ----
        _ops.step_epilogue(ctx, scal, kl_buf, rows, num_data, params["raw_lengthscale"].reshape(-1),
                           params["raw_outputscale"].reshape(-1), params["raw_noise"].reshape(-1), d_hyp,
                           grads["raw_lengthscale"].reshape(-1), grads["raw_outputscale"].reshape(-1),
                           grads["raw_noise"].reshape(-1), grads["constant"].reshape(-1), loss_out)
        loss = loss_out[0]
        return loss, grads, mu, varn

    def _fast_prologue(self, ctx, params, hyp, x, D, rows, background=False):
        """The part of the ELBO fast path that does not depend on L: K_ZX assembly and [S - I | m / (2 vbar)]."""
        Z, V = params["inducing_points"], params["inducing_directions"]
        M, d = Z.shape
        p = V.shape[0] // M if M else 0
        Mp = M * (p + 1)
        B = x.shape[0]
        m = params["variational_mean"]
        LS = params["chol_variational_covar"]
        packZ = self._pending_packZ
        packX = _ops.pack_points(ctx, x.contiguous(), D.contiguous() if p > 0 else None, p, hyp, self.center)
        self._zx_dirs = _ops.stated_directions(D, d, p) if p > 0 else None
        Kzx = self._assemble_kzx(ctx, packZ, M, packX, B, d, p, hyp, Mp)
        # [S - I | m / (2 vbar)]: one solve gives [Q' | a / (2 vbar)].  Rows padded to a multiple of 4 floats: the lean fp64
        # kernel (gemm64.hip) streams a float right-hand side with 16-byte loads
        S32e = self._get("S32e_pad", (Mp, (Mp + 1 + 3) // 4 * 4), f32)[:, :Mp + 1]
        S32 = S32e[:, :Mp]
        # S = tril(L_S) tril(L_S)^T.  On the side stream it is launched as a one-workgroup-per-CU filler: at full occupancy its
        # 4600 workgroups leave no CU with the LDS share a Cholesky step workgroup needs, and two launches of the chain wait
        # 130-190 us each for it to drain
        # (S is symmetric: only its lower triangle is formed -- n^3/6 instead of n^3/3 multiply-adds, half the time this filler
        # spends next to the chain -- and mirrored)
        _ops.gemm(ctx, A_LOWER | TRANS_B | _lib.B_UPPER | OUT_LOWER | (_lib.BACKGROUND if background else 0), LS, LS, S32)
        _ops.mirror_lower_f32_(ctx, S32, Mp)
        _ops.sminus_i_col_(ctx, S32e, Mp, m.contiguous(), hyp, rows)   # S - I and the column m / (2 vbar), 2 vbar = 1 / (noise rows)
        # fp64 copy: the left operand of tril(L^T L-bar) = -2 vbar tril([S - I | m / (2 vbar)][G ; b^T]) (see _elbo_fast)
        S64e = self._get("S64e", (Mp, (Mp + 2) // 2 * 2), f64)[:, :Mp + 1]
        _ops.widen_f32_f64(ctx, S32e, S64e)
        return dict(packX=packX, Kzx=Kzx, S32e=S32e, S64e=S64e)

    def _elbo_fast(self, ctx, params, hyp, packZ, L, dims, x, y, D, rows, num_data, include_kl, scal, kl_buf, dm,
                   dLS, Kb32, Lbar, pro):
        """ELBO mode: dLoss/dvar_j = vbar = 1/(2 noise rows) for every output, hence
             sum_j var_j = prior + tr(L_S^T G L_S) - tr(G),             G = A A^T          (M' x M')
             L_S-bar     = 2 vbar tril(G L_S)
             K_ZX-bar    = L^-T (m mu_bar^T + 2 vbar (S - I) A) = 2 vbar [Q' | a/(2 vbar)] [A ; mu_bar^T]
             L-bar       = -tril(K_ZX-bar A^T)                  = -2 vbar tril([Q' | a/(2 vbar)] [G ; b^T])
           with Q' = L^-T (S - I), a = L^-T m (fp64 solves on M' x M' data), b = A mu_bar.
           L-bar itself is never formed (round 4): the Cholesky backward reads only tril(L^T L-bar), where the tril() of L-bar
           does not matter (row i of the upper-triangular L^T meets rows k >= i of L-bar), and L^T Q' = S - I, L^T a = m:
             tril(L^T L-bar) = -2 vbar tril([S - I | m/(2 vbar)] [G ; b^T])       (one product, fp64 accumulation)
           Three of the six [M', B'] products of the general path (W, U, the fp64 backward solve and the fp64
           L-bar contraction) are replaced by one fp32 Gram product and one fp32 dense product."""
        M, d, p, Mp = dims
        B = x.shape[0]
        pd = self._pd(p)
        Bp = B * (pd + 1)
        dev = self.device
        m = params["variational_mean"]
        LS = params["chol_variational_covar"]
        packX, Kzx, S32e, S64e = pro["packX"], pro["Kzx"], pro["S32e"], pro["S64e"]
        for t in packX:
            t.record_stream(torch.cuda.current_stream(dev))
        if not self._no_middle and self.whitening == "cholesky":
            self._last_fast = ("buffers", "A32e", hyp, pd)
        # the fast path consumes only the fp32 copy of A: with the explicit inverse the fp64 result is never stored
        A64 = None if self.trsm_nb >= Mp else self._get("A64", (Mp, Bp), f64)
        A32e = self._get("A32e", (Mp + 1, Bp), f32)          # [A ; mu_bar^T]
        A32 = A32e[:Mp]
        ws = self._bytes("trsm_ws", _lib.lib.dsvgp_trsm_workspace_bytes(Mp, max(Bp, Mp + 1), self.trsm_nb))
        if ws is not self._inverse_ws:                       # (re-allocated: the factorisation's inverse is not in it)
            _ops.trtri_blocks(ctx, L, max(Bp, Mp + 1), self.trsm_nb, ws, self._potrf_ws)
            self._inverse_ws = ws
        ev = self._event_pair()
        _ops.trsm(ctx, L, Kzx, False, A64, A32, self.trsm_nb, ws, reuse_inverse=True)       # A = L^-1 K_ZX (fp64)
        self._event_done("solve_fwd", ev)
        mu = torch.empty(Bp, dtype=f32, device=dev)
        var0 = torch.empty(Bp, dtype=f32, device=dev)
        sws = self._bytes("stats_ws", _lib.lib.dsvgp_stats_workspace_bytes(Mp, Bp))
        _ops.predictive_stats(ctx, A32, A32, pd, m, params["constant"].reshape(-1), hyp, mu, var0, sws)  # mu = A^T m + c
        mu_bar = A32e[Mp]
        sums = torch.empty(4, dtype=f32, device=dev)
        _ops.residual_terms(ctx, mu, y, hyp, rows, mu_bar, sums)
        # Gram matrix and L_S gradient
        Ge = self._get("Ge", (Mp + 1, Mp), f32)              # [G ; b^T]
        G = Ge[:Mp]
        # tril([A ; mu_bar^T] A^T) = [tril(G) ; b^T], split-K over the minibatch axis: b = A mu_bar rides along as row M'
        _ops.gemm(ctx, TRANS_B | OUT_LOWER, A32e, A32, Ge)
        coll = self.collective if self._global_gram else None
        # one rank: 2 vbar = 1 / (noise rows) multiplies the FINAL gradients on the device (the products below run unscaled),
        # so the noise never has to visit the host -- a precondition for replaying the step from a HIP graph
        dev_scale = self._dev_scale = self.collective is None or self.collective.world == 1
        handle = None
        if coll is not None:
            # data parallel, "global Gram" schedule: everything downstream of [G ; b^T] is linear in it, so the ranks sum
            # THAT (36 MB, under the Q' solve and the K_ZX-bar product) instead of the L_S gradient at the end: L_S-bar and
            # m-bar then come out global on every rank, and the replicated Cholesky backward can be split by column blocks
            self._finish_factor(ctx)
            if self.pack_reduce:                             # [tril(G) | b]: half the bytes on the wire
                handle = self._reduce_tril_async(ctx, coll, G, Ge[Mp])
            else:
                self.early_wire_numel = Ge.numel()
                handle = coll.all_reduce_async(Ge)
        else:
            _ops.mirror_lower_f32_(ctx, G, Mp)
            if not self.capture_mode:
                self._finish_factor(ctx)                     # host sync, hidden behind the queued solve + Gram product
        if dev_scale:
            vbar2 = 1.0
        else:
            noise = self._hyp_host[2]
            vbar2 = 1.0 / (noise * rows)                     # 2 * vbar

        def variational_part():
            if coll is not None and not self.deterministic:
                # global-Gram schedule: every rank forms L_S-bar itself from the same summed G and nobody reduces it again, so
                # THIS product adds its K slices in a fixed order (slab scratch, dsvgp_set_deterministic around the one call):
                # the replicas' L_S and Adam moments stay bitwise equal -- no periodic re-broadcast of the variational parameters
                ctx.set_deterministic(self._bytes("gls_slab", 4 * 4 * Mp * Mp + 4096))
                try:
                    _ops.gemm(ctx, B_LOWER | OUT_LOWER, G, LS, dLS, alpha=vbar2)
                finally:
                    ctx.set_deterministic(None)
            else:
                _ops.gemm(ctx, B_LOWER | OUT_LOWER, G, LS, dLS, alpha=vbar2)    # 2 vbar tril(G tril(L_S))
            dm.copy_(Ge[Mp])                                 # b = A mu_bar, the data part of m-bar (global under coll)
            # ONE pass over (L_S, tril(G L_S)): trace terms |L_S^T A|_F^2 and tr G, 2 vbar tril(G L_S) where the product ran
            # unscaled (one rank: dev_scale), KL value + gradient.  Global-Gram schedule: m-bar / L_S-bar are not reduced again,
            # so every rank adds the KL gradient itself ...
            add_kl = include_kl or coll is not None
            _ops.variational_terms(ctx, m, LS, num_data, dev_scale, add_kl, hyp, rows, G, 1.0 / vbar2, kl_buf, sums, dm, dLS)
            if coll is not None and not include_kl:
                sums[2:4].zero_()                            # ... the trace terms of the GLOBAL G are counted on one rank only
                kl_buf[0:1].zero_()                          # ... and so is the KL value
            _ops.elbo_fast_finalize(ctx, sums, hyp, B, pd, rows, scal)
            if coll is None:
                self._variational_grads_final()

        def solve_part():
            # Q' = L^-T (S - I), a = L^-T m  (fp64 solves), both also as fp32 copies
            Qe64 = self._get("Qe64", (Mp, (Mp + 2) // 2 * 2), f64)[:, :Mp + 1]     # (even rows: 16-byte loads in the fp32 conversion pass)
            # fp32 copy with rows padded to a multiple of 4 floats, pad zeroed once: the LDS-DMA GEMM (gemm32.hip) streams the
            # k-contiguous [Q' | a] in 16-byte chunks and takes K = M' + 1 as it lies in memory (DSVGP_GEMM_K_PADDED)
            Qe32 = self._get_zeroed("Qe32_pad", (Mp, (Mp + 1 + 3) // 4 * 4), f32)[:, :Mp + 1]
            _ops.trsm(ctx, L, S32e, True, Qe64, Qe32, self.trsm_nb, ws, reuse_inverse=True)
            # K_ZX-bar (fp32, dense): hand-written 32x32x2 MFMA kernel; the rocBLAS route stays as a diagnostics comparator
            if self.lib_dense_gemm:
                _ops.gemm_lib_f32(ctx, 0, Qe32, A32e, Kb32, alpha=vbar2)
            else:
                _ops.gemm(ctx, _lib.K_PADDED, Qe32, A32e, Kb32, alpha=vbar2)
            if self._side is not None and not self.capture_mode and not self.deterministic:   # K_ZX-bar is final: its kernel backward may start (side stream)
                self._dense_done = torch.cuda.Event()
                self._dense_done.record(torch.cuda.current_stream(dev))
            return Qe64

        shard = (coll is not None and self.shard_replicated and hasattr(coll, "all_gather_async") and Mp >= self.shard_min_mp
                 and Mp >= 4 * coll.world)
        self.sharded_stage_used = bool(shard)
        self._lbar_f32 = None

        def solve_part_sharded():
            """rank g: columns [c0, c1) of [Q' | a / (2 vbar)] = L^-T [S - I | m / (2 vbar)] (fp64 product, fp32 copy), all-gathered"""
            Gw = coll.world
            w = ((Mp + 1 + Gw - 1) // Gw + 3) // 4 * 4                      # block width: a multiple of 4 floats (16-byte loads)
            c0 = min(coll.rank * w, Mp + 1)
            c1 = min(c0 + w, Mp + 1)
            loc = self._get_zeroed("Qcols32", (Mp, w), f32)                # (pad columns beyond the matrix stay zero)
            if c1 > c0:
                q64 = self._get("Qcols64", (Mp, w), f64)
                _ops.trsm(ctx, L, S32e[:, c0:c1], True, q64[:, :c1 - c0], loc[:, :c1 - c0], self.trsm_nb, ws, reuse_inverse=True)
            allq = self._get("Qall32", (Gw, Mp, w), f32)
            return coll.all_gather_async(allq, loc), allq, w

        def dense_part_sharded(h, allq, w):
            h.wait()
            # block-column-major [rank][M'][w] -> row-major [M'][world w] (one strided copy); columns beyond M' + 1 are the ranks'
            # zero pads, so the k-contiguous operand is zero-filled past K as the LDS-DMA kernel wants it (K_PADDED)
            Qfull = self._get("Qfull32", (Mp, coll.world * w), f32)
            Qfull.view(Mp, coll.world, w).copy_(allq.permute(1, 0, 2))
            Qe32 = Qfull[:, :Mp + 1]
            _ops.gemm(ctx, _lib.K_PADDED, Qe32, A32e, Kb32, alpha=vbar2)
            if self._side is not None and not self.capture_mode and not self.deterministic:
                self._dense_done = torch.cuda.Event()
                self._dense_done.record(torch.cuda.current_stream(dev))
            return Qe32

        def lbar_sharded(Qe32):
            """rank g: rows [r0, r1) of tril(L^T L-bar) = -2 vbar tril([S - I | m/(2 vbar)][G ; b^T]) (fp64 copy of the left operand x
            fp32 [G ; b^T], fp64 accumulation: the one-GPU arithmetic), all-gathered as fp32: the argument of Phi in the
            Cholesky backward"""
            Gw = coll.world
            wr = ((Mp + Gw - 1) // Gw + 1) // 2 * 2          # (the row block of the five-piece C path: dsvgp_step_plan::wr)
            r0 = min(coll.rank * wr, Mp)
            r1 = min(r0 + wr, Mp)
            loc = self._get_zeroed("Lrows32", (wr, Mp), f32)
            if r1 > r0:
                l64 = self._get("Lrows64", (wr, Mp), f64)
                # (only the lower triangle is read: columns [0, r1); the rest of the zero-initialised block stays zero)
                _ops.gemm(ctx, 0, S64e[r0:r1], Ge[:, :r1], l64[:r1 - r0, :r1], alpha=-vbar2, C32=loc[:r1 - r0, :r1])
            allr = self._get("Lall32", (Gw * wr, Mp), f32)
            h = coll.all_gather_async(allr, loc)
            h.wait()
            self._lbar_f32 = allr[:Mp]

        var_done = None
        if shard:
            # sharded replicated stage: own columns of [Q' | a] -> all-gather (under the wait for the summed [G ; b^T] and the
            # L_S / m gradient kernels) -> dense product -> own rows of L-bar -> all-gather
            hq, allq, wq = solve_part_sharded()
            ev_w = self._event_pair()
            handle.wait()
            self._event_done("early_reduce_wait", ev_w)
            _ops.mirror_lower_f32_(ctx, G, Mp)
            if self.var_overlap and self._side is not None and not self.capture_mode and not self.deterministic:
                main = torch.cuda.current_stream(dev)
                fork = torch.cuda.Event()
                fork.record(main)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(fork)
                    ctx.bind()
                    variational_part()
                    var_done = torch.cuda.Event()
                    var_done.record(self._side)
                ctx.bind()
            else:
                variational_part()
            Qe32 = dense_part_sharded(hq, allq, wq)
            lbar_sharded(Qe32)
            if var_done is not None:
                torch.cuda.current_stream(dev).wait_event(var_done)
            return packX, mu
        if coll is None and self.var_overlap and self.collective is None and self._side is not None and not self.capture_mode \
                and not self.deterministic:
            # one GPU: the L_S / m gradients (the few-tile fp32 product G L_S, trace / KL / loss kernels) need only G; they run on
            # the side stream next to the Q' solve and the dense product, joined before L-bar
            main = torch.cuda.current_stream(dev)
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(fork)
                ctx.bind()
                variational_part()
                var_done = torch.cuda.Event()
                var_done.record(self._side)
            ctx.bind()
            Qe64 = solve_part()
        elif coll is None:
            variational_part()
            Qe64 = solve_part()
        elif self.var_overlap and self._side is not None and not self.capture_mode and not self.deterministic:
            # global-Gram schedule: the side stream waits for the summed [G ; b^T], mirrors it and forms the L_S / m gradients
            # (G L_S, trace / KL / loss kernels) while the main stream goes on with L-bar and the Cholesky backward
            Qe64 = solve_part()
            main = torch.cuda.current_stream(dev)
            fork = torch.cuda.Event()
            fork.record(main)
            ev_w = self._event_pair()                        # (bench.py: how long the main stream stalls for [G ; b^T])
            with torch.cuda.stream(self._side):
                self._side.wait_event(fork)
                ctx.bind()
                handle.wait()
                _ops.mirror_lower_f32_(ctx, G, Mp)
                g_ready = torch.cuda.Event()
                g_ready.record(self._side)
                variational_part()
                var_done = torch.cuda.Event()
                var_done.record(self._side)
            ctx.bind()
            main.wait_event(g_ready)
            self._event_done("early_reduce_wait", ev_w)
        else:
            Qe64 = solve_part()
            ev_w = self._event_pair()                        # (bench.py: how long the main stream stalls for [G ; b^T])
            handle.wait()
            self._event_done("early_reduce_wait", ev_w)
            _ops.mirror_lower_f32_(ctx, G, Mp)
            variational_part()
        _ops.gemm(ctx, OUT_LOWER, S64e, Ge, Lbar, alpha=-vbar2)                 # tril(L^T L-bar) (fp64), see the docstring
        if var_done is not None:
            torch.cuda.current_stream(dev).wait_event(var_done)
        return packX, mu
