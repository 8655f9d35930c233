/*
 * dsvgp.h -- C ABI of the MI355X (gfx950) DSVGP minibatch-ELBO hot path.
 *
 * The reference (mishapadidar/GP-Derivatives-Variational-Inference) is pure Python on top of
 * GPyTorch; it has no FFI.  These entry points are what a native binding for the hot path
 * would expose: every function takes plain device pointers, sizes and the context (which carries
 * the HIP stream), returns 0 on success, a negative DSVGP_E* code for bad arguments and a positive
 * hipError_t / rocblas_status (offset by 1000) for runtime failures.  No torch types.
 *
 * Conventions
 *   - all matrices are dense ROW-MAJOR with an explicit leading dimension (elements, not bytes);
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - the library never allocates or frees caller-visible memory; workspaces are passed in and
 *     sized by the *_workspace_bytes helpers;
 *   - kernels are enqueued on the stream given to dsvgp_set_stream and do not synchronise
 *     (exceptions are documented);
 *   - "interleaved" layout: row i*(p+1) is the function value at point i, rows i*(p+1)+1+a the
 *     derivative along point i's a-th direction (reference RBFKernelDirectionalGrad.py:105-107).
 *   - hyp is a device float[4] = { lengthscale, outputscale, noise, unused } produced by
 *     dsvgp_hyp_forward from the raw (softplus-constrained) parameters.
 *
 * Reference interface replaced by each entry point is cited as file:line under /root/reference.
 */
#ifndef DSVGP_H
#define DSVGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsvgp_ctx dsvgp_ctx;

enum {
    DSVGP_OK = 0,
    DSVGP_EINVAL = -1,     /* bad argument / unsupported shape        */
    DSVGP_ENOTPD = -2,     /* Cholesky failed (matrix not PD)         */
    DSVGP_EALIGN = -3,     /* pointer / leading dimension misaligned  */
    DSVGP_ENOSPACE = -4    /* a caller-sized buffer filled up: call again with a larger one */
};

/* ---- context ------------------------------------------------------------------------------- */
int dsvgp_create(dsvgp_ctx** ctx);                 /* creates the rocBLAS handle used by rocSOLVER potrf */
int dsvgp_destroy(dsvgp_ctx* ctx);
int dsvgp_set_stream(dsvgp_ctx* ctx, void* hip_stream);
/* Deterministic mode (round 3).  With a caller-owned scratch buffer set, every split-K product stores its K slices to slabs in
 * the scratch and adds them in a fixed order instead of meeting in floating-point atomics (fp32 Gram product, the fp64 M' x M' x M'
 * products, the small-problem split-K products), and the scalar reductions of the step (residual / likelihood sums) go through
 * per-workgroup partials: two runs on the same inputs are bitwise equal.  scratch == NULL: back to atomics (the default; their
 * rounding depends on the order in which workgroups retire).  The scratch serves the launches queued on the context's current
 * stream, one stream at a time; a product whose slices do not fit it runs with fewer, longer slices (unsplit below two).
 * The reference (CPU torch) is deterministic for a fixed seed; this mode gives the HIP path the same property.          */
int dsvgp_set_deterministic(dsvgp_ctx* ctx, void* scratch, size_t bytes);
const char* dsvgp_version(void);

/* ---- hyper-parameters: gpytorch Positive / GreaterThan(1e-4) softplus constraints ------------
 * replaces ScaleKernel.outputscale, RBFKernel.lengthscale, GaussianLikelihood.noise
 * (directionalvi/directional_vi.py:56,172).  raw = {raw_lengthscale, raw_outputscale, raw_noise}. */
int dsvgp_hyp_forward(dsvgp_ctx* ctx, const float* raw_lengthscale, const float* raw_outputscale,
                      const float* raw_noise, float* hyp);
/* chain rule back to the raw parameters: d_raw += d_hyp * sigmoid(raw) */
int dsvgp_hyp_backward(dsvgp_ctx* ctx, const float* raw_lengthscale, const float* raw_outputscale,
                       const float* raw_noise, const float* d_hyp, float* d_raw_lengthscale,
                       float* d_raw_outputscale, float* d_raw_noise);
/* The scalar tail of one step in a single launch: d_hyp += data-term scalars (scal[4], scal[3], scal[1] of
 * dsvgp_likelihood_terms / dsvgp_elbo_fast_finalize), the softplus chain rule of dsvgp_hyp_backward, d constant +=
 * scal[2], and loss = -scal[0] / rows + kl0[0] / num_data  (VariationalELBO, directional_vi.py:217,245-246).  */
int dsvgp_step_epilogue(dsvgp_ctx* ctx, const float* scal, const float* kl0, double rows, double num_data,
                        const float* raw_lengthscale, const float* raw_outputscale, const float* raw_noise,
                        float* d_hyp, float* d_raw_lengthscale, float* d_raw_outputscale, float* d_raw_noise,
                        float* d_constant, float* loss);

/* ---- kernel assembly: RBFKernelDirectionalGrad.forward (directionalvi/RBFKernelDirectionalGrad.py:41-119)
 *
 * dsvgp_pack_points: x[n,d], v[n*p,d] (raw directions, normalised here, :57-58) ->
 *   P[n*(p+1), DP] packed operand rows (x/ell and unit directions, zero padded, plus the indicator
 *   column used by the backward), self[n*(p+1)] (|x/ell|^2 and (x/ell).v), vnorm[n*p].
 *   DP = dsvgp_packed_width(d).  center[d] (may be NULL = 0): common shift subtracted from x before
 *   the 1/ell scaling; both operands of one kernel call must be packed with the SAME center.  The
 *   kernel is shift invariant; centering (gpytorch 1.4.0 Kernel.covar_dist: `adjustment =
 *   x1.mean(-2)`, reached from RBFKernelDirectionalGrad.py:71) keeps |x~|^2 small so that the fp32
 *   quadratic expansion |x~1|^2 + |x~2|^2 - 2 x~1.x~2 loses ~4x fewer digits.
 * dsvgp_column_mean: out[d] = mean over the n rows of x[n,d] (the center above).                 */
int dsvgp_packed_width(int d);
int dsvgp_column_mean(dsvgp_ctx* ctx, const float* x, int n, int d, float* out);
int dsvgp_pack_points(dsvgp_ctx* ctx, const float* x, const float* v, int n, int d, int p,
                      const float* hyp, const float* center, float* P, float* self, float* vnorm);

/* out[n1*(p+1), n2*(p+1)] = hyp.outputscale * K(x1,x2;v1,v2) (+ jitter on the global diagonal).
 * out_is_double: 0 -> float output, 1 -> double output (fp32 values widened, as the reference's
 * `.double()` cast, DirectionalGradVariationalStrategy.py:74,181).                               */
int dsvgp_kernel_fwd(dsvgp_ctx* ctx, const float* P1, const float* self1, int n1, const float* P2,
                     const float* self2, int n2, int d, int p, const float* hyp, float jitter,
                     void* out, int64_t ld, int out_is_double);
/* The same product when the directions of side 2 are CANONICAL unit vectors shared by all its points -- what the reference's callers
 * pass for K_ZX: select_cols_of_y's E_canonical[idx - 1] tiled over the minibatch (directional_vi.py:81-88, 238), eye(d)[:p] tiled in
 * eval_gp (:292-294).  dir_idx[p] (device memory): direction b is the unit vector of coordinate dir_idx[b] - idx_base (idx_base = 1 takes
 * idx_y[1:] of select_cols_of_y as it stands).  P2 / self2: the packed rows of side 2 (only the value rows are read).  float output, no
 * jitter.  Returns DSVGP_EINVAL for geometries the canonical kernels do not take (p + 1 not in {3, 6} or d > 28): use dsvgp_kernel_fwd. */
int dsvgp_kernel_fwd_canon(dsvgp_ctx* ctx, const float* P1, const float* self1, int n1, const float* P2,
                           const float* self2, int n2, int d, int p, const int* dir_idx, int idx_base,
                           const float* hyp, float* out, int64_t ld);
/* diag=True branch (:110-119): out[n*(p+1)] = outputscale * [1, 1/ell^2, ...]                    */
int dsvgp_kernel_diag(dsvgp_ctx* ctx, int n, int p, const float* hyp, float* out);

/* backward of dsvgp_kernel_fwd w.r.t. (x1, v1, lengthscale, outputscale), given G = dLoss/dOut.
 * symmetric != 0: x1==x2, v1==v2 and G symmetric (the K_ZZ case) -> point/direction grads doubled.
 * Accumulates (+=) into d_x1[n1,d], d_v1[n1*p,d] and d_hyp[0..1].  g_is_double selects G's dtype. */
size_t dsvgp_kernel_bwd_workspace_bytes(int n1, int n2, int d, int p);
int dsvgp_kernel_bwd(dsvgp_ctx* ctx, const void* G, int64_t ldg, int g_is_double, const float* P1,
                     const float* self1, const float* vnorm1, int n1, const float* P2,
                     const float* self2, int n2, int d, int p, const float* hyp, int symmetric,
                     float* d_x1, float* d_v1, float* d_hyp, void* workspace);
/* 1 when dsvgp_kernel_fwd_canon / _bwd_canon take the geometry (d, p), 0 otherwise */
int dsvgp_kernel_canon_supported(int d, int p);
/* backward of dsvgp_kernel_fwd_canon (symmetric = 0 semantics; same workspace size as dsvgp_kernel_bwd) */
int dsvgp_kernel_bwd_canon(dsvgp_ctx* ctx, const void* G, int64_t ldg, int g_is_double, const float* P1,
                           const float* self1, const float* vnorm1, int n1, const float* P2,
                           const float* self2, int n2, int d, int p, const int* dir_idx, int idx_base,
                           const float* hyp, float* d_x1, float* d_v1, float* d_hyp, void* workspace);

/* One-hot directions on BOTH sides, the same index list for every point of both: the full-gradient SVGP (reference
 * directionalvi/GradVariationalStrategy.py:89-99 builds RBFKernelGrad over cat([Z, x]) -- the directional kernel with p = d and the
 * directions I_d at every inducing and data point; BASELINE config 3) or inducing directions fixed to the data's canonical columns.
 * Every block is then an elementwise function of x1 - x2 (no products): K00 = k, K0b = k delta_b / ell, Ka0 = -k delta_a / ell,
 * Kab = k ([a == b] - delta_a delta_b) / ell^2.  P1 / P2: packed rows (only the value rows are read); dir_idx / idx_base as above.
 * _supported: 1 for the geometries taken (p + 1 = 11, d <= 12), else 0 -- the entry points return DSVGP_EINVAL and the caller uses the
 * general kernels.  fwd: float / double output, jitter on the global diagonal as dsvgp_kernel_fwd.  bwd: accumulates (+=) d_x1,
 * d_hyp[0..1] as dsvgp_kernel_bwd (symmetric != 0: K_ZZ, point gradients doubled); d_v1 receives nothing (fixed directions); G's rows
 * must consist of 16-byte pieces (ldg % 4 == 0 floats / % 2 == 0 doubles, base 16-byte aligned), DSVGP_EINVAL otherwise; workspace of
 * dsvgp_kernel_bwd_workspace_bytes(n1, n2, d, p).                                                                                   */
int dsvgp_kernel_canon2_supported(int d, int p);
int dsvgp_kernel_fwd_canon2(dsvgp_ctx* ctx, const float* P1, int n1, const float* P2, int n2, int d, int p, const int* dir_idx,
                            int idx_base, const float* hyp, float jitter, void* out, int64_t ld, int out_is_double);
int dsvgp_kernel_bwd_canon2(dsvgp_ctx* ctx, const void* G, int64_t ldg, int g_is_double, const float* P1, const float* vnorm1, int n1,
                            const float* P2, int n2, int d, int p, const int* dir_idx, int idx_base, const float* hyp, int symmetric,
                            float* d_x1, float* d_v1, float* d_hyp, void* workspace);

/* ---- fp64 model mode (the reference's experiments set torch.set_default_dtype(torch.float64),
 * experiments/synthetic/exp_script.py:56): RBFKernelDirectionalGrad.forward / backward in double precision.
 * Same packed-row formulation as the fp32 assembly; the two contractions T = P1 P2^T and dP1 = Tbar P2 go through
 * dsvgp_gemm (fp64 MFMA), these entry points do the rest:
 *   dsvgp_pack_points_f64           x[n,d], v[n*p,d] -> P[n(p+1), DP] (DP = dsvgp_packed_width(d)), self, vnorm (doubles)
 *   dsvgp_kernel_transform_f64      T[n1(p+1), n2(p+1)] -> outputscale * K in place (+ jitter on the global diagonal)
 *   dsvgp_kernel_bwd_transform_f64  (G = dLoss/dK, T) -> Tbar in place of T; d_hyp[0] += d lengthscale, d_hyp[1] += d outputscale
 *   dsvgp_kernel_bwd_points_f64     dP1[n1(p+1), DP] = Tbar [P2 | indicator] -> d_x1 +=, d_v1 += (x2 when symmetric)
 * hyp: device double[3+] = {lengthscale, outputscale, noise}.  p <= 16.                                              */
int dsvgp_pack_points_f64(dsvgp_ctx* ctx, const double* x, const double* v, int n, int d, int p, const double* hyp,
                          const double* center, double* P, double* self, double* vnorm);
int dsvgp_kernel_transform_f64(dsvgp_ctx* ctx, double* T, int64_t ld, const double* self1, int n1, const double* self2,
                               int n2, int p, const double* hyp, double jitter);
int dsvgp_kernel_bwd_transform_f64(dsvgp_ctx* ctx, const double* G, int64_t ldg, double* T, int64_t ldt, const double* self1,
                                   int n1, const double* self2, int n2, int p, const double* hyp, double* d_hyp);
 /* fp64 column statistics and their backward (DGVS.py:188,192-205): mu_j = sum_i A_ij m_i, cs_j = sum_i (W_ij^2 - A_ij^2) (W may be
 * NULL: mean only);  Abar = m mu_bar^T + 2 (U - A) diag(var_bar), Av = 2 A diag(var_bar) (U NULL: U = A; Av may be NULL) */
int dsvgp_colstats_f64(dsvgp_ctx* ctx, const double* A, int64_t lda, const double* W, int64_t ldw, const double* m, int Mp, int Bp,
                       double* mu, double* cs);
int dsvgp_abar_f64(dsvgp_ctx* ctx, const double* A, int64_t lda, const double* U, int64_t ldu, const double* m, const double* mu_bar,
                   const double* var_bar, int Mp, int Bp, double* Abar, int64_t ldo, double* Av, int64_t ldv);
/* likelihood terms of the fp64 model (GaussianLikelihood + VariationalELBO mll_type 0 / PredictiveLogLikelihood 1,
 * directional_vi.py:172,217,245-246): mu = mu0 + constant, varn = max(s dg + 1e-4 + cs + noise, 1e-6) per output (p derivative
 * outputs per point), mu_bar / var_bar = d loss / d mu0, d cs with loss = -(sum ll) / rows, and
 * scal[8] (zeroed here) = {sum ll, d/d noise, d/d constant, d/d outputscale, d/d lengthscale (prior-diagonal parts), 0, 0, 0}  */
int dsvgp_likelihood_terms_f64(dsvgp_ctx* ctx, const double* mu0, const double* cs, const double* y, const double* constant,
                               int ncols, int p, const double* hyp, int mll_type, double rows, double* mu, double* varn,
                               double* mu_bar, double* var_bar, double* scal);
/* scalar tail of the fp64 ELBO fast path (the variances enter the ELBO only through their sum: tvar[0] = tr(L_S^T G L_S) - tr G on
 * the device): mu = mu0 + constant, mu_bar = d loss / d mu0, and scal[8] = {sum ll, d loss / d noise, d / d constant, d / d outputscale,
 * d / d lengthscale (prior-diagonal parts), vbar = d loss / d tvar = 1 / (2 noise rows), sum r^2, sum r}; ncols = npts (pd + 1) outputs
 * (GaussianLikelihood.expected_log_prob summed, VariationalELBO: directional_vi.py:217,245-246)                                   */
int dsvgp_elbo_fast_tail_f64(dsvgp_ctx* ctx, const double* mu0, const double* y, const double* constant, int ncols, int npts, int pd,
                             const double* hyp, const double* tvar, double rows, double* mu, double* mu_bar, double* scal);
int dsvgp_kernel_bwd_points_f64(dsvgp_ctx* ctx, const double* dP, const double* P1, const double* vnorm1, int n1, int d,
                                int p, const double* hyp, int symmetric, double* d_x1, double* d_v1);

/* ---- Cholesky: psd_safe_cholesky(K_ZZ.double()) (DirectionalGradVariationalStrategy.py:72-75)
 * In-place lower Cholesky of the row-major fp64 matrix A[n,n] (rocSOLVER dpotrf); only the lower
 * triangle is read/written.  info_dev is a device int (0 = ok, k>0 = leading minor k not PD).   */
/* algo 0: rocSOLVER dpotrf (no workspace).  algo 1: blocked right-looking Cholesky, one fused fp64 MFMA launch per
 * 64-column block (trailing update + factorisation / inversion of the next diagonal block out of LDS) and one
 * batched panel launch; workspace of dsvgp_potrf_workspace_bytes.                                           */
size_t dsvgp_potrf_workspace_bytes(int n, int algo);
int dsvgp_potrf(dsvgp_ctx* ctx, double* A, int n, int64_t lda, int* info_dev, int algo, void* workspace);
/* A.diagonal() += delta   (the jitter retries of psd_safe_cholesky)                              */
int dsvgp_add_diag(dsvgp_ctx* ctx, double* A, int n, int64_t lda, double delta);

/* ---- panel triangular solve on MFMA: TriangularLazyTensor.inv_matmul (:181,183)
 * Solves op(L) X = B for the lower-triangular fp64 L[n,n]; trans=0: op(L)=L, trans=1: op(L)=L^T.
 * B[n,nrhs] is float or double (b_is_double); X64[n,nrhs] double output (may alias B when B is
 * double); X32 optional float copy (NULL to skip); X64 may be NULL when X32 is given and nb >= n
 * (single product with the explicit inverse).  Diagonal blocks of size nb are inverted
 * (stored in the workspace) and every update is a v_mfma_f64_16x16x4 GEMM.
 * WORKSPACE SIZE: `workspace` must hold dsvgp_trsm_workspace_bytes(n, nrhs OF THIS CALL, nb) bytes, ALSO with
 * reuse_inverse=1 and ALSO when X64 is NULL: besides the inverted blocks it carries an n x nrhs double scratch that
 * a small fp32-only solve (nb >= n, fewer than 1024 output tiles) uses as the fp64 target of its split-K product.
 * The entry point has no size argument; an undersized workspace is an out-of-bounds write.                       */
size_t dsvgp_trsm_workspace_bytes(int n, int nrhs, int nb);
int dsvgp_trsm(dsvgp_ctx* ctx, const double* L, int64_t ldl, int n, int trans, const void* B,
               int64_t ldb, int b_is_double, int nrhs, double* X64, int64_t ldx64, float* X32,
               int64_t ldx32, int nb, void* workspace, int reuse_inverse);
/* The inversion phase of dsvgp_trsm alone (later solves pass reuse_inverse=1).  potrf_workspace: NULL, or
 * the workspace dsvgp_potrf(algo 1) factored THIS L with (its inverted 64x64 diagonal blocks are reused). */
int dsvgp_trtri(dsvgp_ctx* ctx, const double* L, int64_t ldl, int n, int nb, const void* potrf_workspace,
                void* workspace);
/* dsvgp_potrf(algo 1) and the inversion phase in the SAME launches (blocked Cholesky with a fused forward elimination of
 * the identity): A <- L in place, L^-1 and its transpose into `workspace` (the dsvgp_trsm workspace; later solves pass
 * reuse_inverse=1).  Only for the single-product regime nb >= n; potrf_workspace sized by
 * dsvgp_potrf_workspace_bytes(n, 1).                                                                              */
int dsvgp_potrf_inverse(dsvgp_ctx* ctx, double* A, int n, int64_t lda, int* info_dev, void* potrf_workspace, int nb,
                        void* workspace);

/* ---- generic MFMA GEMM used by the predictive / backward contractions
 *   C = alpha * op(A) op(B) + beta * Cin,   compute type = double (is_double=1) or float.
 * flags: see DSVGP_GEMM_* ; kscale (optional, length K, float) multiplies op(A)[:,k].             */
enum {
    DSVGP_GEMM_TRANS_A = 1,      /* A stored [K,M]                                             */
    DSVGP_GEMM_TRANS_B = 2,      /* B stored [N,K]                                             */
    DSVGP_GEMM_A_LOWER = 4,      /* op(A)[m,k] == 0 for k > m (masked, memory above ignored)   */
    DSVGP_GEMM_A_UPPER = 8,      /* op(A)[m,k] == 0 for k < m                                  */
    DSVGP_GEMM_B_LOWER = 16,     /* op(B)[k,n] == 0 for n > k                                  */
    DSVGP_GEMM_B_UPPER = 32,     /* op(B)[k,n] == 0 for n < k                                  */
    DSVGP_GEMM_OUT_LOWER = 64,   /* only m >= n is computed; m < n is written as 0             */
    DSVGP_GEMM_B_IS_FLOAT = 128, /* (double compute only) B is float                           */
    DSVGP_GEMM_CIN_IS_FLOAT = 256,/* (double compute only) Cin is float                        */
    DSVGP_GEMM_K_PADDED = 512,   /* (float compute) operands stored with K as their minor axis are zero-filled by the
                                    caller from K up to the next multiple of 4 (lets the LDS-DMA kernel take K % 4 != 0) */
    DSVGP_GEMM_BACKGROUND = 1024 /* filler product overlapped with a latency-bound chain on another stream: launched with one
                                    workgroup per CU, so that every CU keeps LDS room for a workgroup of the chain */
};
/* dst[M, N] (double) = src[M, N] (float): the fp64 copy of [S - I | m / (2 vbar)] that the Cholesky backward multiplies with
 * [G ; b^T] under fp64 accumulation ("Phi(L^T L-bar) without L-bar")                              */
int dsvgp_widen_f32_f64(dsvgp_ctx* ctx, const float* src, int64_t ld, double* dst, int64_t ldd, int M, int N);
int dsvgp_gemm(dsvgp_ctx* ctx, int is_double, int flags, int M, int N, int K, double alpha,
               const void* A, int64_t lda, const void* B, int64_t ldb, double beta, const void* Cin,
               int64_t ldcin, void* C, int64_t ldc, float* C32, int64_t ldc32, const float* kscale);

/* ---- ELBO terms: GaussianLikelihood.expected_log_prob / log_marginal + VariationalELBO
 * (directional_vi.py:217-219,245-246) on top of the predictive of
 * DirectionalGradVariationalStrategy.py:188-205.
 *
 * dsvgp_predictive_stats: mu[j] = sum_i A[i,j] m[i] + c ; var[j] = s*dg_j + 1e-4 + sum_i (W^2 - A^2)
 * with A = L^-1 K_ZX, W = L_S^T A, both [Mp, nb] float.                                          */
size_t dsvgp_stats_workspace_bytes(int Mp, int ncols);
int dsvgp_predictive_stats(dsvgp_ctx* ctx, const float* A, int64_t lda, const float* W, int64_t ldw,
                           int Mp, int ncols, int p, const float* m, const float* constant,
                           const float* hyp, float* mu, float* var, void* workspace);
/* per-output log-likelihood terms and their gradients.  mll_type 0 = ELBO, 1 = PLL.
 * out_scalars (device float[8]): {sum_ll, d_noise, d_constant, d_outputscale(diag part),
 *  d_lengthscale(diag part), 0,0,0}; mu_bar/var_bar are dLoss/dmu, dLoss/dvar with
 *  loss = -(sum_ll/global_rows - KL/num_data).  varn_out = clamp(var + noise) (likelihood variance). */
int dsvgp_likelihood_terms(dsvgp_ctx* ctx, const float* mu, const float* var, const float* y,
                           int ncols, int p, const float* hyp, int mll_type, double global_rows,
                           float* mu_bar, float* var_bar, float* varn_out, float* out_scalars);
/* Abar[i,j] = m[i]*mu_bar[j] + 2*var_bar[j]*(U[i,j] - A[i,j]),   U = L_S W                       */
int dsvgp_abar(dsvgp_ctx* ctx, const float* A, int64_t lda, const float* U, int64_t ldu, int Mp,
               int ncols, const float* m, const float* mu_bar, const float* var_bar, float* Abar,
               int64_t ldab);
/* d_m[i] += sum_j A[i,j]*mu_bar[j]                                                                */
int dsvgp_rowdot(dsvgp_ctx* ctx, const float* A, int64_t lda, int Mp, int ncols, const float* vec,
                 float* out_accum);
/* KL(q(u)||N(0,I)) and its gradients (gpytorch kl_mvn_mvn with the whitened prior of
 * DirectionalGradVariationalStrategy.py:77-87): kl_out (1+Mp floats): kl_out[0] = KL, rest scratch;
 * d_m += m/num_data ; d_LS(lower) += (L_S - diag(1/L_S_ii))/num_data ; d_LS(upper) = 0.          */
int dsvgp_kl_terms(dsvgp_ctx* ctx, const float* m, const float* LS, int64_t ldls, int Mp,
                   double num_data, float* kl_out, float* d_m, float* d_LS, int64_t lddls);
/* Cholesky backward helper: S = Phi(G) + Phi(G)^T from the lower triangle of G (in place).        */
int dsvgp_phi_symmetrize(dsvgp_ctx* ctx, double* G, int n, int64_t ldg);
/* out[cols, rows] = in[rows, cols]^T (out of place)                                              */
int dsvgp_transpose_f64(dsvgp_ctx* ctx, const double* in, int64_t ldi, int rows, int cols, double* out,
                        int64_t ldo);
int dsvgp_transpose_f32(dsvgp_ctx* ctx, const float* in, int64_t ldi, int rows, int cols, float* out,
                        int64_t ldo);
/* fp64 matrix-vector product on a row-major M x N matrix: trans = 0: y[M] = A x[N];  trans = 1: y[N] = A^T x[M] (sums meet in
 * fp64 atomics: run-order rounding).  The float64 model mode's predictive mean A^T m and b = A mu-bar
 * (DirectionalGradVariationalStrategy.py:181-183 under torch.set_default_dtype(torch.float64), experiments/synthetic/exp_script.py:56). */
int dsvgp_gemv_f64(dsvgp_ctx* ctx, int trans, const double* A, int64_t lda, int M, int N, const double* x, double* y);

/* ---- ELBO-mode fast path.  With mll_type == ELBO, dLoss/dvar_j = 1/(2 noise rows) is the same for every
 * output, so the data term only needs  sum_j (y_j - mu_j)^2  and  sum_j var_j = prior + |L_S^T A|_F^2 - |A|_F^2,
 * i.e. the M' x M' Gram matrix G = A A^T instead of W = L_S^T A, U = L_S W and the [M', B'] fp64 products of the
 * backward (same quantities as DirectionalGradVariationalStrategy.py:188-205 + directional_vi.py:245-246).
 * sums (device float[4]) = { sum r^2, sum mu_bar, sum_{i>=j} L_S_ij (G L_S)_ij, trace G }.                    */
int dsvgp_residual_terms(dsvgp_ctx* ctx, const float* mu, const float* y, int ncols, const float* hyp,
                         double global_rows, float* mu_bar, float* sums);          /* zeroes sums first  */
int dsvgp_trace_terms(dsvgp_ctx* ctx, const float* LS, int64_t ldls, const float* T1, int64_t ldt,
                      const float* G, int64_t ldg, int n, float t1_scale,          /* T1 = t1_scale * given */
                      float* sums);                                                /* adds to sums[2..3] */
/* out_scalars: same layout as dsvgp_likelihood_terms                                                    */
int dsvgp_elbo_fast_finalize(dsvgp_ctx* ctx, const float* sums, const float* hyp, int npts, int p,
                             double global_rows, float* out_scalars);
/* copy the lower triangle of the float matrix onto its upper triangle (symmetrise a tril GEMM result)   */
int dsvgp_mirror_lower_f32(dsvgp_ctx* ctx, float* G, int n, int64_t ldg);
int dsvgp_add_diag_f32(dsvgp_ctx* ctx, float* A, int n, int64_t lda, float delta);
/* A[n, n + 1] (lda >= n + 1): A[i][i] -= 1, A[i][n] = m[i] * noise * rows -- [S - I | m / (2 vbar)], the right-hand side of the
 * [Q' | a] solve of the ELBO fast path (DGVS.py:192-205: the (S - I) middle term and the mean term) in one pass            */
int dsvgp_sminus_i_col(dsvgp_ctx* ctx, float* A, int n, int64_t lda, const float* m, const float* hyp, double rows);

/* ---- data-parallel all-reduce operand (the reference is single-process and has no counterpart):
 * dst = [ packed lower triangle of src[n,n] (row i at offset i(i+1)/2) | extra[nextra] ], i.e. n(n+1)/2 + nextra floats
 * instead of n*n + nextra ([tril(G) ; b^T] or [tril(L_S-bar) ; m-bar]: 18 MB instead of 36 MB at M' = 3000).
 * dsvgp_tril_unpack_f32 writes the lower triangle of dst (the strict upper part is left untouched) and extra back.  */
int dsvgp_tril_pack_f32(dsvgp_ctx* ctx, const float* src, int64_t ld, int n, const float* extra, int nextra,
                        float* dst);
int dsvgp_tril_unpack_f32(dsvgp_ctx* ctx, const float* src, int n, float* dst, int64_t ld, float* extra, int nextra);

/* ---- minibatch gather: DataLoader batch + select_cols_of_y (directional_vi.py:68-90,229-241)
 * xb[b,:] = X[idx[b],:] ; yb[b*(p+1)+c] = Y[idx[b], cols[c]]  (cols[0] == 0); with E[d, d] != NULL also the batch's
 * derivative directions Db[(b p + a), :] = E[cols[a+1] - 1, :]  (derivative_directions.repeat(n_samples, 1), :238)  */
int dsvgp_gather_batch(dsvgp_ctx* ctx, const float* X, const float* Y, const int64_t* idx, int nb,
                       int d, int ycols, const int* cols, int p, float* xb, float* yb, const float* E, float* Db);

/* ---- fused Adam (torch.optim.Adam semantics, directional_vi.py:193-199,251-254)
 * lr_dev / step_dev are device scalars so the update is graph-capturable.                         */
int dsvgp_adam_step(dsvgp_ctx* ctx, float* param, const float* grad, float* exp_avg,
                    float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step);

/* the same update for up to DSVGP_ADAM_MAX_TENSORS tensors that share (lr, betas, eps, step) in ONE launch: the parameter
 * groups of one torch.optim.Adam (directional_vi.py:193-199); the arrays are HOST arrays of device pointers / element counts
 * (round 6) sizes[k] = -n names an n x n row-major matrix of which only the LOWER TRIANGLE is a parameter (chol_variational_covar: the
 * reference masks the rest in its forward -- variational_strategy / CholeskyVariationalDistribution -- so gradient, both moments and the
 * update are exactly zero there): the strict upper triangle is then not read or written (dsvgp_adam_step_multi only).             */
#define DSVGP_ADAM_MAX_TENSORS 16
int dsvgp_adam_step_multi(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads,
                          float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* sizes, float lr,
                          float beta1, float beta2, float eps, int step);
/* the same update for the float64 model mode (parameters, gradients and moments in double) */
/* dsvgp_adam_step_multi skipped ON THE DEVICE while *guard_dev != 0 (round 6): guard_dev = the status word of the one-call step that made the
 * gradients (dsvgp_elbo_step_locate, which = 5): a host that queues the update without waiting for the factorisation's status leaves the
 * parameters untouched when it failed, reads the status later (dsvgp_elbo_step_status) and repeats the step through the jitter ladder. */
int dsvgp_adam_step_multi_guarded(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads,
                                  float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* sizes, float lr,
                                  float beta1, float beta2, float eps, int step, const int* guard_dev);
int dsvgp_adam_step_multi_f64(dsvgp_ctx* ctx, int count, double* const* params, const double* const* grads,
                              double* const* exp_avgs, double* const* exp_avg_sqs, const int64_t* sizes, double lr,
                              double beta1, double beta2, double eps, int step);

/* ---- graph-capturable variants (HIP-graph replay of the steady-state step: every per-step scalar lives in device memory)
 * dsvgp_adam_step_multi_dev: dsvgp_adam_step_multi with {lr, step} read from the device float[2] lr_step_dev (the host
 *   refreshes it through a captured copy; directional_vi.py:251-254 steps the LR schedulers every iteration) and an optional
 *   guard: while *guard_dev != 0 (the potrf status word) nothing is updated.
 * dsvgp_scale_by_vbar: x_k *= 1 / (noise * global_rows) (= 2 dLoss/dvar of the ELBO) for up to three arrays.
 * dsvgp_kl_terms_scaled: dsvgp_kl_terms with d_LS(lower) first multiplied by 1 / (noise * global_rows); add_kl = 0 leaves the
 *   KL value and gradients out (kl_out[0] = 0).                                                                          */
int dsvgp_adam_step_multi_dev(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads,
                              float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* sizes,
                              const float* lr_step_dev, float beta1, float beta2, float eps, const int* guard_dev);
int dsvgp_scale_by_vbar(dsvgp_ctx* ctx, float* x0, int64_t n0, float* x1, int64_t n1, float* x2, int64_t n2,
                        const float* hyp, double global_rows);
int dsvgp_kl_terms_scaled(dsvgp_ctx* ctx, const float* m, const float* LS, int64_t ldls, int Mp, double num_data, int add_kl,
                          const float* hyp, double global_rows, float* kl_out, float* d_m, float* d_LS, int64_t lddls);
/* The variational block of the ELBO fast path in ONE pass over (L_S, T = tril(G L_S)) (round 3; replaces the sequence
 * dsvgp_trace_terms + dsvgp_kl_terms[_scaled] on the hot path): sums[2] = t1_scale * sum_{i>=j} L_S,ij T_ij, sums[3] = trace(G),
 * d_LS <- [T / (noise rows) when flags & 1] + [dKL/dL_S / num_data when flags & 2], d_m += m / num_data (flags & 2),
 * kl_out[0] = KL (0 without flag 2).  kl_out: 1 + 2 Mp floats.  One wave per row, 16-byte accesses, no atomics: the per-row
 * partial sums are added in a fixed order (bitwise reproducible).  Reference: gpytorch kl_divergence of the whitened prior
 * (DGVS.py:77-87) + the trace terms of the expected log-likelihood (directional_vi.py:217,245).                            */
int dsvgp_variational_terms(dsvgp_ctx* ctx, const float* m, const float* LS, int64_t ldls, int Mp, double num_data,
                            int flags, const float* hyp, double global_rows, const float* G, int64_t ldg,
                            float t1_scale, float* kl_out, float* sums, float* d_m, float* d_LS, int64_t lddls);

/* ---- fp32-equivalent products on the bf16 matrix pipe (round 4, OPT-IN; the default step computes in fp32 MFMA).  An fp32 operand
 * is cut once into three bf16 planes x = h + m + l (dsvgp_split3_bf16; remainder < 2^-26 |x|) and a product keeps the six terms
 * down to 2^-16 of (h + m + l)(h' + m' + l') on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (csrc/gemm3b.hip).
 *   planes of a [rows_out, K] operand: 3 x rows_out x dsvgp_split3_kpad(K) bf16, plane after plane (dsvgp_split3_bytes), 16-byte
 *   aligned; transpose = 1 splits the TRANSPOSE of src[R, Cc] (rows_out = Cc, K = R): both operands of dsvgp_gemm3b are
 *   k-contiguous.   dsvgp_gemm3b: C[M, N] = alpha A B^T, flags 0 or DSVGP_GEMM_OUT_LOWER.              */
int dsvgp_split3_kpad(int K);
size_t dsvgp_split3_bytes(int rows_out, int K);
int dsvgp_split3_bf16(dsvgp_ctx* ctx, const float* src, int64_t ld, int R, int Cc, int transpose, void* planes);
int dsvgp_gemm3b(dsvgp_ctx* ctx, int flags, int M, int N, int K, float alpha, const void* Aplanes, int a_rows, const void* Bplanes,
                 int b_rows, float* C, int64_t ldc);

/* Plain dense fp32 GEMM through rocBLAS (row-major, flags: DSVGP_GEMM_TRANS_A / _TRANS_B only): for products without
 * structure or fused epilogue (the dense K_ZX-bar product of the ELBO fast path); everything else is dsvgp_gemm.      */
int dsvgp_gemm_lib_f32(dsvgp_ctx* ctx, int flags, int M, int N, int K, float alpha, const float* A, int64_t lda,
                       const float* B, int64_t ldb, float beta, float* C, int64_t ldc);

/* ---- contour-integral-quadrature whitening: lazify(K_ZZ).sqrt_inv_matmul(K_ZX)
 * (directionalvi/CiqDirectionalGradVariationalStrategy.py:255-256; quadrature + msMINRES of gpytorch 1.4.0
 * utils/contour_integral_quad.py, utils/minres.py).  Layout: one right-hand side per ROW, R[t, n], K[n, n] symmetric.
 * dsvgp_ciq_lanczos: `iters` Lanczos steps from v0 -> alpha[iters], beta[iters] (host takes the Ritz values of the
 *   tridiagonal as eigenvalue bounds, max_lanczos_iter = 20).  workspace: 3 n + 2 floats.
 * dsvgp_ciq_solve: out[t, n] = sum_q omega_q (K + sigma_q I)^-1 R for all shifts from one Lanczos process per row; stops
 *   when the mean of |update| / |solution| over (shift, row) < tol, tested every `check_every` iterations (gpytorch: tol 1e-4,
 *   every 10, at most 1000).  Basis-resident msMINRES: basis[cap + 1][t][n] receives the Lanczos rows q_0 .. q_J, the
 *   per-shift solutions are not stored but left as coefficients, x_q[row] = rnorm[row] sum_j ycoef[row][j][q] q_j[row]
 *   (ycoef[t][cap][QP], QP = Q rounded up to a multiple of 4); DSVGP_ENOSPACE when J would exceed cap.
 *   workspace: dsvgp_ciq_workspace_bytes(Q, t, n, cap).
 * dsvgp_ciq_mix: out[k][row][:] = rowscale[row] sum_{j<J} C[row][j][k] basis[j][row][:], k < Kout (C[t][ldj][KP], KP % 4 == 0):
 *   materialises solves (C = ycoef) or any per-row combination of them.
 * dsvgp_ciq_cross: C[t][Jb][KPa] = rn_a rn_b sum_q omega_q ya[.][ia][q] yb[.][jb][q] (KPa = Ja rounded up to 4): with it the
 *   backward's sum_q omega_q A_q^T B_q = stack_ia(basisA)^T stack_ia(mix(basisB, C)) is ONE product of depth Ja t instead of Q
 *   products of depth t (gpytorch's sqrt_inv_matmul backward, utils/contour_integral_quad.py + functions/_sqrt_inv_matmul.py).
 * dsvgp_ciq_rowstats / dsvgp_ciq_tbar: the mean / variance interpolation terms of _NgdInterpTerms (:65-69,265-266)
 *   and the gradient factors of its backward (:94-118) for rows of T = (K^-1/2 K_ZX)^T and ST = T S.
 * dsvgp_sym_average_f32: out = (A + A^T) / 2.                                                                  */
size_t dsvgp_ciq_workspace_bytes(int Q, int t, int n, int cap);
int dsvgp_ciq_lanczos(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* v0, int n, int iters, float* alpha,
                      float* beta, void* workspace);
int dsvgp_ciq_solve(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* R, int64_t ldr, int t, int n,
                    const float* sigma, const float* omega, int Q, float tol, int max_iter, int check_every, float* basis,
                    int cap, float* ycoef, float* rnorm, float* out, int64_t ldo, void* workspace, int* iters_out);
int dsvgp_ciq_mix(dsvgp_ctx* ctx, const float* basis, int J, int t, int n, const float* C, int ldj, int KP, int Kout,
                  const float* rowscale, float* out, int64_t ldo);
int dsvgp_ciq_cross(dsvgp_ctx* ctx, const float* ya, int Ja, int lda, const float* yb, int Jb, int ldb, const float* omega,
                    int Q, int t, const float* rn_a, const float* rn_b, float* Cout);
int dsvgp_ciq_rowstats(dsvgp_ctx* ctx, const float* T, const float* ST, int t, int n, int p, const float* m,
                       const float* constant, const float* hyp, float kxx_jitter, float* imean, float* mu, float* var,
                       float* live);   /* kxx_jitter: 0 for the directional strategy (:235-239), 1e-4 for gpytorch's plain
                                          CiqVariationalStrategy (its forward is quoted at :243-251)                    */
int dsvgp_ciq_tbar(dsvgp_ctx* ctx, const float* T, const float* ST, int t, int n, const float* m, const float* mu_bar,
                   const float* var_bar, const float* live, const float* imean, float* Tbar, float* VT, float* cvec);
int dsvgp_sym_average_f32(dsvgp_ctx* ctx, const float* A, int n, int64_t lda, float* out, int64_t ldo);
/* The same entry points in double precision, for a model built under torch.set_default_dtype(torch.float64) with
 * use_ciq=True (reference experiments/bunny/exp_bunny.py:66,78): every array is double, the arithmetic is the float64
 * msMINRES of gpytorch run on a float64 LazyTensor; workspace of dsvgp_ciq_lanczos_f64: 3 n + 2 doubles.            */
size_t dsvgp_ciq_workspace_bytes_f64(int Q, int t, int n, int cap);
int dsvgp_ciq_lanczos_f64(dsvgp_ctx* ctx, const double* K, int64_t ldk, const double* v0, int n, int iters, double* alpha,
                          double* beta, void* workspace);
int dsvgp_ciq_solve_f64(dsvgp_ctx* ctx, const double* K, int64_t ldk, const double* R, int64_t ldr, int t, int n,
                        const double* sigma, const double* omega, int Q, double tol, int max_iter, int check_every,
                        double* basis, int cap, double* ycoef, double* rnorm, double* out, int64_t ldo, void* workspace,
                        int* iters_out);
int dsvgp_ciq_mix_f64(dsvgp_ctx* ctx, const double* basis, int J, int t, int n, const double* C, int ldj, int KP, int Kout,
                      const double* rowscale, double* out, int64_t ldo);
int dsvgp_ciq_cross_f64(dsvgp_ctx* ctx, const double* ya, int Ja, int lda, const double* yb, int Jb, int ldb,
                        const double* omega, int Q, int t, const double* rn_a, const double* rn_b, double* Cout);
int dsvgp_ciq_rowstats_f64(dsvgp_ctx* ctx, const double* T, const double* ST, int t, int n, int p, const double* m,
                           const double* constant, const double* hyp, double kxx_jitter, double* imean, double* mu,
                           double* var, double* live);
int dsvgp_ciq_tbar_f64(dsvgp_ctx* ctx, const double* T, const double* ST, int t, int n, const double* m, const double* mu_bar,
                       const double* var_bar, const double* live, const double* imean, double* Tbar, double* VT,
                       double* cvec);
int dsvgp_sym_average_f64(dsvgp_ctx* ctx, const double* A, int n, int64_t lda, double* out, int64_t ldo);

/* ---- the whole ELBO step from ONE host call (round 3; csrc/step.hip)
 * dsvgp_elbo_step_f32 queues forward + backward of one minibatch ELBO evaluation -- `output = model(x, derivative_directions=D);
 * loss = -mll(output, y); loss.backward()` of the reference's train_gp (directionalvi/directional_vi.py:245-249) with the
 * composition of DirectionalGradVariationalStrategy.forward (DGVS.py:89-208) -- as ~70 launches on two HIP streams, without a
 * host synchronisation: the ELBO fast path (Gram formulation) of the Cholesky-whitened strategy, every data
 * point with its p directional derivatives, explicit-inverse regime (M(p+1) <= 8192).
 *   plan       host object for one (M, d, p, B): workspace layout, the second stream, events, pinned status word
 *   workspace  caller-owned device memory of dsvgp_elbo_step_workspace_bytes(M, d, p, B) bytes, 256-byte aligned, kept between steps
 *              and used by THIS plan only (the plan clears the zero padding of one operand once per workspace address; pass
 *              flag 8 whenever anything else may have written to the buffer since this plan's last step)
 *   io         device pointers (below); io->flat[0 .. flat_floats) is cleared by the call and must contain every gradient slot
 *   flags      1: overlap on the plan's second stream; 2: include the KL term (data-parallel ranks > 0 leave it out); 4: record
 *              HIP-event timings (dsvgp_elbo_step_timings); 8: the workspace contents are undefined (re-clear the paddings);
 *              16 (with 1): the Cholesky backward on the second stream under the dense K_ZX-bar product (probe: no gain);
 *              32: the two big fp32 products as bf16 x 3 split products (io->split_ws; opt-in, see dsvgp_split3_bf16);
 *              64: tril(L^T L-bar) = -tril([S - I | m'][G ; b^T]) with fp64 accumulation (default: both operands are fp32 data, the
 *              product runs on the fp32 MFMA kernel and its result is widened for the fp64 Cholesky backward)
 * Gradients are those of loss = -(sum_j ll_j / global_rows - KL / num_data); a factorisation that fails leaves NaNs in the
 * outputs and a non-zero status word: read it with dsvgp_elbo_step_status (waits for the factorisation only, not for the step)
 * and run the jitter ladder on the piecewise path.  Threading: one host thread per context.                                  */
typedef struct dsvgp_step_plan dsvgp_step_plan;
typedef struct dsvgp_elbo_step_io {
    /* parameters (float32, device): inducing points [M,d], directions [M p,d], q(u) mean [M'], Cholesky factor [M',M'] (lower
     * triangle read), constant mean [1], raw hyper-parameters [1] each                                                       */
    const float *Z, *V, *m, *LS; int64_t ldls;
    const float *constant, *raw_lengthscale, *raw_outputscale, *raw_noise;
    /* minibatch: x [B,d], interleaved targets y [B(p+1)], derivative directions D [B p, d]                                    */
    const float *x, *y, *D;
    /* outputs: the flat gradient buffer (cleared here) and the slots inside it; d_hyp[4] is scratch inside flat as well       */
    float* flat; size_t flat_floats;
    float *dZ, *dV, *dm, *dLS; int64_t lddls;
    float *d_hyp, *d_constant, *d_raw_lengthscale, *d_raw_outputscale, *d_raw_noise, *loss;
    float* mu;                       /* [B(p+1)] predictive mean of q(f) at the batch (constant mean included)                  */
    double num_data, global_rows;    /* VariationalELBO num_data ((d+1) N); B'(global) of the minibatch                          */
    float kzz_jitter;                /* LazyTensor.add_jitter() default 1e-3 (DGVS.py:144)                                      */
    /* flag 32 (opt-in): scratch of dsvgp_elbo_step_split_bytes(M, d, p, B) bytes for the bf16 plane triples of [A ; mu_bar^T],
     * its transpose and [Q' | a] -- the Gram product and the dense K_ZX-bar product then run on the bf16 matrix pipe as six
     * bf16 products per fp32 product (csrc/gemm3b.hip); NULL / flag clear: v_mfma_f32_32x32x2_f32 (the default)                */
    void* split_ws; size_t split_ws_bytes;
    /* optional (NULL: not stated): the minibatch's derivative directions as an INDEX LIST.  The reference's training step builds D as
     * E_canonical[idx - 1] tiled over the minibatch (directional_vi.py:81-88, 238: the same p coordinates for every point), its
     * evaluation as eye(d)[:p] tiled (:292-294).  dir_idx [p] (device, int32): row j p + b of D is e_{dir_idx[b] - dir_idx_base}
     * for every point j -- the caller's statement about D (which is still read: the packed rows).  K_ZX and its backward then run
     * on the canonical-direction kernels (dsvgp_kernel_fwd_canon / _bwd_canon) where those take the geometry (p + 1 in {3, 6},
     * d <= 28), on the general ones otherwise.  Appended in round 6: callers that zero-initialise the struct keep the old meaning. */
    const int* dir_idx; int dir_idx_base;
    /* non-zero: the inducing directions V are the SAME unit vectors (row m p + a of V is e_{dir_idx[a] - dir_idx_base} for every
     * inducing point m) and are not parameters -- the full-gradient SVGP (reference GradVariationalStrategy.py:89-99).  K_ZZ, K_ZX
     * and their backwards then run on dsvgp_kernel_fwd_canon2 / _bwd_canon2 where those take the geometry; dV is left zero.        */
    int v_one_hot;
} dsvgp_elbo_step_io;
size_t dsvgp_elbo_step_split_bytes(int M, int d, int p, int B);
size_t dsvgp_elbo_step_workspace_bytes(int M, int d, int p, int B);
int dsvgp_elbo_step_plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, dsvgp_step_plan** out);
int dsvgp_elbo_step_plan_destroy(dsvgp_step_plan* plan);
size_t dsvgp_elbo_step_plan_bytes(const dsvgp_step_plan* plan);
int dsvgp_elbo_step_f32(dsvgp_ctx* ctx, dsvgp_step_plan* plan, const dsvgp_elbo_step_io* io, void* workspace,
                        size_t workspace_bytes, int flags);
int dsvgp_elbo_step_status(dsvgp_step_plan* plan, float* hyp4, int* info);
/* The PER-OUTPUT step from one host call (round 5): objectives that read every output's own predictive variance -- mll_type = "PLL"
 * (PredictiveLogLikelihood, reference directionalvi/directional_vi.py:218-219; what tests/test_grad_svgp.py:19-36 trains with) or the
 * ELBO with the per-output variances returned -- for which the Gram formulation of dsvgp_elbo_step_f32 does not apply: forward
 * A = L^-1 K_ZX, W = L_S^T A, mu / var per output (DGVS.py:181-205), likelihood terms, and the backward through U = L_S W,
 * A-bar, K_ZX-bar = L^-T A-bar, L-bar = -tril(K_ZX-bar A^T), the Cholesky factor and both kernel assemblies.  Same io as above;
 * plan from dsvgp_elbo_step_po_plan_create, workspace of dsvgp_elbo_step_po_workspace_bytes; varn [B(p+1)] (device) receives
 * var + noise per output (what likelihood(model(x)).variance returns).  flags: bit 0 overlap, bit 1 include the KL term,
 * bit 8 (256) PLL objective (clear: ELBO).  Status word / jitter ladder as for dsvgp_elbo_step_f32.                            */
size_t dsvgp_elbo_step_po_workspace_bytes(int M, int d, int p, int B);
int dsvgp_elbo_step_po_plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, dsvgp_step_plan** out);
int dsvgp_elbo_step_po_f32(dsvgp_ctx* ctx, dsvgp_step_plan* plan, const dsvgp_elbo_step_io* io, float* varn, void* workspace,
                           size_t workspace_bytes, int flags);
/* Where an intermediate of the step queued last lies inside the caller's workspace (valid until the next step on it): which = 0:
 * [A ; mu_bar^T], A = L^-1 K_ZX (float [M'+1, B']); 1: K_ZX (float [M', B']); 2: the Cholesky factor L (double [M', M'], lower);
 * 3: L^-1 (double [M', M'], lower); 4: the constrained {lengthscale, outputscale, noise, 0} (float [1, 4]); 5: the factorisation's status word
 * (int32 [1], 0 = positive definite; valid from the step's factorisation until the next step on the workspace clears it).  The reference's every-50th-step nll print (directional_vi.py:255-260) reads the predictive
 * variance of the function-value rows of the forward pass it has just differentiated: W = L_S^T A[:, ::p+1] is all it takes.  */
int dsvgp_elbo_step_locate(const dsvgp_step_plan* plan, int which, size_t* offset_bytes, int* rows, int* cols, int64_t* ld);
/* ---- one RANK of a data-parallel job (the reference is single-process): the same step in five pieces with
 * the caller's collectives (RCCL through whatever binding the host uses) between them -- global-Gram schedule: the ranks sum
 * [tril(G) | b] instead of the L_S gradient, every rank forms the identical L_S-bar / m-bar itself, the replicated M'^3 stage
 * is sharded (columns of [Q' | a], rows of L-bar, column blocks of K_ZZ-bar) and met again by two all-gathers.
 *   plan      dsvgp_elbo_step_dp_plan_create(M, d, p, B = THIS RANK's rows, world); workspace of dsvgp_elbo_step_dp_workspace_bytes
 *   io        as above with global_rows = B'(global); io->flat is cleared in phase 0
 *   dp        rank / world and the collective operands (caller-owned device memory, float):
 *               wire       [wire_floats >= M'(M'+1)/2 + M'] packed [tril(G) | b]; anything beyond that count is the caller's
 *                          padding (keep it zero)             phase 0 writes it -> ALL-REDUCE(sum) -> phase 2 reads it
 *               q_local    [M', wq], wq = roundup4(ceil((M'+1)/world)), ZERO-INITIALISED once (the pad columns stay zero)
 *                                                              phase 1 writes it -> ALL-GATHER -> q_all [world, M', wq], phase 3 reads
 *               lbar_local [wr, M'], wr = roundup2(ceil(M'/world)), zero-initialised once
 *                                                              phase 2 writes it -> ALL-GATHER -> lbar_all [world wr, M'], phase 4 reads
 *   phase     0 .. 4 in this order, all on the context's stream (the collectives must be ordered against it by the caller);
 *             after phase 4: ALL-REDUCE(sum) of the slots of io->flat that hold Z-bar, V-bar, the hyper-parameter gradients and
 *             the loss; m-bar and L_S-bar are final and identical on every rank after phase 2 / 4 (do not reduce them).
 *   flags     as dsvgp_elbo_step_f32; bit 1 (value 2) on exactly ONE rank (it counts the KL value and the trace terms of the
 *             global Gram matrix; every rank adds the KL gradient).  Status word as above (dsvgp_elbo_step_status after phase 0).  */
typedef struct dsvgp_elbo_step_dp {
    int rank, world;
    float* wire; size_t wire_floats;
    float* q_local; const float* q_all;
    float* lbar_local; const float* lbar_all;
} dsvgp_elbo_step_dp;
size_t dsvgp_elbo_step_dp_workspace_bytes(int M, int d, int p, int B, int world);
int dsvgp_elbo_step_dp_plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, int world, dsvgp_step_plan** out);
int dsvgp_elbo_step_dp_f32(dsvgp_ctx* ctx, dsvgp_step_plan* plan, const dsvgp_elbo_step_io* io, const dsvgp_elbo_step_dp* dp,
                           void* workspace, size_t workspace_bytes, int flags, int phase);
/* flags & 4 in dsvgp_elbo_step_f32: HIP-event pairs around the forward solve, the K_ZX assembly and K_ZX-bar's kernel backward,
 * each on the stream its kernel runs on; ms3 = their durations in ms (waits for the step) -- bench.py's roofline entries      */
int dsvgp_elbo_step_timings(dsvgp_step_plan* plan, int steps_back, float* ms3);   /* the plan keeps the last 128 timed steps */
/* the same three durations plus ms5[3]: the Gram product [tril(G) ; b^T] = tril([A ; mu_bar^T] A^T) and ms5[4]: the dense product
 * K_ZX-bar = [Q' | a][A ; mu_bar^T] (the two fp32 [M', B'] products of the step; no counterpart in the reference, whose autograd runs
 * them inside DGVS.py:192-205's backward) */
int dsvgp_elbo_step_timings5(dsvgp_step_plan* plan, int steps_back, float* ms5);
long dsvgp_elbo_step_timed_count(const dsvgp_step_plan* plan);   /* steps queued with flag 4 so far (index of the last: count - 1) */

/* ---- measurement aid (bench.py `roofline.sustained`): the MFMA rate this card holds with no memory traffic, ~`millis` ms of
 * v_mfma_f64_16x16x4_f64 (is_double = 1) or v_mfma_f32_32x32x2_f32 (0) on every CU; synchronises the stream.
 * scratch: 2 MiB of device memory.  Not part of the reference's interface (SURVEY.md 8d asks for achieved-vs-peak; the
 * data-sheet peak is at 2.4 GHz, which the card does not hold under matrix load).                                        */
int dsvgp_mfma_rate(dsvgp_ctx* ctx, int is_double, int millis, void* scratch, double* tflops);
/* mode bit 0 = is_double, bit 1 = ONE wave per SIMD (one 256-thread workgroup per CU; four otherwise); burst_tflops (may be NULL): the rate
 * of the very first ~2 ms launch from an idle card, before the power management lowers the clock; clock_ghz (may be NULL): the in-kernel clock
 * of the last launch (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups).                                            */
int dsvgp_mfma_rate2(dsvgp_ctx* ctx, int mode, int millis, void* scratch, double* tflops, double* burst_tflops, double* clock_ghz);

#ifdef __cplusplus
}
#endif
#endif /* DSVGP_H */
