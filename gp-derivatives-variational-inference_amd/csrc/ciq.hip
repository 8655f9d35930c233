// Contour-integral-quadrature whitening for CiqDirectionalGradVariationalStrategy (reference
// directionalvi/CiqDirectionalGradVariationalStrategy.py:255-256: lazify(K_ZZ).sqrt_inv_matmul(K_ZX)), gfx950.
//
// gpytorch 1.4.0 evaluates K^{-1/2} R = sum_q omega_q (K + sigma_q I)^-1 R with ONE Lanczos process per right-hand side
// shared by all Q shifts (msMINRES).  Layout here: every Krylov object is stored "one right-hand side per ROW"
// ([t, n], t = B(p+1) right-hand sides, n = M(p+1)): the per-RHS reductions of the Lanczos step are row reductions
// (one workgroup per row, coalesced), the shared product with the symmetric K is the row-major MFMA GEMM
// [t, n] x [n, n] of gemm.hip, and the per-shift recurrences stream [Q, t, n] arrays with unit stride.
//   per iteration:  V = Qcur K (gemm.hip)  ->  ciq_lanczos_kernel (alpha, beta, next Lanczos row)
//                   ciq_givens_kernel (Paige-Saunders rotations for every (shift, row): Q*t scalars)
//                   ciq_update_kernel (w, x updates for all shifts: the HBM-bound part, 5 passes over [Q, t, n])
#include <vector>

#include "common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// row norms of R and the first Lanczos rows: q = R / |R| (rows with |R| < 1e-10 are divided by 1, like gpytorch's minres)
__global__ __launch_bounds__(256) void ciq_init_kernel(const float* __restrict__ R, int64_t ldr, int t, int n,
                                                       float* __restrict__ q, float* __restrict__ rnorm) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const float* r = R + (int64_t)j * ldr;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)r[i] * r[i];
    float nr = (float)sqrt(block_sum(acc, red));
    if (nr < 1e-10f) nr = 1.f;
    if (threadIdx.x == 0) rnorm[j] = nr;
    const float inv = 1.f / nr;
    for (int i = threadIdx.x; i < n; i += 256) q[(int64_t)j * n + i] = r[i] * inv;
}

// One Lanczos step for every row j:  v = V_j - beta_j qprev_j;  alpha = q_j . v;  v -= alpha q_j;  beta' = |v|;
// qnext = v / beta'  (written over qprev).  The row lives in LDS between the passes.
__global__ __launch_bounds__(256) void ciq_lanczos_kernel(const float* __restrict__ V, const float* __restrict__ qcur,
                                                          float* __restrict__ qprev_next, const float* __restrict__ beta,
                                                          int n, float* __restrict__ alpha_out,
                                                          float* __restrict__ beta_out) {
    extern __shared__ float row[];
    __shared__ double red[4];
    const int j = blockIdx.x;
    const int64_t o = (int64_t)j * n;
    const float b = beta[j];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = V[o + i] - b * qprev_next[o + i];
        row[i] = v;
        acc += (double)qcur[o + i] * v;
    }
    const float a = (float)block_sum(acc, red);
    acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = row[i] - a * qcur[o + i];
        row[i] = v;
        acc += (double)v * v;
    }
    const float bn = (float)sqrt(block_sum(acc, red));
    if (threadIdx.x == 0) { alpha_out[j] = a; beta_out[j] = bn; }
    const float inv = 1.f / fmaxf(bn, 1e-30f);
    for (int i = threadIdx.x; i < n; i += 256) qprev_next[o + i] = row[i] * inv;
}

// Paige-Saunders rotations for every (shift q, row j).  state[5][Q*t] = cs, sn, dbar, eps, phibar;
// coef[4][Q*t] = oldeps, delta, 1/gamma, phi for ciq_update_kernel.
__global__ void ciq_givens_kernel(const float* __restrict__ alpha, const float* __restrict__ beta_next,
                                  const float* __restrict__ sigma, int Q, int t, float* __restrict__ state,
                                  float* __restrict__ coef) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = Q * t;
    if (e >= N) return;
    const int q = e / t, j = e - q * t;
    const float cs = state[e], sn = state[N + e], dbar = state[2 * N + e], eps = state[3 * N + e], phibar = state[4 * N + e];
    const float alfa = alpha[j] + sigma[q], bn = beta_next[j];
    const float delta = cs * dbar + sn * alfa;
    const float gbar = sn * dbar - cs * alfa;
    const float gamma = fmaxf(sqrtf(gbar * gbar + bn * bn), 1e-30f);
    const float cs2 = gbar / gamma, sn2 = bn / gamma;
    coef[e] = eps;                    // oldeps
    coef[N + e] = delta;
    coef[2 * N + e] = 1.f / gamma;
    coef[3 * N + e] = cs2 * phibar;   // phi
    state[e] = cs2;
    state[N + e] = sn2;
    state[2 * N + e] = -cs * bn;      // dbar
    state[3 * N + e] = sn * bn;       // eps
    state[4 * N + e] = sn2 * phibar;
}

// w = (v - oldeps w1 - delta w2) / gamma (written over w1, the oldest direction);  x += phi w   -- for all shifts.
__global__ __launch_bounds__(256) void ciq_update_kernel(const float* __restrict__ v, float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ x,
                                                         const float* __restrict__ coef, int Q, int t, int n) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int N = Q * t;
    const int64_t o = (int64_t)j * n + i, S = (int64_t)t * n;
    const float vv = v[o];
    for (int q = 0; q < Q; ++q) {
        const int e = q * t + j;
        const float w = (vv - coef[e] * w1[q * S + o] - coef[N + e] * w2[q * S + o]) * coef[2 * N + e];
        w1[q * S + o] = w;
        x[q * S + o] += coef[3 * N + e] * w;
    }
}

// |phi| |w| / |x| for every (shift, row): gpytorch's convergence statistic (its mean is compared with the tolerance)
__global__ __launch_bounds__(256) void ciq_conv_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                       const float* __restrict__ coef, int Q, int t, int n,
                                                       float* __restrict__ ratio) {
    __shared__ double red[4];
    const int e = blockIdx.x;                 // q * t + j
    const int64_t o = (int64_t)e * n;
    double aw = 0.0, ax = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { aw += (double)w[o + i] * w[o + i]; ax += (double)x[o + i] * x[o + i]; }
    aw = block_sum(aw, red);
    ax = block_sum(ax, red);
    if (threadIdx.x == 0) ratio[e] = (float)(fabs((double)coef[3 * Q * t + e]) * sqrt(aw) / fmax(sqrt(ax), 1e-30));
}

// x[q] *= rnorm_j (the solves of the un-normalised system) and out = sum_q omega_q x[q]
__global__ __launch_bounds__(256) void ciq_combine_kernel(float* __restrict__ x, const float* __restrict__ omega,
                                                          const float* __restrict__ rnorm, int Q, int t, int n,
                                                          float* __restrict__ out, int64_t ldo) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = (int64_t)j * n + i, S = (int64_t)t * n;
    const float rn = rnorm[j];
    float acc = 0.f;
    for (int q = 0; q < Q; ++q) {
        const float xv = x[q * S + o] * rn;
        x[q * S + o] = xv;
        acc = fmaf(omega[q], xv, acc);
    }
    out[(int64_t)j * ldo + i] = acc;
}

// ---- _NgdInterpTerms pieces (reference CiqDirectionalGradVariationalStrategy.py:65-69,265-266) in the row layout ----
// per row j of T [t, n]: imean = T_j . m, ivar = (ST)_j . T_j, tsq = |T_j|^2;
// mu = imean + c, var = max(s dg_j - tsq + ivar, 1e-6), live = var not clamped
__global__ __launch_bounds__(256) void ciq_rowstats_kernel(const float* __restrict__ T, const float* __restrict__ ST,
                                                           const float* __restrict__ m, const float* __restrict__ constant,
                                                           const float* __restrict__ hyp, int p, int n, float kxx_jitter,
                                                           float* __restrict__ imean, float* __restrict__ mu,
                                                           float* __restrict__ var, float* __restrict__ live) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const int64_t o = (int64_t)j * n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float tv = T[o + i];
        a0 += (double)tv * m[i];
        a1 += (double)ST[o + i] * tv;
        a2 += (double)tv * tv;
    }
    a0 = block_sum(a0, red);
    a1 = block_sum(a1, red);
    a2 = block_sum(a2, red);
    if (threadIdx.x == 0) {
        const float ell = hyp[0], s = hyp[1];
        const float dg = (j % (p + 1) == 0) ? s : s / (ell * ell);
        const float v = (float)((double)dg + (double)kxx_jitter - a2 + a1);      // (data_data_covar.add_jitter(1e-4) of gpytorch's plain CIQ strategy)
        imean[j] = (float)a0;
        mu[j] = (float)a0 + constant[0];
        var[j] = fmaxf(v, 1e-6f);
        live[j] = v > 1e-6f ? 1.f : 0.f;
    }
}

// Tbar = 2 vbar (ST - T) + mubar m^T  (:94-96 plus the -sum T^2 term of :265);  VT = vbar T (left factor of d eta_2, :115);
// cvec = mubar - 2 vbar imean (coefficients of d eta_1, :102-107)
__global__ __launch_bounds__(256) void ciq_tbar_kernel(const float* __restrict__ T, const float* __restrict__ ST,
                                                       const float* __restrict__ m, const float* __restrict__ mu_bar,
                                                       const float* __restrict__ var_bar, const float* __restrict__ live,
                                                       const float* __restrict__ imean, int t, int n,
                                                       float* __restrict__ Tbar, float* __restrict__ VT,
                                                       float* __restrict__ cvec) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float vb = var_bar[j] * live[j], mb = mu_bar[j];
    if (i == 0) cvec[j] = mb - 2.f * vb * imean[j];
    if (i >= n) return;
    const int64_t o = (int64_t)j * n + i;
    const float tv = T[o];
    Tbar[o] = 2.f * vb * (ST[o] - tv) + mb * m[i];
    VT[o] = vb * tv;
}

// out = (A + A^T) / 2 (square, fp32, out != A)
__global__ void sym_average_f32_kernel(const float* __restrict__ A, int n, int64_t lda, float* __restrict__ out,
                                       int64_t ldo) {
    __shared__ float tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x, tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8) {
        const int gi = bj * 32 + r, gj = bi * 32 + tx;          // element (gi, gj) of the transposed block
        tile[r][tx] = (gi < n && gj < n) ? A[(int64_t)gi * lda + gj] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int gi = bi * 32 + r, gj = bj * 32 + tx;
        if (gi < n && gj < n) out[(int64_t)gi * ldo + gj] = 0.5f * (A[(int64_t)gi * lda + gj] + tile[tx][r]);
    }
}

}  // namespace

extern "C" size_t dsvgp_ciq_workspace_bytes(int Q, int t, int n) {
    if (Q <= 0 || t <= 0 || n <= 0) return 0;
    // qa, qb, V [t,n]; w1, w2 [Q,t,n]; alpha, beta(2), rnorm [t]; state[5], coef[4], ratio [Q t]
    return sizeof(float) * ((size_t)3 * t * n + (size_t)2 * Q * t * n + (size_t)4 * t + (size_t)10 * Q * t) + 256;
}

// X[Q, t, n] = (K + sigma_q I)^-1 R_j for every shift / row, out[t, n] = sum_q omega_q X[q]  (fp32 msMINRES).
// K[n, n] symmetric fp32 (ldk), R[t, n] (ldr).  Iterates in blocks of `check_every` until the mean update ratio drops
// below tol (one host read per block) or max_iter; returns the iteration count in *iters_out.
extern "C" int dsvgp_ciq_solve(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* R, int64_t ldr, int t, int n,
                               const float* sigma, const float* omega, int Q, float tol, int max_iter, int check_every,
                               float* X, float* out, int64_t ldo, void* workspace, int* iters_out) {
    if (!ctx || !K || !R || !sigma || !omega || !X || !out || !workspace || t <= 0 || n <= 0 || Q <= 0 || ldk < n ||
        ldr < n || ldo < n || max_iter < 1 || check_every < 1)
        return DSVGP_EINVAL;
    if ((size_t)n * sizeof(float) > 64 * 1024) return DSVGP_EINVAL;      // one Lanczos row must fit the LDS stage
    hipStream_t st = ctx->stream;
    float* qa = (float*)workspace;
    float* qb = qa + (size_t)t * n;
    float* V = qb + (size_t)t * n;
    float* w1 = V + (size_t)t * n;
    float* w2 = w1 + (size_t)Q * t * n;
    float* alpha = w2 + (size_t)Q * t * n;
    float* beta0 = alpha + t;
    float* beta1 = beta0 + t;
    float* rnorm = beta1 + t;
    float* state = rnorm + t;
    float* coef = state + (size_t)5 * Q * t;
    float* ratio = coef + (size_t)4 * Q * t;
    const size_t qtn = (size_t)Q * t * n;
    hipError_t e;
    if ((e = hipMemsetAsync(qb, 0, sizeof(float) * (size_t)t * n, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(w1, 0, sizeof(float) * 2 * qtn, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(X, 0, sizeof(float) * qtn, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(beta0, 0, sizeof(float) * t, st)) != hipSuccess) return 1000 + (int)e;
    hipLaunchKernelGGL(ciq_init_kernel, dim3(t), dim3(256), 0, st, R, ldr, t, n, qa, rnorm);
    DSVGP_LAUNCH_CHECK();
    {   // state: cs = -1, sn = 0, dbar = 0, eps = 0, phibar = 1 (unit right-hand sides)
        std::vector<float> h((size_t)5 * Q * t, 0.f);
        for (int i = 0; i < Q * t; ++i) { h[i] = -1.f; h[(size_t)4 * Q * t + i] = 1.f; }
        if ((e = hipMemcpyAsync(state, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice, st)) != hipSuccess)
            return 1000 + (int)e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return 1000 + (int)e;      // h goes out of scope
    }
    float* qcur = qa;
    float* qprev = qb;      // becomes the next Lanczos block in place
    float* bprev = beta0;
    float* bnext = beta1;
    std::vector<float> hr((size_t)Q * t);
    int it = 0;
    while (it < max_iter) {
        ++it;
        GemmArgs g{};
        g.M = t; g.N = n; g.K = n; g.A = qcur; g.lda = n; g.B = K; g.ldb = ldk; g.C = V; g.ldc = n;
        g.alpha = 1.0; g.beta = 0.0; g.flags = 0; g.batch = 1; g.splitk = 1;
        int rc = launch_gemm(st, 0, g);
        if (rc) return rc;
        hipLaunchKernelGGL(ciq_lanczos_kernel, dim3(t), dim3(256), sizeof(float) * n, st, V, qcur, qprev, bprev, n, alpha,
                           bnext);
        DSVGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(ciq_givens_kernel, dim3(cdiv((int64_t)Q * t, 256)), dim3(256), 0, st, alpha, bnext, sigma, Q, t,
                           state, coef);
        DSVGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(ciq_update_kernel, dim3(cdiv(n, 256), t), dim3(256), 0, st, qcur, w1, w2, X, coef, Q, t, n);
        DSVGP_LAUNCH_CHECK();
        if (it % check_every == 0 || it == max_iter) {
            hipLaunchKernelGGL(ciq_conv_kernel, dim3(Q * t), dim3(256), 0, st, w1, X, coef, Q, t, n, ratio);
            DSVGP_LAUNCH_CHECK();
            if ((e = hipMemcpyAsync(hr.data(), ratio, sizeof(float) * hr.size(), hipMemcpyDeviceToHost, st)) != hipSuccess)
                return 1000 + (int)e;
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return 1000 + (int)e;
            double mean = 0.0;
            for (float v : hr) mean += v;
            mean /= (double)hr.size();
            if (mean < tol) break;
        }
        float* tmp = w1; w1 = w2; w2 = tmp;              // (w1, w2) <- (w2, w)
        tmp = qcur; qcur = qprev; qprev = tmp;           // (qprev, qcur) <- (qcur, qnext)
        tmp = bprev; bprev = bnext; bnext = tmp;
    }
    hipLaunchKernelGGL(ciq_combine_kernel, dim3(cdiv(n, 256), t), dim3(256), 0, st, X, omega, rnorm, Q, t, n, out, ldo);
    DSVGP_LAUNCH_CHECK();
    if (iters_out) *iters_out = it;
    return 0;
}

// `iters` Lanczos steps from the row v0[n]: alpha[iters], beta[iters] (beta[k] couples steps k and k+1) for the Ritz-value
// estimate of the spectrum's ends (contour_integral_quad's linear_cg(n_tridiag=1), max_lanczos_iter = 20).
extern "C" int dsvgp_ciq_lanczos(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* v0, int n, int iters,
                                 float* alpha, float* beta, void* workspace) {
    if (!ctx || !K || !v0 || !alpha || !beta || !workspace || n <= 0 || iters <= 0 || ldk < n) return DSVGP_EINVAL;
    if ((size_t)n * sizeof(float) > 64 * 1024) return DSVGP_EINVAL;
    hipStream_t st = ctx->stream;
    float* qa = (float*)workspace;
    float* qb = qa + n;
    float* V = qb + n;
    float* b0 = V + n;
    float* rn = b0 + 1;
    hipError_t e;
    if ((e = hipMemsetAsync(qb, 0, sizeof(float) * n, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(b0, 0, sizeof(float), st)) != hipSuccess) return 1000 + (int)e;
    hipLaunchKernelGGL(ciq_init_kernel, dim3(1), dim3(256), 0, st, v0, (int64_t)n, 1, n, qa, rn);
    DSVGP_LAUNCH_CHECK();
    float* qcur = qa;
    float* qprev = qb;
    for (int k = 0; k < iters; ++k) {
        GemmArgs g{};
        g.M = 1; g.N = n; g.K = n; g.A = qcur; g.lda = n; g.B = K; g.ldb = ldk; g.C = V; g.ldc = n;
        g.alpha = 1.0; g.beta = 0.0; g.flags = 0; g.batch = 1; g.splitk = 1;
        int rc = launch_gemm(st, 0, g);
        if (rc) return rc;
        hipLaunchKernelGGL(ciq_lanczos_kernel, dim3(1), dim3(256), sizeof(float) * n, st, V, qcur, qprev,
                           k == 0 ? b0 : beta + (k - 1), n, alpha + k, beta + k);
        DSVGP_LAUNCH_CHECK();
        float* tmp = qcur; qcur = qprev; qprev = tmp;
    }
    return 0;
}

extern "C" int dsvgp_ciq_rowstats(dsvgp_ctx* ctx, const float* T, const float* ST, int t, int n, int p, const float* m,
                                  const float* constant, const float* hyp, float kxx_jitter, float* imean, float* mu, float* var,
                                  float* live) {
    if (!ctx || !T || !ST || !m || !constant || !hyp || !imean || !mu || !var || !live || t <= 0 || n <= 0 || p < 0)
        return DSVGP_EINVAL;
    hipLaunchKernelGGL(ciq_rowstats_kernel, dim3(t), dim3(256), 0, ctx->stream, T, ST, m, constant, hyp, p, n, kxx_jitter,
                       imean, mu, var, live);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_ciq_tbar(dsvgp_ctx* ctx, const float* T, const float* ST, int t, int n, const float* m,
                              const float* mu_bar, const float* var_bar, const float* live, const float* imean,
                              float* Tbar, float* VT, float* cvec) {
    if (!ctx || !T || !ST || !m || !mu_bar || !var_bar || !live || !imean || !Tbar || !VT || !cvec || t <= 0 || n <= 0)
        return DSVGP_EINVAL;
    hipLaunchKernelGGL(ciq_tbar_kernel, dim3(cdiv(n, 256), t), dim3(256), 0, ctx->stream, T, ST, m, mu_bar, var_bar, live,
                       imean, t, n, Tbar, VT, cvec);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_sym_average_f32(dsvgp_ctx* ctx, const float* A, int n, int64_t lda, float* out, int64_t ldo) {
    if (!ctx || !A || !out || A == out || n <= 0 || lda < n || ldo < n) return DSVGP_EINVAL;
    const int nb = cdiv(n, 32);
    hipLaunchKernelGGL(sym_average_f32_kernel, dim3(nb, nb), dim3(32, 8), 0, ctx->stream, A, n, lda, out, ldo);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
