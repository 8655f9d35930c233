// Contour-integral-quadrature whitening for CiqDirectionalGradVariationalStrategy (reference
// directionalvi/CiqDirectionalGradVariationalStrategy.py:255-256: lazify(K_ZZ).sqrt_inv_matmul(K_ZX)), gfx950.
//
// gpytorch 1.4.0 evaluates K^{-1/2} R = sum_q omega_q (K + sigma_q I)^-1 R with ONE Lanczos process per right-hand side
// shared by all Q shifts (msMINRES).  Layout here: every Krylov object is stored "one right-hand side per ROW"
// ([t, n], t = B(p+1) right-hand sides, n = M(p+1)): the per-RHS reductions of the Lanczos step are row reductions
// (one workgroup per row, coalesced) and the shared product with the symmetric K is the row-major MFMA GEMM
// [t, n] x [n, n] of gemm.hip.
//
// The Lanczos BASIS stays resident in HBM ([J + 1, t, n]: 75 MB per iteration at the C5 size, a few GB of the 288) and the
// per-shift MINRES iterates are never stored: Paige-Saunders' recurrences  w_j = (v_j - eps_j w_{j-2} - delta_j w_{j-1}) / gamma_j,
// x_j = x_{j-1} + phi_j w_j  say  V = W R  (R upper triangular with three bands) and  x_J = W phi = V (R^-1 phi),  so each
// (shift, row) only carries its O(J) rotation scalars; y = R^-1 phi is a back-substitution in fp64 per (shift, row), and
// everything that needs vectors is a MIX of basis rows with per-row coefficients (ciq_mix_kernel: the convergence test's
// |w_J| and |x_J|, the output sum_q omega_q x_q, and the factors of the backward's sum_q omega_q Y_q^T X_q).  The
// per-iteration HBM traffic drops from 5 passes over [Q, t, n] (the w / x updates of every shift) to the Lanczos step's
// 4 passes over [t, n].
//   per iteration:  Vt = q_j K (gemm.hip)  ->  ciq_lanczos_kernel (alpha, beta, q_{j+1} into the next basis slot)
//                   ciq_givens_kernel (rotations for every (shift, row): Q*t scalars, kept for all iterations)
//   per test:       ciq_backsub_kernel (y, z = R^-1 phi, R^-1 e_J)  ->  ciq_norms_kernel (|V z|, |V y| per (shift, row))
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
// qnext = v / beta'  (qnext may be qprev: every element is read and written by the same thread; qprev == nullptr on the
// first step).  The row lives in LDS between the passes.
__global__ __launch_bounds__(256) void ciq_lanczos_kernel(const float* __restrict__ V, const float* __restrict__ qcur,
                                                          const float* qprev, float* qnext, const float* __restrict__ beta,
                                                          int n, float* __restrict__ alpha_out,
                                                          float* __restrict__ beta_out) {
    extern __shared__ float row[];
    __shared__ double red[4];
    const int j = blockIdx.x;
    const int64_t o = (int64_t)j * n;
    const float b = qprev ? beta[j] : 0.f;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = qprev ? V[o + i] - b * qprev[o + i] : V[o + i];
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
    for (int i = threadIdx.x; i < n; i += 256) qnext[o + i] = row[i] * inv;
}

// y = K x for ONE vector (the Ritz-bound Lanczos run): one wave per row of the symmetric row-major K, 16-byte loads along the
// row -- HBM-bound (the GEMM path spends a 128-row tile on the single row: 115 us at n = 6144 against ~35 here)
__global__ __launch_bounds__(256) void ciq_symv_kernel(const float* __restrict__ K, int64_t ldk, const float* __restrict__ x,
                                                       int n, float* __restrict__ y, int vec4) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* k = K + (int64_t)row * ldk;
    double acc = 0.0;
    if (vec4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int j = lane * 4; j < n; j += 256) {
            const float4 kv = *(const float4*)(k + j), xv = *(const float4*)(x + j);
            a0 = fmaf(kv.x, xv.x, a0); a1 = fmaf(kv.y, xv.y, a1); a2 = fmaf(kv.z, xv.z, a2); a3 = fmaf(kv.w, xv.w, a3);
        }
        acc = ((double)a0 + a1) + ((double)a2 + a3);
    } else {
        float a0 = 0.f;
        for (int j = lane; j < n; j += 64) a0 = fmaf(k[j], x[j], a0);
        acc = a0;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) y[row] = (float)acc;
}

// Paige-Saunders rotations for every (shift q, row j).  state[5][Q*t] = cs, sn, dbar, eps, phibar;
// coef[4][Q*t] = oldeps, delta, 1/gamma, phi of this iteration (one slot of the history ciq_backsub_kernel reads).
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

// y = R^-1 phi and z = R^-1 e_J for every (shift q, row j) from the rotation scalars of iterations 1..J
// (hist[J][4][Q*t] = eps, delta, 1/gamma, phi as ciq_givens_kernel left them): R_{i,i} = gamma_i, R_{i-1,i} = delta_i,
// R_{i-2,i} = eps_i.  Back-substitution in fp64; tables ycoef / zcoef [t][ldj][QP] (row, iteration, shift).
__global__ void ciq_backsub_kernel(const float* __restrict__ hist, int J, int Q, int t, int ldj, int QP,
                                   float* __restrict__ ycoef, float* __restrict__ zcoef) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = Q * t;
    if (e >= N) return;
    const int q = e / t, row = e - q * t;
    const int64_t S = (int64_t)4 * N;
    double y1 = 0.0, y2 = 0.0, z1 = 0.0, z2 = 0.0;        // entries i+1, i+2
    double d1 = 0.0, e1 = 0.0, e2 = 0.0;                  // delta_{i+1}, eps_{i+1}, eps_{i+2}
    float* yo = ycoef + ((int64_t)row * ldj) * QP + q;
    float* zo = zcoef ? zcoef + ((int64_t)row * ldj) * QP + q : nullptr;
    for (int i = J - 1; i >= 0; --i) {
        const float* h = hist + (int64_t)i * S + e;
        const double eps = h[0], delta = h[N], ig = h[2 * N], phi = h[3 * N];
        const double y = (phi - d1 * y1 - e2 * y2) * ig;
        const double z = ((i == J - 1 ? 1.0 : 0.0) - d1 * z1 - e2 * z2) * ig;
        yo[(int64_t)i * QP] = (float)y;
        if (zo) zo[(int64_t)i * QP] = (float)z;
        y2 = y1; y1 = y; z2 = z1; z1 = z;
        e2 = e1; e1 = eps; d1 = delta;
    }
}

// |phi_J| |V z| / |V y| for every (shift, row): gpytorch's convergence statistic (its mean is compared with the
// tolerance).  One workgroup per row; 16 shifts per pass; the coefficients are wave-uniform (scalar loads).
template <int VEC>
__global__ __launch_bounds__(256) void ciq_norms_kernel(const float* __restrict__ basis, int64_t bstride, int J, int n,
                                                        const float* __restrict__ ycoef, const float* __restrict__ zcoef,
                                                        int ldj, int QP, const float* __restrict__ phi, int Q, int t,
                                                        float* __restrict__ ratio) {
    __shared__ double red[4];
    const int row = blockIdx.x;
    const float* b = basis + (int64_t)row * n;
    for (int q0 = 0; q0 < QP; q0 += 16) {
        const int G = min(4, (QP - q0) / 4);
        const float* yc = ycoef + ((int64_t)row * ldj) * QP + q0;
        const float* zc = zcoef + ((int64_t)row * ldj) * QP + q0;
        float sw[16], sx[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { sw[k] = 0.f; sx[k] = 0.f; }
        for (int i = threadIdx.x * VEC; i < n; i += 256 * VEC) {
            float w[16][VEC], x[16][VEC];
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int u = 0; u < VEC; ++u) { w[k][u] = 0.f; x[k][u] = 0.f; }
            for (int j = 0; j < J; ++j) {
                float v[VEC];
                if (VEC == 2) { const float2 vv = *(const float2*)(b + (int64_t)j * bstride + i); v[0] = vv.x; v[VEC - 1] = vv.y; }
                else v[0] = b[(int64_t)j * bstride + i];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < G) {
                        const float4 cy = *(const float4*)(yc + (int64_t)j * QP + 4 * g);
                        const float4 cz = *(const float4*)(zc + (int64_t)j * QP + 4 * g);
#pragma unroll
                        for (int u = 0; u < VEC; ++u) {
                            x[4 * g + 0][u] = fmaf(cy.x, v[u], x[4 * g + 0][u]); x[4 * g + 1][u] = fmaf(cy.y, v[u], x[4 * g + 1][u]);
                            x[4 * g + 2][u] = fmaf(cy.z, v[u], x[4 * g + 2][u]); x[4 * g + 3][u] = fmaf(cy.w, v[u], x[4 * g + 3][u]);
                            w[4 * g + 0][u] = fmaf(cz.x, v[u], w[4 * g + 0][u]); w[4 * g + 1][u] = fmaf(cz.y, v[u], w[4 * g + 1][u]);
                            w[4 * g + 2][u] = fmaf(cz.z, v[u], w[4 * g + 2][u]); w[4 * g + 3][u] = fmaf(cz.w, v[u], w[4 * g + 3][u]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int u = 0; u < VEC; ++u) { sw[k] = fmaf(w[k][u], w[k][u], sw[k]); sx[k] = fmaf(x[k][u], x[k][u], sx[k]); }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double aw = block_sum((double)sw[k], red);
            const double ax = block_sum((double)sx[k], red);
            const int q = q0 + k;
            if (threadIdx.x == 0 && q < Q)
                ratio[q * t + row] = (float)(fabs((double)phi[q * t + row]) * sqrt(aw) / fmax(sqrt(ax), 1e-30));
        }
    }
}

// out[k][row][:] = scale_row * sum_j C[row][j][k0 + k] basis[j][row][:]  for k < 4 G (stored for k0 + k < Kout): the
// one vector operation of the basis-resident scheme.  The coefficients are wave-uniform (scalar loads); J independent
// row loads per thread are in flight at once.
template <int G, int VEC>
__global__ __launch_bounds__(256) void ciq_mix_kernel(const float* __restrict__ basis, int64_t bstride, int J, int n,
                                                      const float* __restrict__ C, int64_t ldrow, int KP, int k0, int Kout,
                                                      const float* __restrict__ rowscale, float* __restrict__ out,
                                                      int64_t ldo, int64_t ostride) {
    const int row = blockIdx.y;
    const int col = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (col >= n) return;
    const float* c = C + (int64_t)row * ldrow + k0;
    const float* b = basis + (int64_t)row * n + col;
    float acc[4 * G][VEC];
#pragma unroll
    for (int k = 0; k < 4 * G; ++k)
#pragma unroll
        for (int u = 0; u < VEC; ++u) acc[k][u] = 0.f;
#pragma unroll 4
    for (int j = 0; j < J; ++j) {
        float v[VEC];
        if (VEC == 4) {
            const float4 vv = *(const float4*)(b + (int64_t)j * bstride);
            v[0] = vv.x; v[1 % VEC] = vv.y; v[2 % VEC] = vv.z; v[3 % VEC] = vv.w;
        } else {
            v[0] = b[(int64_t)j * bstride];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 cc = *(const float4*)(c + (int64_t)j * KP + 4 * g);
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                acc[4 * g + 0][u] = fmaf(cc.x, v[u], acc[4 * g + 0][u]);
                acc[4 * g + 1][u] = fmaf(cc.y, v[u], acc[4 * g + 1][u]);
                acc[4 * g + 2][u] = fmaf(cc.z, v[u], acc[4 * g + 2][u]);
                acc[4 * g + 3][u] = fmaf(cc.w, v[u], acc[4 * g + 3][u]);
            }
        }
    }
    const float sc = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int k = 0; k < 4 * G; ++k) {
        if (k0 + k < Kout) {
            float* o = out + (int64_t)(k0 + k) * ostride + (int64_t)row * ldo + col;
            if (VEC == 4) *(float4*)o = make_float4(sc * acc[k][0], sc * acc[k][1 % VEC], sc * acc[k][2 % VEC], sc * acc[k][3 % VEC]);
            else o[0] = sc * acc[k][0];
        }
    }
}

// cout[row][j][0] = sum_q omega_q y[row][j][q]  (coefficients of out = sum_q omega_q x_q; entries 1..3 stay zero)
__global__ void ciq_cout_kernel(const float* __restrict__ ycoef, const float* __restrict__ omega, int Q, int t, int J,
                                int ldj, int QP, float* __restrict__ cout) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t * J) return;
    const int row = e / J, j = e - row * J;
    const float* y = ycoef + ((int64_t)row * ldj + j) * QP;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) acc += (double)omega[q] * y[q];
    cout[((int64_t)row * ldj + j) * 4] = (float)acc;
}

// C[row][jb][ia] = rn_a[row] rn_b[row] sum_q omega_q ya[row][ia][q] yb[row][jb][q]: the per-row coefficients of
// sum_q omega_q A_q^T B_q = sum_ia basisA_ia^T (sum_jb C[.][jb][ia] basisB_jb)  for A_q = rn_a (basisA ya_q), B_q likewise
__global__ void ciq_cross_kernel(const float* __restrict__ ya, int Ja, int lda, const float* __restrict__ yb, int Jb,
                                 int ldb, int QP, const float* __restrict__ omega, int Q, int t,
                                 const float* __restrict__ rn_a, const float* __restrict__ rn_b, float* __restrict__ Cout,
                                 int KPa) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = Ja * Jb;
    if (e >= (int64_t)t * per) return;
    const int row = (int)(e / per), r = (int)(e - (int64_t)row * per);
    const int jb = r / Ja, ia = r - jb * Ja;
    const float* a = ya + ((int64_t)row * lda + ia) * QP;
    const float* b = yb + ((int64_t)row * ldb + jb) * QP;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) acc += (double)omega[q] * a[q] * b[q];
    Cout[((int64_t)row * Jb + jb) * KPa + ia] = (float)(acc * rn_a[row] * rn_b[row]);
}

template <int VEC>
int launch_mix(hipStream_t st, const float* basis, int64_t bstride, int J, int t, int n, const float* C, int64_t ldrow, int KP,
               int Kout, const float* rowscale, float* out, int64_t ldo, int64_t ostride) {
    const dim3 grid(cdiv(n, 256 * VEC), t);
    for (int k0 = 0; k0 < Kout; k0 += 16) {
        const int G = (min(Kout, k0 + 16) - k0 + 3) / 4;
#define MIX_CASE(g)                                                                                                         \
    case g:                                                                                                                 \
        hipLaunchKernelGGL((ciq_mix_kernel<g, VEC>), grid, dim3(256), 0, st, basis, bstride, J, n, C, ldrow, KP, k0, Kout,   \
                           rowscale, out, ldo, ostride);                                                                    \
        break;
        switch (G) { MIX_CASE(1) MIX_CASE(2) MIX_CASE(3) MIX_CASE(4) }
#undef MIX_CASE
        DSVGP_LAUNCH_CHECK();
    }
    return 0;
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

static inline int ciq_qp(int Q) { return 4 * ((Q + 3) / 4); }

extern "C" size_t dsvgp_ciq_workspace_bytes(int Q, int t, int n, int cap) {
    if (Q <= 0 || t <= 0 || n <= 0 || cap <= 0) return 0;
    // Vt [t,n]; alpha, beta(2) [t]; state[5], ratio [Q t]; hist [cap][4][Q t]; zcoef [t][cap][QP]; cout [t][cap][4]
    return sizeof(float) * ((size_t)t * n + (size_t)3 * t + (size_t)6 * Q * t + (size_t)cap * 4 * Q * t +
                            (size_t)t * cap * ciq_qp(Q) + (size_t)t * cap * 4) + 256;
}

// out[t, n] = sum_q omega_q (K + sigma_q I)^-1 R_j (fp32 msMINRES, basis-resident: see the header of this file).
// K[n, n] symmetric fp32 (ldk), R[t, n] (ldr).  basis[cap + 1][t][n] receives the Lanczos rows q_0 .. q_J, ycoef[t][cap][QP]
// (QP = Q rounded up to 4) the coefficients of the NORMALISED solves (x_q,row = rnorm[row] * sum_j ycoef[row][j][q] q_j,row),
// rnorm[t] the row norms of R.  Iterates in blocks of `check_every` until the mean update ratio drops below tol (one host
// read per block) or max_iter; returns the iteration count J in *iters_out, or DSVGP_ENOSPACE when `cap` iterations did
// not suffice (nothing useful is left in the outputs: call again with a larger basis).
extern "C" int dsvgp_ciq_solve(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* R, int64_t ldr, int t, int n,
                               const float* sigma, const float* omega, int Q, float tol, int max_iter, int check_every,
                               float* basis, int cap, float* ycoef, float* rnorm, float* out, int64_t ldo, void* workspace,
                               int* iters_out) {
    if (!ctx || !K || !R || !sigma || !omega || !basis || !ycoef || !rnorm || !out || !workspace || t <= 0 || n <= 0 ||
        Q <= 0 || cap <= 0 || ldk < n || ldr < n || ldo < n || max_iter < 1 || check_every < 1)
        return DSVGP_EINVAL;
    if ((size_t)n * sizeof(float) > 64 * 1024) return DSVGP_EINVAL;      // one Lanczos row must fit the LDS stage
    hipStream_t st = ctx->stream;
    const int QP = ciq_qp(Q);
    const size_t tn = (size_t)t * n, qt = (size_t)Q * t;
    float* Vt = (float*)workspace;
    float* alpha = Vt + tn;
    float* beta0 = alpha + t;
    float* beta1 = beta0 + t;
    float* state = beta1 + t;
    float* ratio = state + 5 * qt;
    float* hist = ratio + qt;
    float* zcoef = hist + (size_t)cap * 4 * qt;
    float* cout = zcoef + (size_t)t * cap * QP;
    hipError_t e;
    if ((e = hipMemsetAsync(ycoef, 0, sizeof(float) * (size_t)t * cap * QP, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(zcoef, 0, sizeof(float) * ((size_t)t * cap * QP + (size_t)t * cap * 4), st)) != hipSuccess)
        return 1000 + (int)e;                                            // (zcoef and cout are adjacent)
    hipLaunchKernelGGL(ciq_init_kernel, dim3(t), dim3(256), 0, st, R, ldr, t, n, basis, rnorm);
    DSVGP_LAUNCH_CHECK();
    {   // state: cs = -1, sn = 0, dbar = 0, eps = 0, phibar = 1 (unit right-hand sides)
        std::vector<float> h(5 * qt, 0.f);
        for (size_t i = 0; i < qt; ++i) { h[i] = -1.f; h[4 * qt + i] = 1.f; }
        if ((e = hipMemcpyAsync(state, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice, st)) != hipSuccess)
            return 1000 + (int)e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return 1000 + (int)e;      // h goes out of scope
    }
    const bool vec2 = n % 2 == 0 && ((uintptr_t)basis % 8) == 0;
    float* bprev = beta0;
    float* bnext = beta1;
    std::vector<float> hr(qt);
    int it = 0;
    bool done = false;
    while (it < max_iter) {
        if (it >= cap) return DSVGP_ENOSPACE;
        ++it;
        const float* qcur = basis + (size_t)(it - 1) * tn;
        GemmArgs g{};
        g.M = t; g.N = n; g.K = n; g.A = qcur; g.lda = n; g.B = K; g.ldb = ldk; g.C = Vt; g.ldc = n;
        g.alpha = 1.0; g.beta = 0.0; g.flags = 0; g.batch = 1; g.splitk = 1;
        g.slab = ctx->det_slab; g.slab_bytes = ctx->det_bytes;
        int rc = launch_gemm(st, 0, g);
        if (rc) return rc;
        hipLaunchKernelGGL(ciq_lanczos_kernel, dim3(t), dim3(256), sizeof(float) * n, st, Vt, qcur,
                           it >= 2 ? basis + (size_t)(it - 2) * tn : nullptr, basis + (size_t)it * tn, bprev, n, alpha, bnext);
        DSVGP_LAUNCH_CHECK();
        float* coef = hist + (size_t)(it - 1) * 4 * qt;
        hipLaunchKernelGGL(ciq_givens_kernel, dim3(cdiv((int64_t)qt, 256)), dim3(256), 0, st, alpha, bnext, sigma, Q, t,
                           state, coef);
        DSVGP_LAUNCH_CHECK();
        if (it % check_every == 0 || it == max_iter) {
            hipLaunchKernelGGL(ciq_backsub_kernel, dim3(cdiv((int64_t)qt, 256)), dim3(256), 0, st, hist, it, Q, t, cap, QP,
                               ycoef, zcoef);
            DSVGP_LAUNCH_CHECK();
            if (vec2)
                hipLaunchKernelGGL(ciq_norms_kernel<2>, dim3(t), dim3(256), 0, st, basis, (int64_t)tn, it, n, ycoef, zcoef, cap,
                                   QP, coef + 3 * qt, Q, t, ratio);
            else
                hipLaunchKernelGGL(ciq_norms_kernel<1>, dim3(t), dim3(256), 0, st, basis, (int64_t)tn, it, n, ycoef, zcoef, cap,
                                   QP, coef + 3 * qt, Q, t, ratio);
            DSVGP_LAUNCH_CHECK();
            if ((e = hipMemcpyAsync(hr.data(), ratio, sizeof(float) * hr.size(), hipMemcpyDeviceToHost, st)) != hipSuccess)
                return 1000 + (int)e;
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return 1000 + (int)e;
            double mean = 0.0;
            for (float v : hr) mean += v;
            mean /= (double)hr.size();
            if (mean < tol || it == max_iter) { done = true; break; }
        }
        float* tmp = bprev; bprev = bnext; bnext = tmp;
    }
    if (!done) return DSVGP_EINVAL;                                      // (not reached: the last iteration always tests)
    hipLaunchKernelGGL(ciq_cout_kernel, dim3(cdiv((int64_t)t * it, 256)), dim3(256), 0, st, ycoef, omega, Q, t, it, cap, QP,
                       cout);
    DSVGP_LAUNCH_CHECK();
    const bool vec4 = n % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)basis % 16) == 0 && ((uintptr_t)out % 16) == 0;
    int rc = vec4 ? launch_mix<4>(st, basis, (int64_t)tn, it, t, n, cout, (int64_t)cap * 4, 4, 1, rnorm, out, ldo, 0)
                  : launch_mix<1>(st, basis, (int64_t)tn, it, t, n, cout, (int64_t)cap * 4, 4, 1, rnorm, out, ldo, 0);
    if (rc) return rc;
    if (iters_out) *iters_out = it;
    return 0;
}

// C[t][Jb][KPa] (KPa = Ja rounded up to 4, zero-padded) = rn_a rn_b sum_q omega_q ya[.][ia][q] yb[.][jb][q] from the
// coefficient tables of two solves ([t][lda | ldb][QP] as dsvgp_ciq_solve leaves them)
extern "C" int dsvgp_ciq_cross(dsvgp_ctx* ctx, const float* ya, int Ja, int lda, const float* yb, int Jb, int ldb,
                               const float* omega, int Q, int t, const float* rn_a, const float* rn_b, float* Cout) {
    if (!ctx || !ya || !yb || !omega || !rn_a || !rn_b || !Cout || Ja <= 0 || Jb <= 0 || lda < Ja || ldb < Jb || Q <= 0 || t <= 0)
        return DSVGP_EINVAL;
    hipStream_t st = ctx->stream;
    const int KPa = ciq_qp(Ja);
    hipError_t e;
    if ((e = hipMemsetAsync(Cout, 0, sizeof(float) * (size_t)t * Jb * KPa, st)) != hipSuccess) return 1000 + (int)e;
    hipLaunchKernelGGL(ciq_cross_kernel, dim3(cdiv((int64_t)t * Ja * Jb, 256)), dim3(256), 0, st, ya, Ja, lda, yb, Jb, ldb,
                       ciq_qp(Q), omega, Q, t, rn_a, rn_b, Cout, KPa);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// out[k][row][:] = rowscale[row] * sum_{j < J} C[row][j][k] basis[j][row][:]  for k < Kout  (C[t][ldj][KP], KP a multiple of
// 4 >= Kout; rowscale may be null; out[Kout][t][ldo])
extern "C" int dsvgp_ciq_mix(dsvgp_ctx* ctx, const float* basis, int J, int t, int n, const float* C, int ldj, int KP,
                             int Kout, const float* rowscale, float* out, int64_t ldo) {
    if (!ctx || !basis || !C || !out || J <= 0 || t <= 0 || n <= 0 || ldj < J || KP % 4 || Kout <= 0 || Kout > KP || ldo < n)
        return DSVGP_EINVAL;
    if ((uintptr_t)C % 16) return DSVGP_EALIGN;
    const bool vec4 = n % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)basis % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const int64_t tn = (int64_t)t * n;
    return vec4 ? launch_mix<4>(ctx->stream, basis, tn, J, t, n, C, (int64_t)ldj * KP, KP, Kout, rowscale, out, ldo, (int64_t)t * ldo)
                : launch_mix<1>(ctx->stream, basis, tn, J, t, n, C, (int64_t)ldj * KP, KP, Kout, rowscale, out, ldo, (int64_t)t * ldo);
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
    const int vec4 = n % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)workspace % 16) == 0;
    for (int k = 0; k < iters; ++k) {
        hipLaunchKernelGGL(ciq_symv_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, K, ldk, qcur, n, V, vec4);   // V = K q (K symmetric)
        DSVGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(ciq_lanczos_kernel, dim3(1), dim3(256), sizeof(float) * n, st, V, qcur, (const float*)qprev, qprev,
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
