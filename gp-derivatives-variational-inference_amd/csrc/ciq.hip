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
//
// Every kernel and entry point is a template over the scalar type: float for the reference's default model, double for a
// model built under torch.set_default_dtype(torch.float64) (reference experiments/bunny/exp_bunny.py:66,78 runs
// use_ciq=True that way) -- the *_f64 entry points at the end of the file.  Dot products and norms accumulate in double
// in both.
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<float> { using v2 = float2; using v4 = float4; };
template <> struct Vec<double> { using v2 = double2; using v4 = double4; };
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ float tfma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double tfma<double>(double a, double b, double c) { return fma(a, b, c); }

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
template <typename T>
__global__ __launch_bounds__(256) void ciq_init_kernel(const T* __restrict__ R, int64_t ldr, int t, int n,
                                                       T* __restrict__ q, T* __restrict__ rnorm) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const T* r = R + (int64_t)j * ldr;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)r[i] * r[i];
    T nr = (T)sqrt(block_sum(acc, red));
    if (nr < (T)1e-10) nr = (T)1;
    if (threadIdx.x == 0) rnorm[j] = nr;
    const T inv = (T)1 / nr;
    for (int i = threadIdx.x; i < n; i += 256) q[(int64_t)j * n + i] = r[i] * inv;
}

// One Lanczos step for every row j:  v = V_j - beta_j qprev_j;  alpha = q_j . v;  v -= alpha q_j;  beta' = |v|;
// qnext = v / beta'  (qnext may be qprev: every element is read and written by the same thread; qprev == nullptr on the
// first step).  The row lives in LDS between the passes.
template <typename T>
__global__ __launch_bounds__(256) void ciq_lanczos_kernel(const T* __restrict__ V, const T* __restrict__ qcur,
                                                          const T* qprev, T* qnext, const T* __restrict__ beta,
                                                          int n, T* __restrict__ alpha_out,
                                                          T* __restrict__ beta_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char row_raw[];
    T* row = reinterpret_cast<T*>(row_raw);
    __shared__ double red[4];
    const int j = blockIdx.x;
    const int64_t o = (int64_t)j * n;
    const T b = qprev ? beta[j] : (T)0;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const T v = qprev ? V[o + i] - b * qprev[o + i] : V[o + i];
        row[i] = v;
        acc += (double)qcur[o + i] * v;
    }
    const T a = (T)block_sum(acc, red);
    acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const T v = row[i] - a * qcur[o + i];
        row[i] = v;
        acc += (double)v * v;
    }
    const T bn = (T)sqrt(block_sum(acc, red));
    if (threadIdx.x == 0) { alpha_out[j] = a; beta_out[j] = bn; }
    const T inv = (T)1 / tmax(bn, (T)1e-30);
    for (int i = threadIdx.x; i < n; i += 256) qnext[o + i] = row[i] * inv;
}

// y = K x for ONE vector (the Ritz-bound Lanczos run): one wave per row of the symmetric row-major K, 16-byte loads along the
// row -- HBM-bound (the GEMM path spends a 128-row tile on the single row: 115 us at n = 6144 against ~35 here)
template <typename T>
__global__ __launch_bounds__(256) void ciq_symv_kernel(const T* __restrict__ K, int64_t ldk, const T* __restrict__ x,
                                                       int n, T* __restrict__ y, int vec4) {
    using T4 = typename Vec<T>::v4;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const T* k = K + (int64_t)row * ldk;
    double acc = 0.0;
    if (vec4) {
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int j = lane * 4; j < n; j += 256) {
            const T4 kv = *(const T4*)(k + j), xv = *(const T4*)(x + j);
            a0 = tfma<T>(kv.x, xv.x, a0); a1 = tfma<T>(kv.y, xv.y, a1); a2 = tfma<T>(kv.z, xv.z, a2); a3 = tfma<T>(kv.w, xv.w, a3);
        }
        acc = ((double)a0 + a1) + ((double)a2 + a3);
    } else {
        T a0 = 0;
        for (int j = lane; j < n; j += 64) a0 = tfma<T>(k[j], x[j], a0);
        acc = a0;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) y[row] = (T)acc;
}

// Paige-Saunders rotations for every (shift q, row j).  state[5][Q*t] = cs, sn, dbar, eps, phibar;
// coef[4][Q*t] = oldeps, delta, 1/gamma, phi of this iteration (one slot of the history ciq_backsub_kernel reads).
template <typename T>
__global__ void ciq_givens_kernel(const T* __restrict__ alpha, const T* __restrict__ beta_next,
                                  const T* __restrict__ sigma, int Q, int t, T* __restrict__ state,
                                  T* __restrict__ coef) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = Q * t;
    if (e >= N) return;
    const int q = e / t, j = e - q * t;
    const T cs = state[e], sn = state[N + e], dbar = state[2 * N + e], eps = state[3 * N + e], phibar = state[4 * N + e];
    const T alfa = alpha[j] + sigma[q], bn = beta_next[j];
    const T delta = cs * dbar + sn * alfa;
    const T gbar = sn * dbar - cs * alfa;
    const T gamma = tmax((T)sqrt(gbar * gbar + bn * bn), (T)1e-30);
    const T cs2 = gbar / gamma, sn2 = bn / gamma;
    coef[e] = eps;                    // oldeps
    coef[N + e] = delta;
    coef[2 * N + e] = (T)1 / gamma;
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
template <typename T>
__global__ void ciq_backsub_kernel(const T* __restrict__ hist, int J, int Q, int t, int ldj, int QP,
                                   T* __restrict__ ycoef, T* __restrict__ zcoef) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = Q * t;
    if (e >= N) return;
    const int q = e / t, row = e - q * t;
    const int64_t S = (int64_t)4 * N;
    double y1 = 0.0, y2 = 0.0, z1 = 0.0, z2 = 0.0;        // entries i+1, i+2
    double d1 = 0.0, e1 = 0.0, e2 = 0.0;                  // delta_{i+1}, eps_{i+1}, eps_{i+2}
    T* yo = ycoef + ((int64_t)row * ldj) * QP + q;
    T* zo = zcoef ? zcoef + ((int64_t)row * ldj) * QP + q : nullptr;
    for (int i = J - 1; i >= 0; --i) {
        const T* h = hist + (int64_t)i * S + e;
        const double eps = h[0], delta = h[N], ig = h[2 * N], phi = h[3 * N];
        const double y = (phi - d1 * y1 - e2 * y2) * ig;
        const double z = ((i == J - 1 ? 1.0 : 0.0) - d1 * z1 - e2 * z2) * ig;
        yo[(int64_t)i * QP] = (T)y;
        if (zo) zo[(int64_t)i * QP] = (T)z;
        y2 = y1; y1 = y; z2 = z1; z1 = z;
        e2 = e1; e1 = eps; d1 = delta;
    }
}

// |phi_J| |V z| / |V y| for every (shift, row): gpytorch's convergence statistic (its mean is compared with the
// tolerance).  One workgroup per row; 16 shifts per pass; the coefficients are wave-uniform (scalar loads).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ciq_norms_kernel(const T* __restrict__ basis, int64_t bstride, int J, int n,
                                                        const T* __restrict__ ycoef, const T* __restrict__ zcoef,
                                                        int ldj, int QP, const T* __restrict__ phi, int Q, int t,
                                                        T* __restrict__ ratio) {
    using T2 = typename Vec<T>::v2;
    using T4 = typename Vec<T>::v4;
    __shared__ double red[4];
    const int row = blockIdx.x;
    const T* b = basis + (int64_t)row * n;
    for (int q0 = 0; q0 < QP; q0 += 16) {
        const int G = min(4, (QP - q0) / 4);
        const T* yc = ycoef + ((int64_t)row * ldj) * QP + q0;
        const T* zc = zcoef + ((int64_t)row * ldj) * QP + q0;
        T sw[16], sx[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { sw[k] = 0; sx[k] = 0; }
        for (int i = threadIdx.x * VEC; i < n; i += 256 * VEC) {
            T w[16][VEC], x[16][VEC];
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int u = 0; u < VEC; ++u) { w[k][u] = 0; x[k][u] = 0; }
            for (int j = 0; j < J; ++j) {
                T v[VEC];
                if (VEC == 2) { const T2 vv = *(const T2*)(b + (int64_t)j * bstride + i); v[0] = vv.x; v[VEC - 1] = vv.y; }
                else v[0] = b[(int64_t)j * bstride + i];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < G) {
                        const T4 cy = *(const T4*)(yc + (int64_t)j * QP + 4 * g);
                        const T4 cz = *(const T4*)(zc + (int64_t)j * QP + 4 * g);
#pragma unroll
                        for (int u = 0; u < VEC; ++u) {
                            x[4 * g + 0][u] = tfma<T>(cy.x, v[u], x[4 * g + 0][u]); x[4 * g + 1][u] = tfma<T>(cy.y, v[u], x[4 * g + 1][u]);
                            x[4 * g + 2][u] = tfma<T>(cy.z, v[u], x[4 * g + 2][u]); x[4 * g + 3][u] = tfma<T>(cy.w, v[u], x[4 * g + 3][u]);
                            w[4 * g + 0][u] = tfma<T>(cz.x, v[u], w[4 * g + 0][u]); w[4 * g + 1][u] = tfma<T>(cz.y, v[u], w[4 * g + 1][u]);
                            w[4 * g + 2][u] = tfma<T>(cz.z, v[u], w[4 * g + 2][u]); w[4 * g + 3][u] = tfma<T>(cz.w, v[u], w[4 * g + 3][u]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int u = 0; u < VEC; ++u) { sw[k] = tfma<T>(w[k][u], w[k][u], sw[k]); sx[k] = tfma<T>(x[k][u], x[k][u], sx[k]); }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double aw = block_sum((double)sw[k], red);
            const double ax = block_sum((double)sx[k], red);
            const int q = q0 + k;
            if (threadIdx.x == 0 && q < Q)
                ratio[q * t + row] = (T)(fabs((double)phi[q * t + row]) * sqrt(aw) / fmax(sqrt(ax), 1e-30));
        }
    }
}

// out[k][row][:] = scale_row * sum_j C[row][j][k0 + k] basis[j][row][:]  for k < 4 G (stored for k0 + k < Kout): the
// one vector operation of the basis-resident scheme.  The coefficients are wave-uniform (scalar loads); J independent
// row loads per thread are in flight at once.  VEC = 4 (float), 2 (double) or 1: 16 bytes per lane and basis row.
template <typename T, int G, int VEC>
__global__ __launch_bounds__(256) void ciq_mix_kernel(const T* __restrict__ basis, int64_t bstride, int J, int n,
                                                      const T* __restrict__ C, int64_t ldrow, int KP, int k0, int Kout,
                                                      const T* __restrict__ rowscale, T* __restrict__ out,
                                                      int64_t ldo, int64_t ostride) {
    using T2 = typename Vec<T>::v2;
    using T4 = typename Vec<T>::v4;
    const int row = blockIdx.y;
    const int col = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (col >= n) return;
    const T* c = C + (int64_t)row * ldrow + k0;
    const T* b = basis + (int64_t)row * n + col;
    T acc[4 * G][VEC];
#pragma unroll
    for (int k = 0; k < 4 * G; ++k)
#pragma unroll
        for (int u = 0; u < VEC; ++u) acc[k][u] = 0;
#pragma unroll 4
    for (int j = 0; j < J; ++j) {
        T v[VEC];
        if (VEC == 4) {
            const T4 vv = *(const T4*)(b + (int64_t)j * bstride);
            v[0] = vv.x; v[1 % VEC] = vv.y; v[2 % VEC] = vv.z; v[3 % VEC] = vv.w;
        } else if (VEC == 2) {
            const T2 vv = *(const T2*)(b + (int64_t)j * bstride);
            v[0] = vv.x; v[1 % VEC] = vv.y;
        } else {
            v[0] = b[(int64_t)j * bstride];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const T4 cc = *(const T4*)(c + (int64_t)j * KP + 4 * g);
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                acc[4 * g + 0][u] = tfma<T>(cc.x, v[u], acc[4 * g + 0][u]);
                acc[4 * g + 1][u] = tfma<T>(cc.y, v[u], acc[4 * g + 1][u]);
                acc[4 * g + 2][u] = tfma<T>(cc.z, v[u], acc[4 * g + 2][u]);
                acc[4 * g + 3][u] = tfma<T>(cc.w, v[u], acc[4 * g + 3][u]);
            }
        }
    }
    const T sc = rowscale ? rowscale[row] : (T)1;
#pragma unroll
    for (int k = 0; k < 4 * G; ++k) {
        if (k0 + k < Kout) {
            T* o = out + (int64_t)(k0 + k) * ostride + (int64_t)row * ldo + col;
            if (VEC == 4) { T4 r; r.x = sc * acc[k][0]; r.y = sc * acc[k][1 % VEC]; r.z = sc * acc[k][2 % VEC]; r.w = sc * acc[k][3 % VEC]; *(T4*)o = r; }
            else if (VEC == 2) { T2 r; r.x = sc * acc[k][0]; r.y = sc * acc[k][1 % VEC]; *(T2*)o = r; }
            else o[0] = sc * acc[k][0];
        }
    }
}

// cout[row][j][0] = sum_q omega_q y[row][j][q]  (coefficients of out = sum_q omega_q x_q; entries 1..3 stay zero)
template <typename T>
__global__ void ciq_cout_kernel(const T* __restrict__ ycoef, const T* __restrict__ omega, int Q, int t, int J,
                                int ldj, int QP, T* __restrict__ cout) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t * J) return;
    const int row = e / J, j = e - row * J;
    const T* y = ycoef + ((int64_t)row * ldj + j) * QP;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) acc += (double)omega[q] * y[q];
    cout[((int64_t)row * ldj + j) * 4] = (T)acc;
}

// C[row][jb][ia] = rn_a[row] rn_b[row] sum_q omega_q ya[row][ia][q] yb[row][jb][q]: the per-row coefficients of
// sum_q omega_q A_q^T B_q = sum_ia basisA_ia^T (sum_jb C[.][jb][ia] basisB_jb)  for A_q = rn_a (basisA ya_q), B_q likewise
template <typename T>
__global__ void ciq_cross_kernel(const T* __restrict__ ya, int Ja, int lda, const T* __restrict__ yb, int Jb,
                                 int ldb, int QP, const T* __restrict__ omega, int Q, int t,
                                 const T* __restrict__ rn_a, const T* __restrict__ rn_b, T* __restrict__ Cout,
                                 int KPa) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = Ja * Jb;
    if (e >= (int64_t)t * per) return;
    const int row = (int)(e / per), r = (int)(e - (int64_t)row * per);
    const int jb = r / Ja, ia = r - jb * Ja;
    const T* a = ya + ((int64_t)row * lda + ia) * QP;
    const T* b = yb + ((int64_t)row * ldb + jb) * QP;
    double acc = 0.0;
    for (int q = 0; q < Q; ++q) acc += (double)omega[q] * a[q] * b[q];
    Cout[((int64_t)row * Jb + jb) * KPa + ia] = (T)(acc * rn_a[row] * rn_b[row]);
}

template <typename T, int VEC>
int launch_mix(hipStream_t st, const T* basis, int64_t bstride, int J, int t, int n, const T* C, int64_t ldrow, int KP,
               int Kout, const T* rowscale, T* out, int64_t ldo, int64_t ostride) {
    const dim3 grid(cdiv(n, 256 * VEC), t);
    for (int k0 = 0; k0 < Kout; k0 += 16) {
        const int G = (min(Kout, k0 + 16) - k0 + 3) / 4;
#define MIX_CASE(g)                                                                                                         \
    case g:                                                                                                                 \
        hipLaunchKernelGGL((ciq_mix_kernel<T, g, VEC>), grid, dim3(256), 0, st, basis, bstride, J, n, C, ldrow, KP, k0, Kout, \
                           rowscale, out, ldo, ostride);                                                                    \
        break;
        switch (G) { MIX_CASE(1) MIX_CASE(2) MIX_CASE(3) MIX_CASE(4) }
#undef MIX_CASE
        DSVGP_LAUNCH_CHECK();
    }
    return 0;
}
// 16 bytes per lane and basis row when the alignment allows it: four floats / two doubles
template <typename T>
int launch_mix_any(hipStream_t st, const T* basis, int64_t bstride, int J, int t, int n, const T* C, int64_t ldrow, int KP,
                   int Kout, const T* rowscale, T* out, int64_t ldo, int64_t ostride) {
    constexpr int W = 16 / (int)sizeof(T);
    const bool wide = n % W == 0 && ldo % W == 0 && ((uintptr_t)basis % 16) == 0 && ((uintptr_t)out % 16) == 0;
    return wide ? launch_mix<T, W>(st, basis, bstride, J, t, n, C, ldrow, KP, Kout, rowscale, out, ldo, ostride)
                : launch_mix<T, 1>(st, basis, bstride, J, t, n, C, ldrow, KP, Kout, rowscale, out, ldo, ostride);
}

// ---- _NgdInterpTerms pieces (reference CiqDirectionalGradVariationalStrategy.py:65-69,265-266) in the row layout ----
// per row j of T [t, n]: imean = T_j . m, ivar = (ST)_j . T_j, tsq = |T_j|^2;
// mu = imean + c, var = max(s dg_j - tsq + ivar, 1e-6), live = var not clamped
template <typename T>
__global__ __launch_bounds__(256) void ciq_rowstats_kernel(const T* __restrict__ Tm, const T* __restrict__ ST,
                                                           const T* __restrict__ m, const T* __restrict__ constant,
                                                           const T* __restrict__ hyp, int p, int n, T kxx_jitter,
                                                           T* __restrict__ imean, T* __restrict__ mu,
                                                           T* __restrict__ var, T* __restrict__ live) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const int64_t o = (int64_t)j * n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const T tv = Tm[o + i];
        a0 += (double)tv * m[i];
        a1 += (double)ST[o + i] * tv;
        a2 += (double)tv * tv;
    }
    a0 = block_sum(a0, red);
    a1 = block_sum(a1, red);
    a2 = block_sum(a2, red);
    if (threadIdx.x == 0) {
        const T ell = hyp[0], s = hyp[1];
        const T dg = (j % (p + 1) == 0) ? s : s / (ell * ell);
        const T v = (T)((double)dg + (double)kxx_jitter - a2 + a1);      // (data_data_covar.add_jitter(1e-4) of gpytorch's plain CIQ strategy)
        imean[j] = (T)a0;
        mu[j] = (T)a0 + constant[0];
        var[j] = tmax(v, (T)1e-6);
        live[j] = v > (T)1e-6 ? (T)1 : (T)0;
    }
}

// Tbar = 2 vbar (ST - T) + mubar m^T  (:94-96 plus the -sum T^2 term of :265);  VT = vbar T (left factor of d eta_2, :115);
// cvec = mubar - 2 vbar imean (coefficients of d eta_1, :102-107)
template <typename T>
__global__ __launch_bounds__(256) void ciq_tbar_kernel(const T* __restrict__ Tm, const T* __restrict__ ST,
                                                       const T* __restrict__ m, const T* __restrict__ mu_bar,
                                                       const T* __restrict__ var_bar, const T* __restrict__ live,
                                                       const T* __restrict__ imean, int t, int n,
                                                       T* __restrict__ Tbar, T* __restrict__ VT,
                                                       T* __restrict__ cvec) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const T vb = var_bar[j] * live[j], mb = mu_bar[j];
    if (i == 0) cvec[j] = mb - (T)2 * vb * imean[j];
    if (i >= n) return;
    const int64_t o = (int64_t)j * n + i;
    const T tv = Tm[o];
    Tbar[o] = (T)2 * vb * (ST[o] - tv) + mb * m[i];
    VT[o] = vb * tv;
}

// out = (A + A^T) / 2 (square, out != A)
template <typename T>
__global__ void sym_average_kernel(const T* __restrict__ A, int n, int64_t lda, T* __restrict__ out, int64_t ldo) {
    __shared__ T tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x, tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8) {
        const int gi = bj * 32 + r, gj = bi * 32 + tx;          // element (gi, gj) of the transposed block
        tile[r][tx] = (gi < n && gj < n) ? A[(int64_t)gi * lda + gj] : (T)0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int gi = bi * 32 + r, gj = bj * 32 + tx;
        if (gi < n && gj < n) out[(int64_t)gi * ldo + gj] = (T)0.5 * (A[(int64_t)gi * lda + gj] + tile[tx][r]);
    }
}

inline int ciq_qp(int Q) { return 4 * ((Q + 3) / 4); }

template <typename T>
size_t ciq_workspace_bytes(int Q, int t, int n, int cap) {
    if (Q <= 0 || t <= 0 || n <= 0 || cap <= 0) return 0;
    // Vt [t,n]; alpha, beta(2) [t]; state[5], ratio [Q t]; hist [cap][4][Q t]; zcoef [t][cap][QP]; cout [t][cap][4]
    return sizeof(T) * ((size_t)t * n + (size_t)3 * t + (size_t)6 * Q * t + (size_t)cap * 4 * Q * t +
                        (size_t)t * cap * ciq_qp(Q) + (size_t)t * cap * 4) + 256;
}

// out[t, n] = sum_q omega_q (K + sigma_q I)^-1 R_j (msMINRES, basis-resident: see the header of this file).
// K[n, n] symmetric (ldk), R[t, n] (ldr).  basis[cap + 1][t][n] receives the Lanczos rows q_0 .. q_J, ycoef[t][cap][QP]
// (QP = Q rounded up to 4) the coefficients of the NORMALISED solves (x_q,row = rnorm[row] * sum_j ycoef[row][j][q] q_j,row),
// rnorm[t] the row norms of R.  Iterates in blocks of `check_every` until the mean update ratio drops below tol (one host
// read per block) or max_iter; returns the iteration count J in *iters_out, or DSVGP_ENOSPACE when `cap` iterations did
// not suffice (nothing useful is left in the outputs: call again with a larger basis).
template <typename T>
int ciq_solve(dsvgp_ctx* ctx, const T* K, int64_t ldk, const T* R, int64_t ldr, int t, int n, const T* sigma, const T* omega,
              int Q, double tol, int max_iter, int check_every, T* basis, int cap, T* ycoef, T* rnorm, T* out, int64_t ldo,
              void* workspace, int* iters_out) {
    if (!ctx || !K || !R || !sigma || !omega || !basis || !ycoef || !rnorm || !out || !workspace || t <= 0 || n <= 0 ||
        Q <= 0 || cap <= 0 || ldk < n || ldr < n || ldo < n || max_iter < 1 || check_every < 1)
        return DSVGP_EINVAL;
    if ((size_t)n * sizeof(T) > 64 * 1024) return DSVGP_EINVAL;      // one Lanczos row must fit the LDS stage
    constexpr int is_double = std::is_same<T, double>::value ? 1 : 0;
    hipStream_t st = ctx->stream;
    const int QP = ciq_qp(Q);
    const size_t tn = (size_t)t * n, qt = (size_t)Q * t;
    T* Vt = (T*)workspace;
    T* alpha = Vt + tn;
    T* beta0 = alpha + t;
    T* beta1 = beta0 + t;
    T* state = beta1 + t;
    T* ratio = state + 5 * qt;
    T* hist = ratio + qt;
    T* zcoef = hist + (size_t)cap * 4 * qt;
    T* cout = zcoef + (size_t)t * cap * QP;
    hipError_t e;
    if ((e = hipMemsetAsync(ycoef, 0, sizeof(T) * (size_t)t * cap * QP, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(zcoef, 0, sizeof(T) * ((size_t)t * cap * QP + (size_t)t * cap * 4), st)) != hipSuccess)
        return 1000 + (int)e;                                            // (zcoef and cout are adjacent)
    hipLaunchKernelGGL(ciq_init_kernel<T>, dim3(t), dim3(256), 0, st, R, ldr, t, n, basis, rnorm);
    DSVGP_LAUNCH_CHECK();
    {   // state: cs = -1, sn = 0, dbar = 0, eps = 0, phibar = 1 (unit right-hand sides)
        std::vector<T> h(5 * qt, (T)0);
        for (size_t i = 0; i < qt; ++i) { h[i] = (T)-1; h[4 * qt + i] = (T)1; }
        if ((e = hipMemcpyAsync(state, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, st)) != hipSuccess)
            return 1000 + (int)e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return 1000 + (int)e;      // h goes out of scope
    }
    // (float: two floats per lane; double: one -- 32 double accumulators per lane already fill the register file's share)
    const bool vec2 = !is_double && n % 2 == 0 && ((uintptr_t)basis % 8) == 0;
    T* bprev = beta0;
    T* bnext = beta1;
    std::vector<T> hr(qt);
    int it = 0;
    bool done = false;
    while (it < max_iter) {
        if (it >= cap) return DSVGP_ENOSPACE;
        ++it;
        const T* qcur = basis + (size_t)(it - 1) * tn;
        GemmArgs g{};
        g.M = t; g.N = n; g.K = n; g.A = qcur; g.lda = n; g.B = K; g.ldb = ldk; g.C = Vt; g.ldc = n;
        g.alpha = 1.0; g.beta = 0.0; g.flags = 0; g.batch = 1; g.splitk = 1;
        g.slab = ctx->det_slab; g.slab_bytes = ctx->det_bytes;
        int rc = launch_gemm(st, is_double, g);
        if (rc) return rc;
        hipLaunchKernelGGL(ciq_lanczos_kernel<T>, dim3(t), dim3(256), sizeof(T) * n, st, (const T*)Vt, qcur,
                           it >= 2 ? (const T*)(basis + (size_t)(it - 2) * tn) : (const T*)nullptr, basis + (size_t)it * tn,
                           (const T*)bprev, n, alpha, bnext);
        DSVGP_LAUNCH_CHECK();
        T* coef = hist + (size_t)(it - 1) * 4 * qt;
        hipLaunchKernelGGL(ciq_givens_kernel<T>, dim3(cdiv((int64_t)qt, 256)), dim3(256), 0, st, (const T*)alpha, (const T*)bnext,
                           sigma, Q, t, state, coef);
        DSVGP_LAUNCH_CHECK();
        if (it % check_every == 0 || it == max_iter) {
            hipLaunchKernelGGL(ciq_backsub_kernel<T>, dim3(cdiv((int64_t)qt, 256)), dim3(256), 0, st, (const T*)hist, it, Q, t, cap,
                               QP, ycoef, zcoef);
            DSVGP_LAUNCH_CHECK();
            if (vec2)
                hipLaunchKernelGGL((ciq_norms_kernel<T, 2>), dim3(t), dim3(256), 0, st, (const T*)basis, (int64_t)tn, it, n,
                                   (const T*)ycoef, (const T*)zcoef, cap, QP, (const T*)(coef + 3 * qt), Q, t, ratio);
            else
                hipLaunchKernelGGL((ciq_norms_kernel<T, 1>), dim3(t), dim3(256), 0, st, (const T*)basis, (int64_t)tn, it, n,
                                   (const T*)ycoef, (const T*)zcoef, cap, QP, (const T*)(coef + 3 * qt), Q, t, ratio);
            DSVGP_LAUNCH_CHECK();
            if ((e = hipMemcpyAsync(hr.data(), ratio, sizeof(T) * hr.size(), hipMemcpyDeviceToHost, st)) != hipSuccess)
                return 1000 + (int)e;
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return 1000 + (int)e;
            double mean = 0.0;
            for (T v : hr) mean += v;
            mean /= (double)hr.size();
            if (mean < tol || it == max_iter) { done = true; break; }
        }
        T* tmp = bprev; bprev = bnext; bnext = tmp;
    }
    if (!done) return DSVGP_EINVAL;                                      // (not reached: the last iteration always tests)
    hipLaunchKernelGGL(ciq_cout_kernel<T>, dim3(cdiv((int64_t)t * it, 256)), dim3(256), 0, st, (const T*)ycoef, omega, Q, t, it,
                       cap, QP, cout);
    DSVGP_LAUNCH_CHECK();
    int rc = launch_mix_any<T>(st, basis, (int64_t)tn, it, t, n, cout, (int64_t)cap * 4, 4, 1, rnorm, out, ldo, 0);
    if (rc) return rc;
    if (iters_out) *iters_out = it;
    return 0;
}

// C[t][Jb][KPa] (KPa = Ja rounded up to 4, zero-padded) = rn_a rn_b sum_q omega_q ya[.][ia][q] yb[.][jb][q] from the
// coefficient tables of two solves ([t][lda | ldb][QP] as the solve leaves them)
template <typename T>
int ciq_cross(dsvgp_ctx* ctx, const T* ya, int Ja, int lda, const T* yb, int Jb, int ldb, const T* omega, int Q, int t,
              const T* rn_a, const T* rn_b, T* Cout) {
    if (!ctx || !ya || !yb || !omega || !rn_a || !rn_b || !Cout || Ja <= 0 || Jb <= 0 || lda < Ja || ldb < Jb || Q <= 0 || t <= 0)
        return DSVGP_EINVAL;
    hipStream_t st = ctx->stream;
    const int KPa = ciq_qp(Ja);
    hipError_t e;
    if ((e = hipMemsetAsync(Cout, 0, sizeof(T) * (size_t)t * Jb * KPa, st)) != hipSuccess) return 1000 + (int)e;
    hipLaunchKernelGGL(ciq_cross_kernel<T>, dim3(cdiv((int64_t)t * Ja * Jb, 256)), dim3(256), 0, st, ya, Ja, lda, yb, Jb, ldb,
                       ciq_qp(Q), omega, Q, t, rn_a, rn_b, Cout, KPa);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// out[k][row][:] = rowscale[row] * sum_{j < J} C[row][j][k] basis[j][row][:]  for k < Kout  (C[t][ldj][KP], KP a multiple of
// 4 >= Kout; rowscale may be null; out[Kout][t][ldo])
template <typename T>
int ciq_mix(dsvgp_ctx* ctx, const T* basis, int J, int t, int n, const T* C, int ldj, int KP, int Kout, const T* rowscale,
            T* out, int64_t ldo) {
    if (!ctx || !basis || !C || !out || J <= 0 || t <= 0 || n <= 0 || ldj < J || KP % 4 || Kout <= 0 || Kout > KP || ldo < n)
        return DSVGP_EINVAL;
    if ((uintptr_t)C % (4 * sizeof(T))) return DSVGP_EALIGN;
    const int64_t tn = (int64_t)t * n;
    return launch_mix_any<T>(ctx->stream, basis, tn, J, t, n, C, (int64_t)ldj * KP, KP, Kout, rowscale, out, ldo, (int64_t)t * ldo);
}

// `iters` Lanczos steps from the row v0[n]: alpha[iters], beta[iters] (beta[k] couples steps k and k+1) for the Ritz-value
// estimate of the spectrum's ends (contour_integral_quad's linear_cg(n_tridiag=1), max_lanczos_iter = 20).
template <typename T>
int ciq_lanczos(dsvgp_ctx* ctx, const T* K, int64_t ldk, const T* v0, int n, int iters, T* alpha, T* beta, void* workspace) {
    if (!ctx || !K || !v0 || !alpha || !beta || !workspace || n <= 0 || iters <= 0 || ldk < n) return DSVGP_EINVAL;
    if ((size_t)n * sizeof(T) > 64 * 1024) return DSVGP_EINVAL;
    hipStream_t st = ctx->stream;
    T* qa = (T*)workspace;
    T* qb = qa + n;
    T* V = qb + n;
    T* b0 = V + n;
    T* rn = b0 + 1;
    hipError_t e;
    if ((e = hipMemsetAsync(qb, 0, sizeof(T) * n, st)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipMemsetAsync(b0, 0, sizeof(T), st)) != hipSuccess) return 1000 + (int)e;
    hipLaunchKernelGGL(ciq_init_kernel<T>, dim3(1), dim3(256), 0, st, v0, (int64_t)n, 1, n, qa, rn);
    DSVGP_LAUNCH_CHECK();
    T* qcur = qa;
    T* qprev = qb;
    const int vec4 = n % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)K % (4 * sizeof(T))) == 0 && ((uintptr_t)workspace % (4 * sizeof(T))) == 0;
    for (int k = 0; k < iters; ++k) {
        hipLaunchKernelGGL(ciq_symv_kernel<T>, dim3(cdiv(n, 4)), dim3(256), 0, st, K, ldk, (const T*)qcur, n, V, vec4);   // V = K q (K symmetric)
        DSVGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(ciq_lanczos_kernel<T>, dim3(1), dim3(256), sizeof(T) * n, st, (const T*)V, (const T*)qcur, (const T*)qprev,
                           qprev, (const T*)(k == 0 ? b0 : beta + (k - 1)), n, alpha + k, beta + k);
        DSVGP_LAUNCH_CHECK();
        T* tmp = qcur; qcur = qprev; qprev = tmp;
    }
    return 0;
}

template <typename T>
int ciq_rowstats(dsvgp_ctx* ctx, const T* Tm, const T* ST, int t, int n, int p, const T* m, const T* constant, const T* hyp,
                 double kxx_jitter, T* imean, T* mu, T* var, T* live) {
    if (!ctx || !Tm || !ST || !m || !constant || !hyp || !imean || !mu || !var || !live || t <= 0 || n <= 0 || p < 0)
        return DSVGP_EINVAL;
    hipLaunchKernelGGL(ciq_rowstats_kernel<T>, dim3(t), dim3(256), 0, ctx->stream, Tm, ST, m, constant, hyp, p, n, (T)kxx_jitter,
                       imean, mu, var, live);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int ciq_tbar(dsvgp_ctx* ctx, const T* Tm, const T* ST, int t, int n, const T* m, const T* mu_bar, const T* var_bar, const T* live,
             const T* imean, T* Tbar, T* VT, T* cvec) {
    if (!ctx || !Tm || !ST || !m || !mu_bar || !var_bar || !live || !imean || !Tbar || !VT || !cvec || t <= 0 || n <= 0)
        return DSVGP_EINVAL;
    hipLaunchKernelGGL(ciq_tbar_kernel<T>, dim3(cdiv(n, 256), t), dim3(256), 0, ctx->stream, Tm, ST, m, mu_bar, var_bar, live,
                       imean, t, n, Tbar, VT, cvec);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int sym_average(dsvgp_ctx* ctx, const T* A, int n, int64_t lda, T* out, int64_t ldo) {
    if (!ctx || !A || !out || A == out || n <= 0 || lda < n || ldo < n) return DSVGP_EINVAL;
    const int nb = cdiv(n, 32);
    hipLaunchKernelGGL(sym_average_kernel<T>, dim3(nb, nb), dim3(32, 8), 0, ctx->stream, A, n, lda, out, ldo);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ---- C ABI: float (the reference's default model) ----
extern "C" size_t dsvgp_ciq_workspace_bytes(int Q, int t, int n, int cap) { return ciq_workspace_bytes<float>(Q, t, n, cap); }
extern "C" int dsvgp_ciq_solve(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* R, int64_t ldr, int t, int n,
                               const float* sigma, const float* omega, int Q, float tol, int max_iter, int check_every,
                               float* basis, int cap, float* ycoef, float* rnorm, float* out, int64_t ldo, void* workspace,
                               int* iters_out) {
    return ciq_solve<float>(ctx, K, ldk, R, ldr, t, n, sigma, omega, Q, (double)tol, max_iter, check_every, basis, cap, ycoef, rnorm,
                            out, ldo, workspace, iters_out);
}
extern "C" int dsvgp_ciq_cross(dsvgp_ctx* ctx, const float* ya, int Ja, int lda, const float* yb, int Jb, int ldb,
                               const float* omega, int Q, int t, const float* rn_a, const float* rn_b, float* Cout) {
    return ciq_cross<float>(ctx, ya, Ja, lda, yb, Jb, ldb, omega, Q, t, rn_a, rn_b, Cout);
}
extern "C" int dsvgp_ciq_mix(dsvgp_ctx* ctx, const float* basis, int J, int t, int n, const float* C, int ldj, int KP,
                             int Kout, const float* rowscale, float* out, int64_t ldo) {
    return ciq_mix<float>(ctx, basis, J, t, n, C, ldj, KP, Kout, rowscale, out, ldo);
}
extern "C" int dsvgp_ciq_lanczos(dsvgp_ctx* ctx, const float* K, int64_t ldk, const float* v0, int n, int iters,
                                 float* alpha, float* beta, void* workspace) {
    return ciq_lanczos<float>(ctx, K, ldk, v0, n, iters, alpha, beta, workspace);
}
extern "C" int dsvgp_ciq_rowstats(dsvgp_ctx* ctx, const float* T, const float* ST, int t, int n, int p, const float* m,
                                  const float* constant, const float* hyp, float kxx_jitter, float* imean, float* mu, float* var,
                                  float* live) {
    return ciq_rowstats<float>(ctx, T, ST, t, n, p, m, constant, hyp, (double)kxx_jitter, imean, mu, var, live);
}
extern "C" int dsvgp_ciq_tbar(dsvgp_ctx* ctx, const float* T, const float* ST, int t, int n, const float* m,
                              const float* mu_bar, const float* var_bar, const float* live, const float* imean,
                              float* Tbar, float* VT, float* cvec) {
    return ciq_tbar<float>(ctx, T, ST, t, n, m, mu_bar, var_bar, live, imean, Tbar, VT, cvec);
}
extern "C" int dsvgp_sym_average_f32(dsvgp_ctx* ctx, const float* A, int n, int64_t lda, float* out, int64_t ldo) {
    return sym_average<float>(ctx, A, n, lda, out, ldo);
}

// ---- C ABI: double (a model built under torch.set_default_dtype(torch.float64): the fp64 model mode's CIQ strategy) ----
extern "C" size_t dsvgp_ciq_workspace_bytes_f64(int Q, int t, int n, int cap) { return ciq_workspace_bytes<double>(Q, t, n, cap); }
extern "C" int dsvgp_ciq_solve_f64(dsvgp_ctx* ctx, const double* K, int64_t ldk, const double* R, int64_t ldr, int t, int n,
                                   const double* sigma, const double* omega, int Q, double tol, int max_iter, int check_every,
                                   double* basis, int cap, double* ycoef, double* rnorm, double* out, int64_t ldo,
                                   void* workspace, int* iters_out) {
    return ciq_solve<double>(ctx, K, ldk, R, ldr, t, n, sigma, omega, Q, tol, max_iter, check_every, basis, cap, ycoef, rnorm, out,
                             ldo, workspace, iters_out);
}
extern "C" int dsvgp_ciq_cross_f64(dsvgp_ctx* ctx, const double* ya, int Ja, int lda, const double* yb, int Jb, int ldb,
                                   const double* omega, int Q, int t, const double* rn_a, const double* rn_b, double* Cout) {
    return ciq_cross<double>(ctx, ya, Ja, lda, yb, Jb, ldb, omega, Q, t, rn_a, rn_b, Cout);
}
extern "C" int dsvgp_ciq_mix_f64(dsvgp_ctx* ctx, const double* basis, int J, int t, int n, const double* C, int ldj, int KP,
                                 int Kout, const double* rowscale, double* out, int64_t ldo) {
    return ciq_mix<double>(ctx, basis, J, t, n, C, ldj, KP, Kout, rowscale, out, ldo);
}
extern "C" int dsvgp_ciq_lanczos_f64(dsvgp_ctx* ctx, const double* K, int64_t ldk, const double* v0, int n, int iters,
                                     double* alpha, double* beta, void* workspace) {
    return ciq_lanczos<double>(ctx, K, ldk, v0, n, iters, alpha, beta, workspace);
}
extern "C" int dsvgp_ciq_rowstats_f64(dsvgp_ctx* ctx, const double* T, const double* ST, int t, int n, int p, const double* m,
                                      const double* constant, const double* hyp, double kxx_jitter, double* imean, double* mu,
                                      double* var, double* live) {
    return ciq_rowstats<double>(ctx, T, ST, t, n, p, m, constant, hyp, kxx_jitter, imean, mu, var, live);
}
extern "C" int dsvgp_ciq_tbar_f64(dsvgp_ctx* ctx, const double* T, const double* ST, int t, int n, const double* m,
                                  const double* mu_bar, const double* var_bar, const double* live, const double* imean,
                                  double* Tbar, double* VT, double* cvec) {
    return ciq_tbar<double>(ctx, T, ST, t, n, m, mu_bar, var_bar, live, imean, Tbar, VT, cvec);
}
extern "C" int dsvgp_sym_average_f64(dsvgp_ctx* ctx, const double* A, int n, int64_t lda, double* out, int64_t ldo) {
    return sym_average<double>(ctx, A, n, lda, out, ldo);
}
