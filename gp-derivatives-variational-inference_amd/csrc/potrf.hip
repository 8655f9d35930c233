// Blocked right-looking Cholesky of the fp64 K_ZZ on the MFMA GEMM (algo 1 of dsvgp_potrf).
// reference: psd_safe_cholesky(K_ZZ.double()) -> torch.cholesky (DirectionalGradVariationalStrategy.py:72-75).
//
// rocSOLVER's dpotrf spends ~6 ms at M' = 3000 in serial single-workgroup panel kernels.  Here each
// 64 x 64 diagonal block is factored AND inverted by one workgroup out of LDS (potf2_inv_kernel), the
// panel below it is one triangular MFMA GEMM  L_panel = A_panel * inv(L_kk)^T  and the trailing update
// one MFMA GEMM  A_22 -= L_panel L_panel^T  (lower tiles only); 3 launches per block column.
#include "common.h"

namespace {

constexpr int NBC = 64;

// Thread (tx = tid & 63, ty = tid >> 6) keeps ONE register column w[r] <-> row NW*r + ty of column tx: it is
// column tx of A while j < tx (right-looking updates), is published and written back as column tx of L at
// j == tx, and from then on holds column tx of X = L^-1 (forward elimination of [L | I]) -- a lane never needs
// both at once.  Per column j ONE barrier: the owners publish column j of A and row j of X through
// double-buffered LDS vectors, every thread forms 1/sqrt(a_jj) (v_rsq_f64 + 2 Newton steps; no fp64 divide
// or sqrt on the serial chain) and updates its registers.  NW waves (NW*64 threads) share the 64 x 64 block.
#ifndef POTRF_NW
#define POTRF_NW 16
#endif
#ifndef POTRF_NEWTON
#define POTRF_NEWTON 2
#endif
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < POTRF_NEWTON; ++it) y = y + y * (0.5 * fma(-d * y, y, 1.0));
    return y;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void potf2_inv_kernel(double* __restrict__ A, int64_t lda, int r0, int nr,
                                                            double* __restrict__ Dinv, int* __restrict__ info) {
    constexpr int RPT = NBC / NW;            // rows per thread
    __shared__ double colbuf[2][NBC];
    __shared__ double rowbuf[2][NBC];
    __shared__ double Ls[NBC][NBC + 1];      // finished columns of L (no global store on the serial chain)
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
#ifdef POTRF_DEBUG
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
#endif
    double w[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = NW * r + ty;
        double v = (i == tx) ? 1.0 : 0.0;                       // identity padding of a ragged last block
        if (i < nr && tx < nr && tx <= i) v = A[(int64_t)(r0 + i) * lda + r0 + tx];
        w[r] = v;
    }
    for (int j = 0; j < NBC; ++j) {
        double* cb = colbuf[j & 1];
        double* rb = rowbuf[j & 1];
        if (tx == j) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) cb[NW * r + ty] = w[r];
        }
        if (ty == j % NW) {                                      // owner of row j publishes X[j][tx] (1 on the diagonal)
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) v = (r == j / NW) ? w[r] : v;
            rb[tx] = (tx == j) ? 1.0 : v;
        }
        __syncthreads();
        double cbv[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) cbv[r] = cb[NW * r + ty];
        const double dj = cb[j];
        const double rbv = rb[tx], cbt = cb[tx];
        if (tid == 0 && !(dj > 0.0) && j < nr && *info == 0) *info = r0 + j + 1;   // LAPACK: leading minor not PD
        const double rinv = rsqrt_nr(dj);                       // 1 / L_jj
        const bool right = tx > j, own = tx == j;
        const double f = right ? cbt * rinv : rbv * rinv;       // L[tx][j]  or  final X[j][tx]
        const int ilo = right ? tx : j + 1;                     // first row that takes the rank-1 update
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = NW * r + ty;
            const double lij = cbv[r] * rinv;                   // L[i][j]  (i == j: sqrt(a_jj))
            if (own) Ls[i][j] = lij;                            // column j of L is final
            const double base = own ? 0.0 : w[r];               // the X column starts from the identity
            double v = (i >= ilo) ? fma(-lij, f, base) : base;
            v = (!right && i == j) ? f : v;                     // row j of X
            w[r] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) Dinv[(NW * r + ty) * NBC + tx] = w[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = NW * r + ty;
        if (i < nr && tx <= i) A[(int64_t)(r0 + i) * lda + r0 + tx] = Ls[i][tx];
    }
#ifdef POTRF_DEBUG
    if (tid == 0) {   // tools only: shader cycles and 100 MHz real-time ticks of this workgroup, in the unused upper corner
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1r = __builtin_amdgcn_s_memrealtime();
        Dinv[62] = (double)(t1c - t0c);
        Dinv[63] = (double)(t1r - t0r);
    }
#endif
}

// Single-stage fp64 MFMA kernel for the K <= 64 products of the factorisation (panel solve against the
// inverted diagonal block, rank-64 trailing update): C[64x64 tile] = alpha * A[64 x K] * B[64 x K]^T + beta * C.
// Both operands are k-contiguous and fit LDS whole, so a tile costs ONE global round trip instead of the
// K/16 dependent stages of the general GEMM (which is latency-bound at K = 64: 21 / 53 us per launch).
// lower_only: tiles with tn > tm exit, elements with n > m are not stored.  In place (C == A) is safe: every
// workgroup stages its whole A tile before it writes.
__global__ __launch_bounds__(256) void smallk_gemm_kernel(const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb,
                                                          double* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                          double alpha, double beta, int lower_only, int b_upper) {
    __shared__ double As[64][65];
    __shared__ double Bs[64][65];
    const int tm = blockIdx.y, tn = blockIdx.x;
    if (lower_only && tn > tm) return;
    const int m0 = tm * 64, n0 = tn * 64, tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, k = e & 63;
        As[r][k] = (m0 + r < M && k < K) ? A[(int64_t)(m0 + r) * lda + k] : 0.0;
        double bv = 0.0;
        if (n0 + r < N && k < K && !(b_upper && k > n0 + r)) bv = B[(int64_t)(n0 + r) * ldb + k];   // op(B)[k][n] = B[n][k]
        Bs[r][k] = bv;
    }
    __syncthreads();
    using acc_t = double __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;   // wave -> 32 x 32 outputs
    acc_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 4) {
        const int kq = kk + (lane >> 4);
        double a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[wr * 32 + i * 16 + (lane & 15)][kq];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[wc * 32 + j * 16 + (lane & 15)][kq];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * r;      // f64 C/D layout
                const int n = n0 + wc * 32 + j * 16 + (lane & 15);
                if (m >= M || n >= N || (lower_only && n > m)) continue;
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v += beta * C[(int64_t)m * ldc + n];
                C[(int64_t)m * ldc + n] = v;
            }
}

}  // namespace

size_t potrf_blocked_workspace_bytes(int n) { return sizeof(double) * (size_t)cdiv(n, NBC) * NBC * NBC; }

int launch_potrf_blocked(hipStream_t st, double* A, int n, int64_t lda, int* info, double* dinv_ws) {
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int), st);
    if (e != hipSuccess) return 1000 + (int)e;
    const int nblk = cdiv(n, NBC);
    for (int k = 0; k < nblk; ++k) {
        const int r0 = k * NBC, nr = (n - r0 < NBC) ? (n - r0) : NBC, r1 = r0 + nr;
        double* Dk = dinv_ws + (size_t)k * NBC * NBC;
        hipLaunchKernelGGL(potf2_inv_kernel<POTRF_NW>, dim3(1), dim3(POTRF_NW * 64), 0, st, A, lda, r0, nr, Dk, info);
        DSVGP_LAUNCH_CHECK();
        if (r1 >= n) break;
        // panel: L[r1:, r0:r1] = A[r1:, r0:r1] * inv(L_kk)^T      (in place; op(B)[k][n] = Dinv[n][k], zero for k > n)
        const int Mr = n - r1;
        hipLaunchKernelGGL(smallk_gemm_kernel, dim3(1, cdiv(Mr, 64)), dim3(256), 0, st, A + (size_t)r1 * lda + r0, lda,
                           (const double*)Dk, (int64_t)NBC, A + (size_t)r1 * lda + r0, lda, Mr, nr, nr, 1.0, 0.0, 0, 1);
        DSVGP_LAUNCH_CHECK();
        // trailing update: A[r1:, r1:] -= L_panel L_panel^T        (lower tiles only, in place)
        hipLaunchKernelGGL(smallk_gemm_kernel, dim3(cdiv(Mr, 64), cdiv(Mr, 64)), dim3(256), 0, st,
                           A + (size_t)r1 * lda + r0, lda, A + (size_t)r1 * lda + r0, lda, A + (size_t)r1 * lda + r1, lda,
                           Mr, Mr, nr, -1.0, 1.0, 1, 0);
        DSVGP_LAUNCH_CHECK();
    }
    return 0;
}
