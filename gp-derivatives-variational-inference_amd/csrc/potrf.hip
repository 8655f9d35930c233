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

// Thread (tx = tid & 63, ty = tid >> 6) keeps A[4r + ty][tx] and X[4r + ty][tx], r = 0..15, in registers.
// Per column j ONE barrier: the owners publish column j of A and row j of X (both "current") through
// double-buffered LDS vectors, then every thread forms 1/sqrt(a_jj) (v_rsq_f64 + 2 Newton steps; no
// fp64 divide or sqrt on the serial chain) and applies the right-looking update to A and the forward
// elimination of [L | I] to X.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y + y * (0.5 * fma(-d * y, y, 1.0));
    y = y + y * (0.5 * fma(-d * y, y, 1.0));
    return y;
}

__global__ __launch_bounds__(256) void potf2_inv_kernel(double* __restrict__ A, int64_t lda, int r0, int nr,
                                                        double* __restrict__ Dinv, int* __restrict__ info) {
    __shared__ double colbuf[2][NBC];
    __shared__ double rowbuf[2][NBC];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    double a[16], x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 4 * r + ty;
        double v = (i == tx) ? 1.0 : 0.0;                       // identity padding of a ragged last block
        if (i < nr && tx < nr && tx <= i) v = A[(int64_t)(r0 + i) * lda + r0 + tx];
        a[r] = v;
        x[r] = (i == tx) ? 1.0 : 0.0;
    }
    for (int j = 0; j < NBC; ++j) {
        double* cb = colbuf[j & 1];
        double* rb = rowbuf[j & 1];
        if (tx == j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cb[4 * r + ty] = a[r];
        }
        if (ty == (j & 3)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r == (j >> 2)) rb[tx] = x[r];
        }
        __syncthreads();
        const double dj = cb[j];
        if (tid == 0 && !(dj > 0.0) && j < nr && *info == 0) *info = r0 + j + 1;   // LAPACK: leading minor not PD
        const double rinv = rsqrt_nr(dj);                       // 1 / L_jj
        const double xj = rb[tx] * rinv;                        // final X[j][tx]
        const double ck = cb[tx] * rinv;                        // L[tx][j]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (4 * r + 3 < j) continue;                        // wave-uniform: rows above j are final
            const int i = 4 * r + ty;
            const double lij = cb[i] * rinv;                    // L[i][j]  (i == j: sqrt(a_jj))
            if (i > j) {
                x[r] -= lij * xj;
                if (tx > j && i >= tx) a[r] -= lij * ck;
            } else if (i == j) {
                x[r] = xj;
            }
            if (tx == j && i >= j) a[r] = lij;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 4 * r + ty;
        if (i < nr && tx <= i) A[(int64_t)(r0 + i) * lda + r0 + tx] = a[r];
        Dinv[i * NBC + tx] = x[r];
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
        hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, st, A, lda, r0, nr, Dk, info);
        DSVGP_LAUNCH_CHECK();
        if (r1 >= n) break;
        // panel: L[r1:, r0:r1] = A[r1:, r0:r1] * inv(L_kk)^T      (in place: one n-tile per row panel)
        GemmArgs g{};
        g.batch = 1; g.splitk = 1;
        g.M = n - r1; g.N = nr; g.K = nr;
        g.A = A + (size_t)r1 * lda + r0; g.lda = lda;
        g.B = Dk; g.ldb = NBC;
        g.flags = DSVGP_GEMM_TRANS_B | DSVGP_GEMM_B_UPPER;     // op(B)[k][n] = Dinv[n][k], zero for k > n
        g.alpha = 1.0; g.beta = 0.0;
        g.C = A + (size_t)r1 * lda + r0; g.ldc = lda;
        int rc = launch_gemm(st, 1, g);
        if (rc) return rc;
        // trailing update: A[r1:, r1:] -= L_panel L_panel^T        (lower tiles only, in place)
        GemmArgs f{};
        f.batch = 1; f.splitk = 1;
        f.M = n - r1; f.N = n - r1; f.K = nr;
        f.A = A + (size_t)r1 * lda + r0; f.lda = lda;
        f.B = f.A; f.ldb = lda;
        f.flags = DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER | DSVGP_GEMM_KEEP_UPPER;
        f.alpha = -1.0; f.beta = 1.0;
        f.Cin = A + (size_t)r1 * lda + r1; f.ldcin = lda;
        f.C = A + (size_t)r1 * lda + r1; f.ldc = lda;
        rc = launch_gemm(st, 1, f);
        if (rc) return rc;
    }
    return 0;
}
