// MFMA GEMM for the DSVGP hot path (gfx950): C = alpha*op(A)op(B) + beta*Cin.
//
// One kernel family serves the panel triangular solve (L^-1 K_ZX and L^-T Abar, reference
// DirectionalGradVariationalStrategy.py:181,183 and their autograd backward), the variational
// products W = L_S^T A, U = L_S W (:192-205) and the M' x M' Gram-type contractions of the backward.
//
// Tiling (64-wide wavefronts): 128x128 output tile per 256-thread workgroup, 2x2 waves, each wave a
// 4x4 grid of 16x16 MFMA tiles (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, K=4 per
// instruction), BK=16 per LDS stage.  Operands are staged global->registers->LDS with the next
// stage's global loads issued before the current stage's MFMAs (latency hides under the matrix
// pipe; fp64 MFMA is 64 cycles/instruction so the loop is matrix-pipe bound by construction).
// LDS images: a k-contiguous operand is kept [mn][k] with row stride 17, an mn-contiguous one
// [k][mn] with row stride 144; both give conflict-free fragment reads for ds_read_b32/_b64.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int S_MN = 144;  // [k][mn] image: 144 = 128 + 16 -> the two k rows of a 32-lane half hit disjoint bank halves
constexpr int S_K = 17;    // [mn][k] image: odd stride

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    using acc_t = float __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};
template <> struct Mfma<double> {
    using acc_t = double __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout differs from the f32 one: row = (lane>>4) + 4*reg
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// tri: 0 none, 1 keep k <= mn, 2 keep k >= mn   (indices local to the operand)
template <typename TIn, typename TC, bool KC>
__device__ __forceinline__ void load_tile(TC (&r)[8], const TIn* __restrict__ p, int64_t ld, int mn0,
                                          int MN, int k0, int K, int tri, const float* __restrict__ kscale) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int mn, k;
        if (KC) { k = k0 + (t & 15); mn = mn0 + (t >> 4) + 16 * i; }
        else    { mn = mn0 + (t & 127); k = k0 + (t >> 7) * 8 + i; }
        bool ok = (mn < MN) && (k < K);
        if (tri == 1) ok = ok && (k <= mn);
        if (tri == 2) ok = ok && (k >= mn);
        TC v = TC(0);
        if (ok) {
            v = (TC)(KC ? p[(int64_t)mn * ld + k] : p[(int64_t)k * ld + mn]);
            if (kscale) v *= (TC)kscale[k];
        }
        r[i] = v;
    }
}

template <typename TC, bool KC>
__device__ __forceinline__ void store_tile(TC* __restrict__ s, const TC (&r)[8]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (KC) s[((t >> 4) + 16 * i) * S_K + (t & 15)] = r[i];
        else    s[((t >> 7) * 8 + i) * S_MN + (t & 127)] = r[i];
    }
}

template <typename TC, typename TB, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    using M = Mfma<TC>;
    using acc_t = typename M::acc_t;
    __shared__ TC As[AKC ? BM * S_K : BK * S_MN];
    __shared__ TC Bs[BKC ? BN * S_K : BK * S_MN];

    const int bz = blockIdx.z / g.splitk, sp = blockIdx.z % g.splitk;
    const bool last = (bz == g.batch - 1);
    const int Mdim = (last && g.M_last) ? g.M_last : g.M;
    const int Kdim = (last && g.K_last) ? g.K_last : g.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= Mdim) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const TC* A = (const TC*)g.A + (int64_t)bz * g.sA;
    const TB* B = (const TB*)g.B + (int64_t)bz * g.sB;
    TC* C = (TC*)g.C + (int64_t)bz * g.sC;
    const int fl = g.flags;
    const int triA = (fl & DSVGP_GEMM_A_LOWER) ? 1 : ((fl & DSVGP_GEMM_A_UPPER) ? 2 : 0);
    const int triB = (fl & DSVGP_GEMM_B_LOWER) ? 2 : ((fl & DSVGP_GEMM_B_UPPER) ? 1 : 0);
    const bool out_lower = fl & DSVGP_GEMM_OUT_LOWER;

    if (out_lower && n0 >= m0 + BM) {  // tile strictly above the diagonal: defined as zero
        if (g.splitk == 1 || sp == 0) {
            for (int e = threadIdx.x; e < BM * BN; e += 256) {
                int m = m0 + e / BN, n = n0 + e % BN;
                if (m < Mdim && n < g.N) {
                    if (g.splitk == 1) C[(int64_t)m * g.ldc + n] = TC(0);
                    if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = 0.f;
                }
            }
        }
        return;
    }

    // k range implied by the triangular structure, then the split-K slice of it
    int klo = 0, khi = Kdim;
    if (triA == 1) khi = min(khi, m0 + BM);
    if (triA == 2) klo = max(klo, (m0 / BK) * BK);
    if (triB == 2) klo = max(klo, (n0 / BK) * BK);
    if (triB == 1) khi = min(khi, n0 + BN);
    if (g.splitk > 1) {
        int steps = (khi - klo + BK - 1) / BK;
        int per = (steps + g.splitk - 1) / g.splitk;
        int lo = klo + sp * per * BK;
        khi = min(khi, lo + per * BK);
        klo = lo;
    }

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    if (klo < khi) {
        TC ra[8], rb[8];
        load_tile<TC, TC, AKC>(ra, A, g.lda, m0, Mdim, klo, Kdim, triA, g.kscale);
        load_tile<TB, TC, BKC>(rb, B, g.ldb, n0, g.N, klo, Kdim, triB, nullptr);
        for (int k0 = klo; k0 < khi; k0 += BK) {
            __syncthreads();
            store_tile<TC, AKC>(As, ra);
            store_tile<TC, BKC>(Bs, rb);
            __syncthreads();
            if (k0 + BK < khi) {
                load_tile<TC, TC, AKC>(ra, A, g.lda, m0, Mdim, k0 + BK, Kdim, triA, g.kscale);
                load_tile<TB, TC, BKC>(rb, B, g.ldb, n0, g.N, k0 + BK, Kdim, triB, nullptr);
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                TC a[4], b[4];
                const int kq = kk * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mm = wr * 64 + i * 16 + (lane & 15);
                    a[i] = AKC ? As[mm * S_K + kq] : As[kq * S_MN + mm];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nn = wc * 64 + j * 16 + (lane & 15);
                    b[j] = BKC ? Bs[nn * S_K + kq] : Bs[kq * S_MN + nn];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = M::mma(a[i], b[j], acc[i][j]);
            }
        }
    }

    const TC alpha = (TC)g.alpha, beta = (TC)g.beta;
    const bool cin_f = fl & DSVGP_GEMM_CIN_IS_FLOAT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 64 + i * 16 + M::row(lane, r);
                const int n = n0 + wc * 64 + j * 16 + (lane & 15);
                if (m >= Mdim || n >= g.N) continue;
                TC v = alpha * acc[i][j][r];
                if (g.splitk > 1) {
                    if (!(out_lower && n > m)) atomicAdd(&C[(int64_t)m * g.ldc + n], v);
                    continue;
                }
                if (g.Cin && beta != TC(0)) {
                    const int64_t ci = (int64_t)m * g.ldcin + n;
                    v += beta * (cin_f ? (TC)((const float*)g.Cin)[ci] : ((const TC*)g.Cin)[ci]);
                }
                if (out_lower && n > m) v = TC(0);
                C[(int64_t)m * g.ldc + n] = v;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = (float)v;
            }
}

template <typename TC, typename TB>
int dispatch(hipStream_t st, const GemmArgs& g, dim3 grid) {
    const bool akc = !(g.flags & DSVGP_GEMM_TRANS_A);  // A stored [M,K]  -> k contiguous
    const bool bkc = (g.flags & DSVGP_GEMM_TRANS_B);   // B stored [N,K]  -> k contiguous
    if (akc && bkc)        hipLaunchKernelGGL((gemm_kernel<TC, TB, true, true>), grid, dim3(256), 0, st, g);
    else if (akc && !bkc)  hipLaunchKernelGGL((gemm_kernel<TC, TB, true, false>), grid, dim3(256), 0, st, g);
    else if (!akc && bkc)  hipLaunchKernelGGL((gemm_kernel<TC, TB, false, true>), grid, dim3(256), 0, st, g);
    else                   hipLaunchKernelGGL((gemm_kernel<TC, TB, false, false>), grid, dim3(256), 0, st, g);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// ---- inversion of 64x64 diagonal blocks (base case of the blocked trtri) -----------------------
__global__ __launch_bounds__(64) void trtri64_kernel(const double* __restrict__ L, int64_t ldl, int n,
                                                     double* __restrict__ Dinv, int64_t ldd) {
    __shared__ double Ls[64][65];
    __shared__ double Xs[64][65];
    const int r0 = blockIdx.x * 64, c = threadIdx.x;
    for (int r = 0; r < 64; ++r) {
        const int gr = r0 + r, gc = r0 + c;
        double v = (r == c) ? 1.0 : 0.0;  // identity padding past n
        if (gr < n && gc < n && c <= r) v = L[(int64_t)gr * ldl + gc];
        Ls[r][c] = v;
    }
    __syncthreads();
    // thread c owns column c of X = L^-1 (forward substitution against e_c)
    for (int r = 0; r < 64; ++r) {
        double s = (r == c) ? 1.0 : 0.0;
        for (int k = 0; k < r; ++k) s -= Ls[r][k] * Xs[k][c];
        Xs[r][c] = (r < c) ? 0.0 : s / Ls[r][r];
    }
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
        const int gr = r0 + r, gc = r0 + c;
        if (gr < n && gc < n) Dinv[(int64_t)gr * ldd + gc] = Xs[r][c];
    }
}

}  // namespace

int launch_gemm(hipStream_t st, int is_double, const GemmArgs& g) {
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.batch < 1 || g.splitk < 1) return DSVGP_EINVAL;
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch * g.splitk);
    if (is_double) {
        if (g.flags & DSVGP_GEMM_B_IS_FLOAT) return dispatch<double, float>(st, g, grid);
        return dispatch<double, double>(st, g, grid);
    }
    if (g.flags & (DSVGP_GEMM_B_IS_FLOAT | DSVGP_GEMM_CIN_IS_FLOAT)) return DSVGP_EINVAL;
    return dispatch<float, float>(st, g, grid);
}

// Blocked inverse of the nb x nb diagonal blocks of L by recursive doubling:
//   inv([[A,0],[C,D]]) = [[A^-1,0],[-D^-1 C A^-1, D^-1]],  two batched MFMA GEMMs per level.
int launch_trtri_blocks(hipStream_t st, const double* L, int64_t ldl, int n, int nb, double* Dinv,
                        int64_t ldd, double* tmp) {
    if (nb < 64 || (nb & (nb - 1))) return DSVGP_EINVAL;
    hipLaunchKernelGGL(trtri64_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, L, ldl, n, Dinv, ldd);
    DSVGP_LAUNCH_CHECK();
    const int64_t ldt = nb / 2;
    for (int h = 64; h < nb; h *= 2) {
        if (n <= h) break;
        // pair q: top block rows [q*2h, q*2h+h), bottom block rows [q*2h+h, min(n, (q+1)*2h))
        const int npairs = cdiv(n - h, 2 * h);  // pairs whose bottom block is non-empty
        const int last_rows = (n - h) - (npairs - 1) * 2 * h;  // rows of the last bottom block (<= h... clipped)
        GemmArgs g{};
        g.batch = npairs; g.splitk = 1;
        // Tmp = C * A^-1        (A^-1 lower-triangular as the right operand)
        g.M = h; g.M_last = last_rows < h ? last_rows : h; g.N = h; g.K = h; g.K_last = h;
        g.A = L + (int64_t)h * ldl;            g.lda = ldl; g.sA = (int64_t)2 * h * (ldl + 1);
        g.B = Dinv;                            g.ldb = ldd; g.sB = (int64_t)2 * h * (ldd + 1);
        g.C = tmp + (int64_t)h * ldt;          g.ldc = ldt; g.sC = (int64_t)2 * h * ldt;
        g.alpha = 1.0; g.beta = 0.0; g.flags = DSVGP_GEMM_B_LOWER;
        int rc = launch_gemm(st, 1, g);
        if (rc) return rc;
        // bottom-left = -D^-1 * Tmp   (D^-1 lower-triangular as the left operand)
        GemmArgs f{};
        f.batch = npairs; f.splitk = 1;
        f.M = h; f.M_last = g.M_last; f.N = h; f.K = h; f.K_last = g.M_last;
        f.A = Dinv + (int64_t)h * (ldd + 1);   f.lda = ldd; f.sA = (int64_t)2 * h * (ldd + 1);
        f.B = tmp + (int64_t)h * ldt;          f.ldb = ldt; f.sB = (int64_t)2 * h * ldt;
        f.C = Dinv + (int64_t)h * ldd;         f.ldc = ldd; f.sC = (int64_t)2 * h * (ldd + 1);
        f.alpha = -1.0; f.beta = 0.0; f.flags = DSVGP_GEMM_A_LOWER;
        rc = launch_gemm(st, 1, f);
        if (rc) return rc;
    }
    return 0;
}
