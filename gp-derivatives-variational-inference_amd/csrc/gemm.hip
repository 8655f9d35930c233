// MFMA GEMM for the DSVGP hot path (gfx950): C = alpha*op(A)op(B) + beta*Cin.
//
// One kernel family serves the panel triangular solve (L^-1 K_ZX and L^-T Abar, reference
// DirectionalGradVariationalStrategy.py:181,183 and their autograd backward), the variational
// products W = L_S^T A, U = L_S W (:192-205) and the M' x M' Gram-type contractions of the backward.
//
// Tiling (64-wide wavefronts): 128x128 output tile per 512-thread workgroup, 4x2 waves, each wave a
// 2x4 grid of 16x16 MFMA tiles (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, K=4 per
// instruction); BK = 16 (fp64) / 32 (fp32) per stage.
// Pipeline: global -> registers (raw element type, no conversion, so the s_waitcnt lands one full
// MFMA phase later) -> LDS (double buffered, ONE barrier per K stage) -> fragments -> MFMA.
// LDS images: a k-contiguous operand is kept [mn][k] with odd row stride BK+1, an mn-contiguous one
// [k][mn] with row stride 144 (= 16 mod 32); both give conflict-free ds_read_b32/_b64 fragment reads.
// blockIdx -> tile mapping is XCD-aware: the 8 XCDs (blocks are dealt round-robin) each get a
// contiguous chunk of the tile grid, so the A row panel of a chunk stays in that XCD's 4 MiB L2;
// for triangular A the heaviest row panels are scheduled first.
#include "common.h"

// tuning / ablation knobs for tools/gemm_probe (defaults = the shipped configuration)
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0      // 1: no MFMA (loads + LDS only), 2: no global loads inside the K loop
#endif
#ifndef GEMM_BK_F32
#define GEMM_BK_F32 32      // 32-deep fp32 stages: twice the MFMA work behind every fetch (dense 101.8 -> 108.9 TF; 16 for the ablation probes)
#endif
#ifndef GEMM_BK_F64
#define GEMM_BK_F64 16
#endif
#ifndef GEMM_SK_OVH
#define GEMM_SK_OVH 64.0      // per-workgroup fixed cost in units of K steps (split-K policy; probed 64 / 128 / 256 / 1024)
#endif
#ifndef GEMM_SK_BINS
#define GEMM_SK_BINS 256
#endif
#ifndef GEMM_MIN_CHUNK
#define GEMM_MIN_CHUNK 16     // smallest XCD dealing unit (workgroups)
#endif
#ifndef GEMM_XCD
#define GEMM_XCD 1
#endif
#ifndef GEMM_PIPE
#define GEMM_PIPE 0         // 1: next stage's LDS stores + the following global fetch are issued BEFORE the MFMA block
#endif
#ifndef GEMM_FRAGPF
#define GEMM_FRAGPF 0       // 1: fragments of k-step kk+1 are read from LDS before the MFMAs of k-step kk
#endif
#ifndef GEMM_PRIO
#define GEMM_PRIO 1         // 1: raise the wave priority for the MFMA block of a stage (inter-wave phase separation: +1-2 % on the fp32 products, neutral on fp64); 0 off
#endif
#ifndef GEMM_MINW
#define GEMM_MINW 0         // tools: override the min-waves-per-SIMD launch bound (0 = NTH / 128)
#endif
#ifndef GEMM_THREADS
#define GEMM_THREADS 512    // 256: 4 waves of 64 x BN/2 outputs; 512: 8 waves of 32 x BN/2 (half the accumulators per wave)
#endif

#ifdef GEMM_CLOCK   // tools only: in-kernel clock of one mid-grid workgroup (shader cycles vs 100 MHz real time)
__device__ unsigned long long dsvgp_gemm_clock_dbg[2];
extern "C" int dsvgp_debug_gemm_clock(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dsvgp_gemm_clock_dbg), 16);
}
#endif

namespace {

constexpr int BM_MAX = 128;  // BN (128 or 64) and BM (128 or 64) are template parameters
#ifndef GEMM_BM64
#define GEMM_BM64 0         // tools: 1 = 64-row tiles on 4-wave workgroups for products without OUT_LOWER
#endif
#ifndef GEMM_SMALL_LIMIT
#define GEMM_SMALL_LIMIT 384   // (128 x 128 tiles) x (coarse split-K slices) below which a product counts as small
#endif
#ifndef GEMM_SMALL
#define GEMM_SMALL 1        // 64 x 64 tiles on 4-wave workgroups (+ finer split-K) for products too small to fill the chip with
                            // 128 x 128 tiles (M' of a few hundred: the reference's own test sizes)
#endif
constexpr int S_MN = 144;

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    using acc_t = float __attribute__((ext_vector_type(4)));
    static constexpr int BK = GEMM_BK_F32;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};
template <> struct Mfma<double> {
    using acc_t = double __attribute__((ext_vector_type(4)));
    static constexpr int BK = GEMM_BK_F64;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout differs from the f32 one: row = (lane>>4) + 4*reg
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// Per-thread element map of one NT x BK operand stage (NT*BK/256 elements per thread):
//   KC (k contiguous in memory):  k = t % BK,          mn = t / BK + (256/BK) * i
//   MN (mn contiguous in memory): mn = t % NT,         k  = t / NT + (256/NT) * i
// tri: 0 none, 1 keep k <= mn, 2 keep k >= mn   (indices local to the operand)
template <typename TIn, bool KC, int BK, int NT, int NTH>
__device__ __forceinline__ void load_raw(TIn (&r)[NT * BK / NTH], const TIn* __restrict__ p, int64_t ld, int mn0,
                                         int MN, int k0, int K, int tri) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NT * BK / NTH; ++i) {
        int mn, k;
        if (KC) { k = k0 + (t % BK); mn = mn0 + t / BK + (NTH / BK) * i; }
        else    { mn = mn0 + (t % NT); k = k0 + t / NT + (NTH / NT) * i; }
        bool ok = (mn < MN) && (k < K);
        if (tri == 1) ok = ok && (k <= mn);
        if (tri == 2) ok = ok && (k >= mn);
        TIn v = TIn(0);
        if (ok) v = KC ? p[(int64_t)mn * ld + k] : p[(int64_t)k * ld + mn];
        r[i] = v;
    }
}

// Interior stages (tile fully inside the matrix, stage fully inside K and off the diagonal of a triangular
// operand): no predicates, one 32-bit per-thread offset per element added to a wave-uniform stage pointer.
template <bool KC, int BK, int NT, int NTH>
__device__ __forceinline__ void make_offsets(int (&off)[NT * BK / NTH], int64_t ld) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NT * BK / NTH; ++i) {
        if (KC) off[i] = (t / BK + (NTH / BK) * i) * (int)ld + (t % BK);
        else    off[i] = (t / NT + (NTH / NT) * i) * (int)ld + (t % NT);
    }
}
template <typename TIn, int E>
__device__ __forceinline__ void load_fast(TIn (&r)[E], const TIn* __restrict__ ps, const int (&off)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = ps[off[i]];
}
// tri: 0 none, 1 keep k <= mn, 2 keep k >= mn : is the stage [k0, k0+BK) x [mn0, mn0+NT) free of masked elements?
__device__ __forceinline__ bool stage_interior(int mn0, int NT, int MN, int k0, int BK, int K, int tri) {
    bool ok = (mn0 + NT <= MN) && (k0 + BK <= K);
    if (tri == 1) ok = ok && (k0 + BK - 1 <= mn0);
    if (tri == 2) ok = ok && (k0 >= mn0 + NT - 1);
    return ok;
}

template <typename TIn, typename TC, bool KC, int BK, int NT, int NTH>
__device__ __forceinline__ void store_stage(TC* __restrict__ s, const TIn (&r)[NT * BK / NTH], TC scale) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NT * BK / NTH; ++i) {
        const TC v = (TC)r[i] * scale;
        if (KC) s[(t / BK + (NTH / BK) * i) * (BK + 1) + (t % BK)] = v;
        else    s[(t / NT + (NTH / NT) * i) * (NT + 16) + (t % NT)] = v;
    }
}

template <typename TC, typename TB, bool AKC, bool BKC, int BN, int NTH, int BM = BM_MAX>
__global__ __launch_bounds__(NTH, GEMM_MINW ? GEMM_MINW : NTH / 128) void gemm_kernel(GemmArgs g) {
    constexpr int NJ = BN / 32;             // 16-wide MFMA column tiles per wave
    constexpr int RW = BM / (NTH / 128);    // rows per wave: 64 (4 waves) or 32 (8 waves; 4 waves on a 64-row tile)
    constexpr int MI = RW / 16;             // 16-high MFMA row tiles per wave
    using M = Mfma<TC>;
    using acc_t = typename M::acc_t;
    // two k-contiguous fp32 operands: 32-deep stages so that every row segment is a full 128-B line
    constexpr int BK = (sizeof(TC) == 4 && AKC && BKC) ? 32 : M::BK;
    constexpr int SK = BK + 1;
    constexpr int SA = BM + 16, SB = BN + 16;   // row strides of the mn-contiguous LDS images (= 16 mod 32)
    constexpr int ASZ = AKC ? BM * SK : BK * SA;
    constexpr int BSZ = BKC ? BN * SK : BK * SB;
    __shared__ TC As[2][ASZ];
    __shared__ TC Bs[2][BSZ];

    const int fl = g.flags;
    const int triA = (fl & DSVGP_GEMM_A_LOWER) ? 1 : ((fl & DSVGP_GEMM_A_UPPER) ? 2 : 0);
    const int triB = (fl & DSVGP_GEMM_B_LOWER) ? 2 : ((fl & DSVGP_GEMM_B_UPPER) ? 1 : 0);
    const bool out_lower = fl & DSVGP_GEMM_OUT_LOWER;
    const bool keep_upper = fl & DSVGP_GEMM_KEEP_UPPER;   // internal: leave m < n untouched instead of zeroing it

    // XCD-aware tile mapping.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one), each
    // XCD keeps 64 workgroups resident (32 CUs x 2): give every XCD an 8x8 SUPERTILE of output tiles at
    // a time, so that the 64 co-resident workgroups of an XCD read only 8 A panels + 8 B panels per K
    // stage out of its private 4 MiB L2 instead of 64 + 64 from HBM / Infinity Cache.
    const int gx = g.tiles_n, gy = g.tiles_m;
    int tm, tn, zz;
    if (g.supertile && GEMM_XCD) {
        // The ACTIVE tiles (all of them, or for OUT_LOWER on a square grid the T(T+1)/2 on / below the diagonal) x
        // batch x split-K slices are enumerated supertile-major -- 64 consecutive entries share ~8 + 8 operand panels --
        // and dealt to the XCDs in chunks of g.chunk entries, so every XCD gets the same number of workgroups
        // (dealing whole 8 x 8 supertiles left the XCDs 20-35 % unbalanced whenever their number x z was not a
        // multiple of 8 or diagonal supertiles held 36 instead of 64 tiles).
        const int CS = g.chunk;
        const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
        const int lin = ((j / CS) * 8 + xcd) * CS + (j % CS);
        const bool lower = g.supertile == 2;
        const int ntiles = lower ? gy * (gy + 1) / 2 : gx * gy;
        if (lin >= ntiles * g.batch * g.splitk) return;
        zz = lin / ntiles;
        int t = lin - zz * ntiles;
        if (lower) {
            int sm = 0, rows;
            for (;; ++sm) {
                rows = min(8, gy - sm * 8);
                const int rowtot = sm * rows * 8 + rows * (rows + 1) / 2;
                if (t < rowtot) break;
                t -= rowtot;
            }
            if (t < sm * rows * 8) {
                const int sn = t / (rows * 8), li = t - sn * rows * 8;
                tm = sm * 8 + (li >> 3);
                tn = sn * 8 + (li & 7);
            } else {
                const int li = t - sm * rows * 8;
                int r = (int)((sqrtf(8.f * (float)li + 1.f) - 1.f) * 0.5f);
                while ((r + 1) * (r + 2) / 2 <= li) ++r;
                while (r * (r + 1) / 2 > li) --r;
                tm = sm * 8 + r;
                tn = sm * 8 + li - r * (r + 1) / 2;
            }
        } else {
            const int nsm = (gy + 7) >> 3, nsn = (gx + 7) >> 3, rows_last = gy - 8 * (nsm - 1);
            int sm, rows;
            if (triA == 1) {                 // lower-triangular A: the longest K ranges (bottom supertile rows) first
                const int first = rows_last * gx;
                if (t < first) { sm = nsm - 1; rows = rows_last; }
                else { t -= first; const int q = t / (8 * gx); sm = nsm - 2 - q; t -= q * 8 * gx; rows = 8; }
            } else {
                sm = min(t / (8 * gx), nsm - 1);
                t -= sm * 8 * gx;
                rows = (sm == nsm - 1) ? rows_last : 8;
            }
            const int sn = min(t / (rows * 8), nsn - 1);
            t -= sn * rows * 8;
            const int cols = min(8, gx - sn * 8);
            const int r = t / cols;
            tm = sm * 8 + r;
            tn = sn * 8 + t - r * cols;
        }
    } else {
        zz = blockIdx.z;
        tm = blockIdx.x / gx;
        tn = blockIdx.x - tm * gx;
        if (triA == 1 || out_lower) tm = gy - 1 - tm;
    }
    const int bz = zz / g.splitk, sp = zz % g.splitk;
    const bool last = (bz == g.batch - 1);
    const int Mdim = (last && g.M_last) ? g.M_last : g.M;
    const int Kdim = (last && g.K_last) ? g.K_last : g.K;
    const int m0 = tm * BM, n0 = tn * BN;   // BN = template tile width
    if (m0 >= Mdim) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const TC* A = (const TC*)g.A + (int64_t)bz * g.sA;
    const TB* B = (const TB*)g.B + (int64_t)bz * g.sB;
    TC* C = (TC*)g.C + (int64_t)bz * g.sC;

    if (out_lower && n0 >= m0 + BM) {  // tile strictly above the diagonal: defined as zero
        if (!keep_upper && (g.splitk == 1 || sp == 0)) {
            for (int e = threadIdx.x; e < BM * BN; e += NTH) {
                int m = m0 + e / BN, n = n0 + e % BN;
                if (m < Mdim && n < g.N) {
                    if (g.splitk == 1 && g.C) C[(int64_t)m * g.ldc + n] = TC(0);
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

#ifdef GEMM_CLOCK
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
#endif
    acc_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    if (klo < khi) {
        TC ra[BM * BK / NTH];
        TB rb[BN * BK / NTH];
        TC ks = TC(1);
        const float* __restrict__ kscale = g.kscale;   // only with k-contiguous A: one k per thread
        int offa[BM * BK / NTH], offb[BN * BK / NTH];
        make_offsets<AKC, BK, BM, NTH>(offa, g.lda);
        make_offsets<BKC, BK, BN, NTH>(offb, g.ldb);
        // the second (predicate-free) load path costs registers: the fp64 kernels with a k-contiguous A and the
        // kernels with two k-contiguous operands spill with it and run slower, so they keep the single path
        constexpr bool USE_FAST = !(sizeof(TC) == 8 && AKC) && !(AKC && BKC);
        const bool small_ld = USE_FAST && (g.lda < (1 << 23)) && (g.ldb < (1 << 23));     // 32-bit offsets are safe
        auto fetch = [&](int k0) {
            if (small_ld && stage_interior(m0, BM, Mdim, k0, BK, Kdim, triA))
                load_fast<TC, BM * BK / NTH>(ra, AKC ? A + (int64_t)m0 * g.lda + k0 : A + (int64_t)k0 * g.lda + m0, offa);
            else
                load_raw<TC, AKC, BK, BM, NTH>(ra, A, g.lda, m0, Mdim, k0, Kdim, triA);
            if (small_ld && stage_interior(n0, BN, g.N, k0, BK, Kdim, triB))
                load_fast<TB, BN * BK / NTH>(rb, BKC ? B + (int64_t)n0 * g.ldb + k0 : B + (int64_t)k0 * g.ldb + n0, offb);
            else
                load_raw<TB, BKC, BK, BN, NTH>(rb, B, g.ldb, n0, g.N, k0, Kdim, triB);
            if (AKC && kscale) { const int k = k0 + (threadIdx.x % BK); ks = (k < Kdim) ? (TC)kscale[k] : TC(0); }
        };
        fetch(klo);
        store_stage<TC, TC, AKC, BK, BM, NTH>(As[0], ra, ks);
        store_stage<TB, TC, BKC, BK, BN, NTH>(Bs[0], rb, TC(1));
        __syncthreads();
        if (klo + BK < khi) fetch(klo + BK);
        int cur = 0;
        for (int k0 = klo; k0 < khi; k0 += BK) {
            const TC* as = As[cur];
            const TC* bs = Bs[cur];
            if (GEMM_PIPE && k0 + BK < khi) {
                // stage k0+BK (in registers since one MFMA block ago) goes to the other buffer now -- all waves passed
                // the barrier after their last reads of it -- and the loads of stage k0+2BK are issued, so that
                // nothing but the barrier separates this stage's MFMA block from the next one
                store_stage<TC, TC, AKC, BK, BM, NTH>(As[cur ^ 1], ra, ks);
                store_stage<TB, TC, BKC, BK, BN, NTH>(Bs[cur ^ 1], rb, TC(1));
                if (GEMM_ABLATE != 2 && k0 + 2 * BK < khi) fetch(k0 + 2 * BK);
            }
            auto frag = [&](int kk, TC (&a)[MI], TC (&b)[NJ]) {
                const int kq = kk * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int mm = wr * RW + i * 16 + (lane & 15);
                    a[i] = AKC ? as[mm * SK + kq] : as[kq * SA + mm];
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nn = wc * (BN / 2) + j * 16 + (lane & 15);
                    b[j] = BKC ? bs[nn * SK + kq] : bs[kq * SB + nn];
                }
            };
            TC a[2][MI], b[2][NJ];
            if (GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            if (GEMM_PRIO == 2) { if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
            if (GEMM_FRAGPF) frag(0, a[0], b[0]);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int cb = GEMM_FRAGPF ? (kk & 1) : 0;
                if (GEMM_FRAGPF) { if (kk + 1 < BK / 4) frag(kk + 1, a[cb ^ 1], b[cb ^ 1]); }
                else frag(kk, a[0], b[0]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (GEMM_ABLATE == 1) { asm volatile("" :: "v"(a[cb][i]), "v"(b[cb][j])); }
                        else acc[i][j] = M::mma(a[cb][i], b[cb][j], acc[i][j]);
                    }
            }
            if (GEMM_PRIO) __builtin_amdgcn_s_setprio(0);
            if (k0 + BK < khi) {
                if (!GEMM_PIPE) {
                    // stage k0+BK (already in registers) goes to the other buffer: nobody reads it any more,
                    // all waves passed the previous barrier after their last reads of it
                    store_stage<TC, TC, AKC, BK, BM, NTH>(As[cur ^ 1], ra, ks);
                    store_stage<TB, TC, BKC, BK, BN, NTH>(Bs[cur ^ 1], rb, TC(1));
                }
                __syncthreads();
                if (!GEMM_PIPE && GEMM_ABLATE != 2 && k0 + 2 * BK < khi) fetch(k0 + 2 * BK);
                cur ^= 1;
            }
        }
    }

#ifdef GEMM_CLOCK
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {
        dsvgp_gemm_clock_dbg[0] = __builtin_amdgcn_s_memtime() - t0c;
        dsvgp_gemm_clock_dbg[1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
#endif
    const TC alpha = (TC)g.alpha, beta = (TC)g.beta;
    const bool cin_f = fl & DSVGP_GEMM_CIN_IS_FLOAT;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * RW + i * 16 + M::row(lane, r);
                const int n = n0 + wc * (BN / 2) + j * 16 + (lane & 15);
                if (m >= Mdim || n >= g.N) continue;
                TC v = alpha * acc[i][j][r];
                if (g.splitk > 1) {
                    if (out_lower && n > m) continue;
                    if (g.slab) ((TC*)g.slab)[((int64_t)sp * g.M + m) * g.N + n] = v;     // deterministic mode: slabs, summed in order afterwards
                    else atomicAdd(&C[(int64_t)m * g.ldc + n], v);
                    continue;
                }
                if (g.Cin && beta != TC(0)) {
                    const int64_t ci = (int64_t)m * g.ldcin + n;
                    v += beta * (cin_f ? (TC)((const float*)g.Cin)[ci] : ((const TC*)g.Cin)[ci]);
                }
                if (out_lower && n > m) { if (keep_upper) continue; v = TC(0); }
                if (g.C) C[(int64_t)m * g.ldc + n] = v;                 // C == nullptr: only the fp32 copy is wanted
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = (float)v;
            }
}

// 8 waves (512 threads, 32 x BN/2 outputs per wave, <= 128 VGPRs, 4 waves/SIMD) measured 5-7 % faster than
// 4 waves (64 x BN/2 per wave) everywhere except the fp64 kernel with two k-contiguous operands, which spills.
template <typename TC, typename TB, bool AKC, bool BKC, int BN>
int launch_one(hipStream_t st, const GemmArgs& g, dim3 grid) {
    constexpr int NTH = (sizeof(TC) == 8 && AKC && BKC) ? 256 : GEMM_THREADS;
#if GEMM_BM64 || GEMM_SMALL
    if (g.bm == 64) {
        hipLaunchKernelGGL((gemm_kernel<TC, TB, AKC, BKC, BN, 256, 64>), grid, dim3(256), 0, st, g);
        DSVGP_LAUNCH_CHECK();
        return 0;
    }
#endif
    size_t pad = 0;
    if (g.flags & DSVGP_GEMM_BACKGROUND) {
        // filler product next to a latency-bound chain on another stream: ONE workgroup per CU (dynamic LDS padding up to
        // just over half of the CU's 160 KB), so that every CU keeps room for a workgroup of the chain
        static size_t static_lds = ~(size_t)0;
        if (static_lds == ~(size_t)0) {
            hipFuncAttributes at{};
            static_lds = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&gemm_kernel<TC, TB, AKC, BKC, BN, NTH>)) == hipSuccess
                             ? at.sharedSizeBytes : 0;
        }
        const size_t want = 82 * 1024;
        pad = static_lds < want ? want - static_lds : 0;
    }
    hipLaunchKernelGGL((gemm_kernel<TC, TB, AKC, BKC, BN, NTH>), grid, dim3(NTH), pad, st, g);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
template <typename TC, typename TB, int BN>
int dispatch_bn(hipStream_t st, const GemmArgs& g, dim3 grid) {
    const bool akc = !(g.flags & DSVGP_GEMM_TRANS_A);  // A stored [M,K]  -> k contiguous
    const bool bkc = (g.flags & DSVGP_GEMM_TRANS_B);   // B stored [N,K]  -> k contiguous
    if (g.kscale && !akc) return DSVGP_EINVAL;
    if (akc && bkc)  return launch_one<TC, TB, true, true, BN>(st, g, grid);
    if (akc && !bkc) return launch_one<TC, TB, true, false, BN>(st, g, grid);
    if (!akc && bkc) return launch_one<TC, TB, false, true, BN>(st, g, grid);
    return launch_one<TC, TB, false, false, BN>(st, g, grid);
}
template <typename TC, typename TB>
int dispatch(hipStream_t st, const GemmArgs& g, dim3 grid) {
    return g.bn == 64 ? dispatch_bn<TC, TB, 64>(st, g, grid) : dispatch_bn<TC, TB, 128>(st, g, grid);
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

// the blocked Cholesky already inverted every 64 x 64 diagonal block (X_k in its workspace, [k][64][64]): scatter
// them onto the diagonal of Dinv instead of inverting again
__global__ __launch_bounds__(256) void trtri64_copy_kernel(const double* __restrict__ X, int n, double* __restrict__ Dinv,
                                                          int64_t ldd) {
    const int r0 = blockIdx.x * 64;
    const double* Xk = X + (size_t)blockIdx.x * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (r0 + r < n && r0 + c < n) Dinv[(int64_t)(r0 + r) * ldd + r0 + c] = (c <= r) ? Xk[e] : 0.0;
    }
}

}  // namespace

// zero an M x N block with leading dimension ld: one linear memset when the rows are contiguous; padded rows through a fill
// kernel of our own (the pitched 2-D memset kernel of the runtime moves < 1 TB/s: ~100 us for a 3000 x 3000 output)
template <typename T>
__global__ __launch_bounds__(256) void zero_rows_kernel(T* __restrict__ C, int64_t ld, int M, int N) {
    // rows over blockIdx.y, 16 bytes per thread where the row start allows it (no per-element 64-bit division)
    constexpr int V = 16 / sizeof(T);
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * V;
    if (c0 >= N) return;
    const bool vec = (ld % V == 0) && ((uintptr_t)C % 16 == 0) && c0 + V <= N;
    for (int r = blockIdx.y; r < M; r += gridDim.y) {
        T* dst = C + (int64_t)r * ld + c0;
        if (vec) *reinterpret_cast<float4*>(dst) = float4{0.f, 0.f, 0.f, 0.f};
        else
            for (int e = 0; e < V && c0 + e < N; ++e) dst[e] = T(0);
    }
}
hipError_t zero_block(void* C, size_t esz, int64_t ld, int M, int N, hipStream_t st) {
    if (ld == N) return hipMemsetAsync(C, 0, esz * (size_t)M * (size_t)N, st);
    if (esz == 8) hipLaunchKernelGGL(zero_rows_kernel<double>, dim3(cdiv(N, 256 * 2), M < 2048 ? M : 2048), dim3(256), 0, st, (double*)C, ld, M, N);
    else if (esz == 4) hipLaunchKernelGGL(zero_rows_kernel<float>, dim3(cdiv(N, 256 * 4), M < 2048 ? M : 2048), dim3(256), 0, st, (float*)C, ld, M, N);
    else return hipMemset2DAsync(C, esz * (size_t)ld, 0, esz * (size_t)N, (size_t)M, st);
    return hipGetLastError();
}

#ifndef GEMM64
#define GEMM64 1            // 1: fp64 products with two mn-contiguous operands and >= GEMM64_MIN_TILES 64 x 64 tiles go to gemm64.hip
#endif
#ifndef GEMM32
#define GEMM32 1            // 1: plain fp32 products (no triangular operands / Cin / kscale) go to gemm32.hip (32x32x2 MFMA)
#endif
#ifndef GEMM64_MIN_TILES
#define GEMM64_MIN_TILES 1024      // (below: 128 x 128 tiles + split-K of this file)
#endif
#ifndef GEMM64_SMALL
#define GEMM64_SMALL 1
#endif
#ifndef GEMM64_SMALL_MIN_TILES
#define GEMM64_SMALL_MIN_TILES 200     // (from 200 tiles: the forward solve of a small problem, 240 tiles at C2, 48 -> 37 us; the 100-tile M'^3 products are FASTER on split-K here: C2 0.545 -> 0.574 ms with them on the pipelined form)
#endif
#ifndef GEMM64_MIN_K
#define GEMM64_MIN_K 256
#endif

int launch_gemm(hipStream_t st, int is_double, const GemmArgs& g) {
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.batch < 1 || g.splitk < 1) return DSVGP_EINVAL;
#if GEMM64
    if (is_double && (g.C || g.C32) && (int64_t)cdiv(g.M, 64) * cdiv(g.N, 64) >= GEMM64_MIN_TILES && g.K >= GEMM64_MIN_K) {
        const int rc = launch_gemm64(st, g);
        if (rc == 1) return 0;
        if (rc > 1) return rc;
    }
#endif
#if GEMM64 && GEMM64_SMALL
    // small problems (M' of a few hundred): 64 .. 1023 tiles with a K chain of >= 256 -- the four-buffer pipelined kernel of gemm64.hip (the chain runs at
    // the pace of its MFMAs) instead of split-K over fp64 atomics + a conversion pass here
    if (is_double && (g.C || g.C32) && !g.slab && !g.tri_off && !g.wide64 && (int64_t)cdiv(g.M, 64) * cdiv(g.N, 64) >= GEMM64_SMALL_MIN_TILES &&
        g.K >= GEMM64_MIN_K && g.K <= 2048) {
        GemmArgs s_ = g;
        s_.small64 = 1;
        const int rc = launch_gemm64(st, s_);
        if (rc == 1) return 0;
        if (rc > 1) return rc;
    }
#endif
    if (g.tri_off || g.wide64) return DSVGP_EINVAL;                 // (row-range pieces exist on gemm64.hip's wide kernel only)
#if GEMM32
    if (!is_double) {
        const int rc = launch_gemm32(st, g);
        if (rc == 1) return 0;
        if (rc > 1) return rc;
    }
#endif
    GemmArgs a = g;
    const bool out_lower_ = g.flags & DSVGP_GEMM_OUT_LOWER;
    a.bm = (GEMM_BM64 && !out_lower_ && g.batch == 1) ? 64 : 128;
    // tile width: 128 x 64 tiles were measured SLOWER than 128 x 128 + split-K for the M' x M' products
    // (chol. backward 2.9 vs 2.3 ms, Gram 4.1 vs 3.8 ms per step at M'=3000), so 128 is used throughout
    a.bn = g.bn == 64 ? 64 : 128;
    int sk_div = 256, sk_min_k = 512;       // split-K granularity: slices of >= 256 k, only for K >= 512
#if GEMM_SMALL
    {   // too few 128 x 128 tiles to occupy the 256 CUs even with the coarse split: quarter tiles, slices of >= 128 k
        const int64_t t128 = (int64_t)cdiv(g.M, 128) * cdiv(g.N, 128) * g.batch;
        const int64_t coarse = t128 * (g.K >= 512 ? (g.K / 256 < 32 ? g.K / 256 : 32) : 1);
        if (g.splitk == 1 && g.bn != 64 && (out_lower_ ? coarse / 2 : coarse) < GEMM_SMALL_LIMIT) {
            a.bm = 64;
            a.bn = 64;
            sk_div = 128;
            sk_min_k = 256;
        }
    }
#endif
    const int BM = a.bm;
    a.tiles_m = cdiv(g.M, BM);
    a.tiles_n = cdiv(g.N, a.bn);
    const size_t esz = is_double ? 8 : 4;
    const bool out_lower = g.flags & DSVGP_GEMM_OUT_LOWER, keep_upper = g.flags & DSVGP_GEMM_KEEP_UPPER;
    // Few output tiles but a long K (the minibatch axis, or M' x M' x M' products): split K so that the
    // grid fills the 512 resident-workgroup slots a few times over; partial sums meet in atomics.
    const bool inplace_acc = g.Cin && g.Cin == g.C && g.beta == 1.0 && g.ldcin == g.ldc;
    if (!g.C && !g.C32) return DSVGP_EINVAL;
    // (a result wanted in fp32 AND fp64 may split too: the slices meet in the fp64 copy, a conversion pass writes the other)
    float* cvt32 = nullptr;
    bool det_acc = g.splitk > 1;                                     // (caller-requested split-K: the slices add onto the caller's C)
    if (a.batch > 1) a.slab = nullptr;                               // (batched products never split)
    if (a.batch == 1 && a.splitk == 1 && (!a.C32 || (is_double && a.C && !a.Cin)) && (!a.Cin || inplace_acc) && a.K >= sk_min_k) {
        const int active = out_lower ? (a.tiles_m * (a.tiles_m + 1)) / 2 * (a.bm / a.bn) : a.tiles_m * a.tiles_n;
        // split factor: minimise (rounds over the 256 CUs -- two resident workgroups share a CU's matrix pipe, so the
        // CU, not the slot, is the unit of throughput) x (K slice + fixed per-workgroup cost)
        int sk = 1;
        if (active > 0 && active < 1024) {
            const int maxsk = a.K / sk_div < 32 ? a.K / sk_div : 32;
            double best = 1e300;
            for (int c = 1; c <= maxsk; ++c) {
                // fp64 products count rounds over the 512 resident slots (one fp64 workgroup alone leaves its CU's matrix
                // pipe half idle: M' x M' solves 0.65 -> 0.58 ms), fp32 over the 256 CUs (probed both ways)
                const double t = (double)cdiv((int64_t)active * c, is_double ? 2 * GEMM_SK_BINS : GEMM_SK_BINS) *
                                 ((double)a.K / c + GEMM_SK_OVH);
                if (t < best * 0.999) { best = t; sk = c; }
            }
        }
        sk = slab_slices(a, sk, esz);                                // deterministic mode: as many slices as the slab holds
        if (sk > 1) {
            a.splitk = sk;
            if (a.C32) { cvt32 = a.C32; a.C32 = nullptr; }
            if (inplace_acc) { a.Cin = nullptr; a.beta = 0.0; det_acc = true; }      // the slices accumulate onto the existing C
            else if (!a.slab && !(g.flags & DSVGP_GEMM_C_ZEROED)) {
                hipError_t e = zero_block(a.C, esz, a.ldc, a.M, a.N, st);
                if (e != hipSuccess) return 1000 + (int)e;
            }
        }
    }
    if (a.splitk == 1) a.slab = nullptr;                             // (caller-requested split-K keeps its slab)
    else if (a.slab && slab_slices(a, a.splitk, esz) != a.splitk) return DSVGP_ENOSPACE;
    if (out_lower && !keep_upper && a.splitk == 1 && a.batch == 1 && g.Cin != g.C && !(g.flags & DSVGP_GEMM_C_ZEROED)) {
        // supertiles strictly above the diagonal are never visited: define them as zero up front
        hipError_t e = a.C ? zero_block(a.C, esz, a.ldc, a.M, a.N, st) : hipSuccess;
        if (e != hipSuccess) return 1000 + (int)e;
        if (a.C32) {
            e = zero_block(a.C32, 4, a.ldc32, a.M, a.N, st);
            if (e != hipSuccess) return 1000 + (int)e;
        }
    }
    a.supertile = (a.tiles_n * a.tiles_m >= 64) ? 1 : 0;
    dim3 grid(a.tiles_n * a.tiles_m, 1, a.batch * a.splitk);
    if (a.supertile) {
        const bool lower = out_lower && a.tiles_n == a.tiles_m;
        a.supertile = lower ? 2 : 1;
        const int64_t total = (lower ? (int64_t)a.tiles_m * (a.tiles_m + 1) / 2 : (int64_t)a.tiles_m * a.tiles_n) * a.batch * a.splitk;
        // chunk dealt to an XCD at a time: 64 (a full set of resident workgroups) unless that leaves the XCDs > 4 % apart
        int cs = 64;
        for (; cs > GEMM_MIN_CHUNK; cs >>= 1) {
            const int64_t nch = cdiv(total, (int64_t)cs);
            if ((double)(cdiv(nch, 8) * 8) / (double)nch <= 1.04) break;
        }
        a.chunk = cs;
        grid = dim3(cdiv(cdiv(total, (int64_t)cs), 8) * 8 * cs, 1, 1);
    }
    const bool slabbed = a.slab && a.splitk > 1;
    if (is_double) {
        int rc = (g.flags & DSVGP_GEMM_B_IS_FLOAT) ? dispatch<double, float>(st, a, grid) : dispatch<double, double>(st, a, grid);
        if (rc == 0 && slabbed)
            return launch_splitk_reduce(st, 1, a.slab, a.splitk, a.M, a.N, a.C, a.ldc, cvt32, g.ldc32, out_lower ? (keep_upper ? 2 : 1) : 0, det_acc);
        if (rc == 0 && cvt32) {
            launch_cvt_f64_f32(st, (const double*)a.C, a.ldc, cvt32, g.ldc32, a.M, a.N);
            DSVGP_LAUNCH_CHECK();
        }
        return rc;
    }
    if (g.flags & (DSVGP_GEMM_B_IS_FLOAT | DSVGP_GEMM_CIN_IS_FLOAT)) return DSVGP_EINVAL;
    const int rc = dispatch<float, float>(st, a, grid);
    if (rc == 0 && slabbed)
        return launch_splitk_reduce(st, 0, a.slab, a.splitk, a.M, a.N, a.C, a.ldc, nullptr, 0, out_lower ? (keep_upper ? 2 : 1) : 0, det_acc);
    return rc;
}

// ---- deterministic mode: the fixed-order sum of split-K slabs -----------------------------------------------------
// out_mode: 0 dense; 1 lower (n > m written as zero); 2 lower, n > m left untouched (KEEP_UPPER)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const T* __restrict__ slab, int ns, int M, int N, T* __restrict__ C,
                                                            int64_t ldc, float* __restrict__ C32, int64_t ldc32, int out_mode,
                                                            int accumulate) {
    const int64_t total = (int64_t)M * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
        T v = T(0);
        if (out_mode && n > m) {
            if (out_mode == 2) continue;
        } else {
            if (accumulate && C) v = C[(int64_t)m * ldc + n];
            for (int s = 0; s < ns; ++s) v += slab[(int64_t)s * total + e];
        }
        if (C) C[(int64_t)m * ldc + n] = v;
        if (C32) C32[(int64_t)m * ldc32 + n] = (float)v;
    }
}
int launch_splitk_reduce(hipStream_t st, int is_double, const void* slab, int nslices, int M, int N, void* C, int64_t ldc,
                         float* C32, int64_t ldc32, int out_lower, int accumulate) {
    const int64_t tot = (int64_t)M * N;
    const int blocks = (int)((tot + 1023) / 1024 < 8192 ? (tot + 1023) / 1024 : 8192);
    if (is_double)
        hipLaunchKernelGGL(splitk_reduce_kernel<double>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, (const double*)slab, nslices,
                           M, N, (double*)C, ldc, C32, ldc32, out_lower, accumulate);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, (const float*)slab, nslices,
                           M, N, (float*)C, ldc, C32, ldc32, out_lower, accumulate);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// Blocked inverse of the nb x nb diagonal blocks of L by recursive doubling:
//   inv([[A,0],[C,D]]) = [[A^-1,0],[-D^-1 C A^-1, D^-1]],  two batched MFMA GEMMs per level.
int launch_trtri_blocks(hipStream_t st, const double* L, int64_t ldl, int n, int nb, double* Dinv,
                        int64_t ldd, double* tmp, const double* X64) {
    if (nb < 64 || (nb & (nb - 1))) return DSVGP_EINVAL;
    if (X64) hipLaunchKernelGGL(trtri64_copy_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, X64, n, Dinv, ldd);
    else hipLaunchKernelGGL(trtri64_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, L, ldl, n, Dinv, ldd);
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
        const int mlast = last_rows < h ? last_rows : h;
        g.M = h; g.M_last = mlast; g.N = h; g.K = h; g.K_last = h;
        g.A = L + (int64_t)h * ldl;            g.lda = ldl; g.sA = (int64_t)2 * h * (ldl + 1);
        g.B = Dinv;                            g.ldb = ldd; g.sB = (int64_t)2 * h * (ldd + 1);
        g.C = tmp + (int64_t)h * ldt;          g.ldc = ldt; g.sC = (int64_t)2 * h * ldt;
        g.alpha = 1.0; g.beta = 0.0; g.flags = DSVGP_GEMM_B_LOWER;
        if (npairs == 1) { g.M = mlast; g.M_last = 0; g.K_last = 0; }    // a single pair: true extents, so split-K can kick in
        int rc = launch_gemm(st, 1, g);
        if (rc) return rc;
        // bottom-left = -D^-1 * Tmp   (D^-1 lower-triangular as the left operand)
        GemmArgs f{};
        f.batch = npairs; f.splitk = 1;
        f.M = h; f.M_last = mlast; f.N = h; f.K = h; f.K_last = mlast;
        f.A = Dinv + (int64_t)h * (ldd + 1);   f.lda = ldd; f.sA = (int64_t)2 * h * (ldd + 1);
        f.B = tmp + (int64_t)h * ldt;          f.ldb = ldt; f.sB = (int64_t)2 * h * ldt;
        f.C = Dinv + (int64_t)h * ldd;         f.ldc = ldd; f.sC = (int64_t)2 * h * (ldd + 1);
        f.alpha = -1.0; f.beta = 0.0; f.flags = DSVGP_GEMM_A_LOWER;
        if (npairs == 1) { f.M = mlast; f.K = mlast; f.M_last = 0; f.K_last = 0; }
        rc = launch_gemm(st, 1, f);
        if (rc) return rc;
    }
    return 0;
}
