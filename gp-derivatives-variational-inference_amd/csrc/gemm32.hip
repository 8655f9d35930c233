// fp32 MFMA GEMM on v_mfma_f32_32x32x2_f32 for the two big fp32 products of the ELBO fast path:
//   K_ZX-bar = alpha [Q' | a] [A ; mu_bar^T]        dense,  A operand [M][K] or [K][M], B operand [K][N]   (M' x B' x M')
//   [G ; b^T] = tril([A ; mu_bar^T] A^T)             both operands [.][K] (k-contiguous), split-K over the minibatch axis
// 128 x 128 output tiles, 4 waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32 -> 64 accumulator registers), BK-deep stages
// through ONE LDS buffer per operand with the next stage prefetched into registers under the MFMAs of the current one;
// several workgroups per CU hide each other's stage boundaries (the recipe of gemm64.hip).  Why this MFMA shape: a
// 32 x 32 x 2 instruction does 4096 flops on ONE register of A and ONE of B per lane -- half the LDS fragment traffic per
// flop of the 16 x 16 x 4 form gemm.hip is built on -- and a 64 x 64 wave tile needs 4 fragment reads per 8 MFMAs.
//
// Fragment reads.  Lane l = 32 h + r supplies A[row r][k = h] and B[k = h][col r] to one MFMA.  A k-contiguous operand
// sits in LDS as [mn][BK + 2] floats and every lane reads the PAIR (k0 + 2h, k0 + 2h + 1) with one 8-byte read
// (row stride 34 / 18 words = 2 x odd: the 32 lanes of a read group hit 32 distinct even banks of the 64): the two MFMAs
// of a pair then cover k0 .. k0 + 3 in the order {0, 2}, {1, 3} -- any order is fine as long as both operands use the same.
// An mn-contiguous operand sits as [BK][128 + 4] and is read with two 4-byte reads at rows k0 + 2h, k0 + 2h + 1.
// Staging stores: 8-byte stores for the k-contiguous image (rows are 8-byte aligned), 16-byte for the other.
#include <type_traits>

#include "common.h"

namespace {

using acc16 = float __attribute__((ext_vector_type(16)));
using acc4f = float __attribute__((ext_vector_type(4)));
constexpr int TM = 128, TN = 128;

#ifndef G32_BK
#define G32_BK 32
#endif
#ifndef G32_MINW
#define G32_MINW 3          // waves per SIMD the register allocation aims at (workgroups per CU)
#endif
#ifndef G32_BAND
#define G32_BAND 8          // tile columns per XCD-local band of the dense walk
#endif
#ifndef G32_PRIO
#define G32_PRIO 0
#endif
#ifndef G32_DMA_BK
#define G32_DMA_BK 32       // stage depth of the LDS-DMA kernel (32: two workgroups per CU; 16: three)
#endif
#ifndef G32_DMA
#define G32_DMA 1           // 1: stages by LDS-DMA (gemm32_dma_kernel) where eligible, 0: register staging only
#endif
#ifndef G32_MF
#define G32_MF 32           // MFMA shape of the LDS-DMA kernel: 32 (32x32x2) or 16 (16x16x4)
#endif
#ifndef G32_DMAPRIO
#define G32_DMAPRIO 0
#endif
#ifndef G32_SK_BELOW
#define G32_SK_BELOW 4096   // tile count below which the split-K cost model is consulted (few rounds: a ragged last round costs most)
#endif
#ifndef G32_SK_MINK
#define G32_SK_MINK 512     // shortest K that may be split (slices of >= 256)
#endif
#ifndef G32_SK_SLOTS
#define G32_SK_SLOTS 256    // work units per round in the split-K cost model: one per CU (a CU's matrix pipe is shared by its resident workgroups)
#endif
#ifndef G32_SK_MINT
#define G32_SK_MINT 0
#endif
#ifndef G32_TRI_SB
#define G32_TRI_SB 1        // lower-triangular outputs: super-block edge (tiles) of the walk; 1 = plain row-major triangle.  Measured
#endif                      // (round 4, tools/gemm32_variants.sh, same box): Gram at C4 2.169 (1) / 2.193 (8) / 2.222 ms (4) -- no gain
#ifndef G32_XCDK
#define G32_XCDK 0          // 1: (probe) lower-triangular split-K products in eight K slices, one per XCD -- measured, not kept
#endif
#ifndef G32_ABL
#define G32_ABL 0           // timing ablations (results are WRONG): 1 no global fetch in the loop, 2 no LDS stores, 4 no LDS fragment reads, 8 no barriers
#endif

struct G32 {
    const float* A; const float* B; float* C;
    int64_t lda, ldb, ldc;
    int M, N, K, tiles_m, tiles_n, flags, splitk, kslice, ntiles;
    float alpha;
    float* slab;             // deterministic mode: K slice s stores its tiles to slab[s][M][N] (summed in a fixed order afterwards)
};

template <int BK, bool KC>
struct Stage {                       // register image of one thread's share of a 128 x BK stage of one operand
    static constexpr int NV = BK / 8;             // float4 per thread
    float4 v[NV];
};

// Ragged edges never take a scalar path: every load is a 16-byte vector load from a CLAMPED address (rows / columns past the
// edge re-read the last valid ones: their products land in output rows / columns that are never stored; the launcher
// checks that the leading dimension covers the last clamped chunk), and only the stage that crosses the end of the K
// range zeroes its k >= kend elements with selects.
// k-contiguous operand P[mn][k]: thread -> (row = tid / (BK/4) + i * 1024/BK, 4 k at 4 (tid % (BK/4)))
template <int BK>
struct FetchKC {
    static constexpr int CPR = BK / 4, RPP = 256 / CPR, NV = BK / 8;
    const float* P;
    int rowoff[NV];          // clamped row * ld
    int kc, klim;            // this thread's k offset inside a stage; last in-bounds chunk start of a row
    __device__ __forceinline__ void init(const float* P_, int64_t ld, int mn0, int MN, int K, int tid) {
        P = P_;
        kc = (tid % CPR) * 4;
        klim = ((K + 3) / 4) * 4 - 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) rowoff[i] = (int)(min(mn0 + tid / CPR + i * RPP, MN - 1) * ld);
    }
    __device__ __forceinline__ void load(int k0, int kend, Stage<BK, true>& s) const {
        const int k = k0 + kc;
        const float* p = P + min(k, klim);
#pragma unroll
        for (int i = 0; i < NV; ++i) s.v[i] = *reinterpret_cast<const float4*>(p + rowoff[i]);
        if (k0 + BK > kend) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                s.v[i].x = (k < kend) ? s.v[i].x : 0.f;
                s.v[i].y = (k + 1 < kend) ? s.v[i].y : 0.f;
                s.v[i].z = (k + 2 < kend) ? s.v[i].z : 0.f;
                s.v[i].w = (k + 3 < kend) ? s.v[i].w : 0.f;
            }
        }
    }
};
// mn-contiguous operand P[k][mn]: thread -> (k = tid / 32 + 8 i, 4 columns at 4 (tid % 32))
template <int BK>
struct FetchMC {
    static constexpr int NV = BK / 8;
    const float* P;
    int64_t ld;
    int kr, K;
    __device__ __forceinline__ void init(const float* P_, int64_t ld_, int mn0, int MN, int K_, int tid) {
        ld = ld_; K = K_;
        kr = tid >> 5;
        P = P_ + min(mn0 + (tid & 31) * 4, ((MN + 3) / 4) * 4 - 4);
    }
    __device__ __forceinline__ void load(int k0, int kend, Stage<BK, false>& s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = k0 + kr + 8 * i;
            s.v[i] = *reinterpret_cast<const float4*>(P + (int64_t)min(k, K - 1) * ld);
            if (k0 + BK > kend && k >= kend) s.v[i] = float4{0.f, 0.f, 0.f, 0.f};
        }
    }
};

// which (tile, K slice) this workgroup takes: consecutive blocks of one XCD (b, b + 8, ...) get consecutive units
__device__ __forceinline__ bool pick_unit(const G32& g, int& m0, int& n0, int& kbeg, int& kend) {
#if G32_XCDK
    // probe (round 6, profiles/r06_b_gemm32_gram_xcd.txt): a lower-triangular split-K product with EIGHT K slices, slice = the XCD the workgroup
    // lands on (blocks b, b + 8, ... share one): every XCD's L2 then sees one K slice of the operands only
    const bool xk = (g.flags & DSVGP_GEMM_OUT_LOWER) && g.splitk == 8;
    const int u = xk ? 0 : (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
    if (xk ? (int)(blockIdx.x >> 3) >= g.ntiles : u >= g.ntiles * g.splitk) return false;
    const int slice = xk ? (int)(blockIdx.x & 7) : u / g.ntiles;
    const int t = xk ? (int)(blockIdx.x >> 3) : u - slice * g.ntiles;
#else
    const int u = (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
    if (u >= g.ntiles * g.splitk) return false;
    const int slice = u / g.ntiles;
    const int t = u - slice * g.ntiles;
#endif
    int tm, tn;
    if (g.flags & DSVGP_GEMM_OUT_LOWER) {
        // tiles with tn <= tm, row by row: the triangle t = tm (tm + 1) / 2 + tn while tm < tiles_n, full rows of
        // tiles_n tiles below it (tiles_m > tiles_n: e.g. the extra row b^T of [G ; b^T] opening a tile row of its own)
        const int tri = min(g.tiles_m, g.tiles_n), t0 = tri * (tri + 1) / 2;
        if (t < t0) {
#if G32_TRI_SB > 1
            // the triangle in SB x SB super-blocks, row by row (the 64 workgroups an XCD holds at a time then share SB row panels
            // and SB column panels -- the walk of the dense product -- instead of 2-3 whole tile rows: 3 + 21 panels)
            constexpr int SB = G32_TRI_SB;
            int q = t, I = 0, J = 0, rI = 0, cJ = 0;
            const int nsb = (tri + SB - 1) / SB;
            bool found = false;
            for (I = 0; I < nsb && !found; ++I) {
                rI = min(SB, tri - SB * I);
                for (J = 0; J <= I; ++J) {
                    cJ = min(SB, tri - SB * J);
                    const int cnt = (J < I) ? rI * cJ : rI * (rI + 1) / 2;
                    if (q < cnt) { found = true; break; }
                    q -= cnt;
                }
                if (found) break;
            }
            if (J < I) {
                tm = SB * I + q / cJ;
                tn = SB * J + q % cJ;
            } else {
                int a = (int)((sqrtf(8.f * (float)q + 1.f) - 1.f) * 0.5f);
                while ((a + 1) * (a + 2) / 2 <= q) ++a;
                while (a * (a + 1) / 2 > q) --a;
                tm = SB * I + a;
                tn = SB * J + q - a * (a + 1) / 2;
            }
#else
            tm = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
            while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
            while (tm * (tm + 1) / 2 > t) --tm;
            tn = t - tm * (tm + 1) / 2;
#endif
        } else {
            tm = tri + (t - t0) / g.tiles_n;
            tn = (t - t0) % g.tiles_n;
        }
    } else {                                              // G32_BAND-wide bands of tile columns, row by row inside a band
        const int band = t / (G32_BAND * g.tiles_m), q = t - band * G32_BAND * g.tiles_m;
        const int wcols = min(G32_BAND, g.tiles_n - band * G32_BAND);
        tm = q / wcols;
        tn = band * G32_BAND + q - tm * wcols;
    }
    m0 = tm * TM; n0 = tn * TN;
    kbeg = slice * g.kslice; kend = min(g.K, kbeg + g.kslice);
    return true;
}

// C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (c & 3) + 8 (c >> 2) + 4 (lane >> 5)
__device__ __forceinline__ void store_tile(const G32& g, const acc16 (&acc)[2][2], int m0, int n0, int wr, int wc, int h, int r,
                                           float* Cb, int64_t ldcb, bool atomic_) {
#ifdef G32_ABL_NOATOMIC          // (timing ablation only: wrong results under split-K)
    const bool atomic = false, out_lower = g.flags & DSVGP_GEMM_OUT_LOWER;
#else
    const bool atomic = atomic_, out_lower = g.flags & DSVGP_GEMM_OUT_LOWER;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int m = m0 + wr * 64 + i * 32 + (c & 3) + 8 * (c >> 2) + 4 * h;
                const int n = n0 + wc * 64 + j * 32 + r;
                if (m >= g.M || n >= g.N) continue;
                if (out_lower && n > m) continue;                // (the caller zero-fills m < n)
                const float v = g.alpha * acc[i][j][c];
                float* dst = Cb + (int64_t)m * ldcb + n;
                if (atomic) atomicAdd(dst, v);
                else *dst = v;
            }
}
// where this workgroup's tile goes: the output itself (atomics when K is split) or, in deterministic mode, its K slice's slab
#define G32_DEST()                                                                                     \
    const int slice_ = g.kslice > 0 ? kbeg / g.kslice : 0;                                             \
    float* const Cb = g.slab ? g.slab + (int64_t)slice_ * g.M * g.N : g.C;                            \
    const int64_t ldcb = g.slab ? (int64_t)g.N : g.ldc;                                                \
    const bool atomic_ = g.splitk > 1 && !g.slab

template <int BK, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, G32_MINW) void gemm32_kernel(const G32 g) {
    constexpr int SKC = BK + 2, SMC = TM + 4;            // LDS row strides (floats) of the two images
    constexpr int A_WORDS = A_KC ? TM * SKC : BK * SMC, B_WORDS = B_KC ? TN * SKC : BK * SMC;
    __shared__ __attribute__((aligned(16))) float lds[A_WORDS + B_WORDS];
    float* As = lds;
    float* Bs = lds + A_WORDS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int h = lane >> 5, r = lane & 31;
    int m0, n0, kbeg, kend;
    if (!pick_unit(g, m0, n0, kbeg, kend)) return;

    Stage<BK, A_KC> ra;
    Stage<BK, B_KC> rb;
    typename std::conditional<A_KC, FetchKC<BK>, FetchMC<BK>>::type fa;
    typename std::conditional<B_KC, FetchKC<BK>, FetchMC<BK>>::type fb;
    fa.init(g.A, g.lda, m0, g.M, g.K, tid);
    fb.init(g.B, g.ldb, n0, g.N, g.K, tid);
    auto fetch = [&](int k0) {
        fa.load(k0, kend, ra);
        fb.load(k0, kend, rb);
    };
    auto commit = [&]() {
        if constexpr (A_KC) {
            constexpr int CPR = BK / 4, RPP = 256 / CPR;
            float* p = As + (tid / CPR) * SKC + (tid % CPR) * 4;
#pragma unroll
            for (int i = 0; i < BK / 8; ++i) {
                *reinterpret_cast<float2*>(p + i * RPP * SKC) = float2{ra.v[i].x, ra.v[i].y};
                *reinterpret_cast<float2*>(p + i * RPP * SKC + 2) = float2{ra.v[i].z, ra.v[i].w};
            }
        } else {
            float* p = As + (tid >> 5) * SMC + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < BK / 8; ++i) *reinterpret_cast<float4*>(p + 8 * i * SMC) = ra.v[i];
        }
        if constexpr (B_KC) {
            constexpr int CPR = BK / 4, RPP = 256 / CPR;
            float* p = Bs + (tid / CPR) * SKC + (tid % CPR) * 4;
#pragma unroll
            for (int i = 0; i < BK / 8; ++i) {
                *reinterpret_cast<float2*>(p + i * RPP * SKC) = float2{rb.v[i].x, rb.v[i].y};
                *reinterpret_cast<float2*>(p + i * RPP * SKC + 2) = float2{rb.v[i].z, rb.v[i].w};
            }
        } else {
            float* p = Bs + (tid >> 5) * SMC + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < BK / 8; ++i) *reinterpret_cast<float4*>(p + 8 * i * SMC) = rb.v[i];
        }
    };

    acc16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[i][j][c] = 0.f;

    // per-lane fragment bases inside the LDS images
    const float* af = A_KC ? As + (wr * 64 + r) * SKC + 2 * h : As + (2 * h) * SMC + wr * 64 + r;
    const float* bf = B_KC ? Bs + (wc * 64 + r) * SKC + 2 * h : Bs + (2 * h) * SMC + wc * 64 + r;

    if (kbeg < kend) {
        fetch(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            if (!(G32_ABL & 2)) commit();
            if (!(G32_ABL & 8)) __syncthreads();
            if (!(G32_ABL & 1) && k0 + BK < kend) fetch(k0 + BK);                 // in flight under the MFMAs of this stage
            if (G32_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float2 a[2], b[2];
                const int kr_ = (G32_ABL & 4) ? 0 : kk;        // (ablation: the same fragment every step -> reads hoisted)
#define kk kr_
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (A_KC) a[i] = *reinterpret_cast<const float2*>(af + i * 32 * SKC + 4 * kk);
                    else a[i] = float2{af[(4 * kk) * SMC + i * 32], af[(4 * kk + 1) * SMC + i * 32]};
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (B_KC) b[j] = *reinterpret_cast<const float2*>(bf + j * 32 * SKC + 4 * kk);
                    else b[j] = float2{bf[(4 * kk) * SMC + j * 32], bf[(4 * kk + 1) * SMC + j * 32]};
                }
#undef kk
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    }
            }
            if (G32_PRIO) __builtin_amdgcn_s_setprio(0);
            if (!(G32_ABL & 8)) __syncthreads();
        }
    }

    G32_DEST();
    store_tile(g, acc, m0, n0, wr, wc, h, r, Cb, ldcb, atomic_);
}

// -------------------------------------------------------------------------------------------------
// The same product with the stages brought in by LDS-DMA (global_load_lds_dwordx4: global -> LDS without a register
// round trip and without ds_write traffic; ablations of the register-staged kernel above: the global loads cost 12 % and
// the LDS stores 10 % of its time).  BK = 32, TWO 32 KB LDS buffers (64 KB per workgroup, two workgroups per CU): the DMA of
// stage s + 1 is issued before the 64 MFMAs of stage s and waited for (vmcnt(0) + barrier) after them.
// An LDS-DMA instruction writes 64 x 16 bytes CONTIGUOUSLY (wave-uniform base + 16 lane), so the images cannot be padded:
//   mn-contiguous operand: [32 k][128] floats, linear -- fragment reads (consecutive lanes, consecutive words) need no pad;
//   k-contiguous operand:  [128 rows][32 k] floats with the 16-byte chunks of row R stored at position c ^ ((R >> 1) & 7)
//                          (the swizzle is applied to the SOURCE address of each lane); lane (h, r) reads chunk 2 j + h of its
//                          row with ONE 16-byte read per 4 MFMAs: the 16 lanes of a read group hit 16 distinct 4-bank slots.
// k order inside a stage: MFMA e of chunk pair j takes k = 8 j + 4 h + e from both operands.
// K tail: lanes whose chunk / row lies past K read a 16-byte zero block instead; a k-contiguous operand whose K is not a
// multiple of 4 must be zero-padded up to it by the caller (DSVGP_GEMM_K_PADDED) -- otherwise the register-staged kernel runs.
// -------------------------------------------------------------------------------------------------
#ifdef G32_STAMP   // diagnostic build only (tools/gemm32_probe.cpp prints the shares): where a stage of a mid-grid wave goes
__device__ unsigned long long g32_stamps[8];
#define G32_T(var) unsigned long long var; do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G32_T(var) do { } while (0)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __attribute__((aligned(16))) float g32_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

// One LDS-DMA instruction, hidden from the compiler's s_waitcnt bookkeeping: issued through the builtin, hipcc treats the
// transfer as an LDS store that any later ds_read may alias and parks a vmcnt(0) in front of the first fragment read of
// the stage -- the DMA then overlaps nothing.  As an asm statement it is register-safe (no VGPR destination) and WE count
// it: one "s_waitcnt vmcnt(0)" before the barrier that ends the stage.  M0 (the LDS destination base) is compiler-reserved:
// saved, set and restored inside the statement.
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// Round 4, measured and taken out again (profiles/r04_b_gemm32_patterns.txt): dealing only the USEFUL 32 x 32 MFMA tiles of a
// diagonal tile of a lower output (10 of 16, as 3 / 3 / 2 / 2 per wave) and of a ragged last tile row (M' = 3000: 56 valid rows -> 2
// MFMA tiles per wave) through per-wave copies of the K loop.  Ragged rows: Gram 2.077 -> 2.083 ms, dense 3.53 -> 3.55 (nothing:
// the two workgroups of a CU alternate MFMA and DMA-issue phases in lockstep, a shorter MFMA phase of one does not shorten the
// period of the pair); diagonal tiles: Gram 2.077 -> 2.124 ms at C4 and 0.633 -> 0.774 ms at C3 (slower); the mere presence of
// the extra loop copies costs 1-2 % (2.077 -> 2.114 with the patterns switched off at run time).
// MF = 32: 2 x 2 tiles of v_mfma_f32_32x32x2_f32 per wave;  MF = 16: 4 x 4 tiles of v_mfma_f32_16x16x4_f32 (same 64 x 64 wave tile,
// same LDS images and DMA; twice the fragment reads per flop, but the smaller shape holds a higher clock under load).
// Lane l = TS h + r (TS = 32 / 16 rows per tile, h = k slot 0..1 / 0..3) reads chunk (64 / MF) j + h of its row per read.
// BK (G32_DMA_BK): 32 -- two 32 KB buffers, two workgroups per CU; 16 -- two 16 KB buffers, LDS and registers (<= 129) leave room for
// three: a third wave per SIMD fills the matrix pipe while the other two sit in their DMA-issue / barrier phases (round 5).  A
// k-contiguous image then has 64-byte rows of four 16-byte chunks, chunk c of row R at position c ^ ((R >> 2) & 3) (the four rows
// of a read group that share R mod 4 -- the same quarter of the 256-byte bank row -- differ in (R >> 2) & 3), 16 rows per 1 KB piece.
template <int MF, bool A_KC, bool B_KC, int BK>
__global__ __launch_bounds__(256, BK == 16 ? 3 : 2) void gemm32_dma_kernel(const G32 g) {
    static_assert(BK == 32 || (BK == 16 && MF == 32), "the 64-byte-row swizzle is derived for the 32 x 32 x 2 shape");
    constexpr int OPW = 128 * BK;                         // words per operand image
    constexpr int NP = BK / 8;                            // 1 KB DMA pieces per operand, stage and wave
    constexpr int CH = BK / 4, RPP = 256 / BK;            // 16-byte chunks per k-contiguous row; rows per piece
    constexpr int TS = MF, NT = 64 / TS, NH = 64 / TS, NJ = BK / (4 * NH);   // tile size, tiles per wave side, k slots, chunk groups / stage
    __shared__ __attribute__((aligned(16))) float lds[2][2 * OPW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int h = lane / TS, r = lane % TS;
    int m0, n0, kbeg, kend;
    if (!pick_unit(g, m0, n0, kbeg, kend)) return;

    // ---- DMA sources: NP instructions per operand and stage, instruction i of wave w fills the 1 KB block NP w + i
    const float* asrc[NP];
    const float* bsrc[NP];
    int akk[NP], bkk[NP];                                 // k of the lane's chunk / row inside a stage (for the K tail)
    const int K4 = (g.K + 3) / 4 * 4;
    auto swz = [](int R) { return BK == 32 ? ((R >> 1) & 7) : ((R >> 2) & 3); };
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int blk = wave * NP + i;
        if constexpr (A_KC) {
            const int R = blk * RPP + lane / CH, c = (lane % CH) ^ swz(R);
            asrc[i] = g.A + (int64_t)min(m0 + R, g.M - 1) * g.lda + kbeg + 4 * c;
            akk[i] = 4 * c;
        } else {
            const int k = blk * 2 + (lane >> 5);
            asrc[i] = g.A + (int64_t)(kbeg + k) * g.lda + min(m0 + 4 * (lane & 31), (g.M + 3) / 4 * 4 - 4);
            akk[i] = k;
        }
        if constexpr (B_KC) {
            const int R = blk * RPP + lane / CH, c = (lane % CH) ^ swz(R);
            bsrc[i] = g.B + (int64_t)min(n0 + R, g.N - 1) * g.ldb + kbeg + 4 * c;
            bkk[i] = 4 * c;
        } else {
            const int k = blk * 2 + (lane >> 5);
            bsrc[i] = g.B + (int64_t)(kbeg + k) * g.ldb + min(n0 + 4 * (lane & 31), (g.N + 3) / 4 * 4 - 4);
            bkk[i] = k;
        }
    }
    const int64_t astep = A_KC ? BK : (int64_t)BK * g.lda, bstep = B_KC ? BK : (int64_t)BK * g.ldb;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)&lds[0][0];
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma = [&](int buf, int k0) {
        const unsigned dst = lds_base + (unsigned)buf * (2 * OPW * 4) + wave_u * (NP * 1024);     // byte address, wave-uniform
        const bool tail = k0 + BK > kend;                 // (only the last stage of the last K slice)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float* sa = asrc[i];
            const float* sb = bsrc[i];
            if (tail) {
                if (k0 + akk[i] >= (A_KC ? K4 : g.K)) sa = g32_zero_chunk;
                if (k0 + bkk[i] >= (B_KC ? K4 : g.K)) sb = g32_zero_chunk;
            }
            lds_dma16(sa, dst + i * 1024);
            lds_dma16(sb, dst + OPW * 4 + i * 1024);
            asrc[i] += astep;
            bsrc[i] += bstep;
        }
    };

    using accT = typename std::conditional<MF == 32, acc16, acc4f>::type;
    accT acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < (MF == 32 ? 16 : 4); ++c) acc[i][j][c] = 0.f;

    // per-lane fragment offsets (words) inside an operand image; tile i of the wave adds i * TS rows / columns
    const int q7 = h ^ swz(r);
    int aoff[NJ], boff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        aoff[j] = A_KC ? (wr * 64 + r) * BK + 4 * ((NH * j) ^ q7) : (4 * NH * j + 4 * h) * 128 + wr * 64 + r;
        boff[j] = B_KC ? (wc * 64 + r) * BK + 4 * ((NH * j) ^ q7) : (4 * NH * j + 4 * h) * 128 + wc * 64 + r;
    }

    // fragments of one chunk group j (4 NH k): 4 k-values per lane for each of the NT + NT MFMA tiles
    struct Frag { float a[NT][4], b[NT][4]; };
    auto load_frag = [&](const float* As, const float* Bs, int j, Frag& f) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if constexpr (A_KC) {
                const float4 v = *reinterpret_cast<const float4*>(As + aoff[j] + i * TS * BK);
                f.a[i][0] = v.x; f.a[i][1] = v.y; f.a[i][2] = v.z; f.a[i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.a[i][e] = As[aoff[j] + e * 128 + i * TS];
            }
            if constexpr (B_KC) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + boff[j] + i * TS * BK);
                f.b[i][0] = v.x; f.b[i][1] = v.y; f.b[i][2] = v.z; f.b[i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.b[i][e] = Bs[boff[j] + e * 128 + i * TS];
            }
        }
    };

    if (kbeg < kend) {
        dma(0, kbeg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // stage 0 has landed for every wave
        int buf = 0;
#ifdef G32_STAMP
        unsigned long long d_dma = 0, d_mfma = 0, d_vm = 0, d_bar = 0, n_st = 0;
#endif
        for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
            G32_T(t0);
            if (k0 + BK < kend) dma(buf ^ 1, k0 + BK);     // lands under the MFMAs below
            G32_T(t1);
            const float* As = &lds[buf][0];
            const float* Bs = As + OPW;
            if (G32_PRIO) __builtin_amdgcn_s_setprio(1);
            // the fragments of chunk group j + 1 are requested before the MFMAs of group j: the LDS latency rides under them
            Frag f[2];
            load_frag(As, Bs, 0, f[0]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) load_frag(As, Bs, j + 1, f[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);         // (hipcc otherwise sinks the reads to just before their use)
                const Frag& c = f[j & 1];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int jn = 0; jn < NT; ++jn) {
                            if constexpr (MF == 32)
                                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[i][e], c.b[jn][e], acc[i][jn], 0, 0, 0);
                            else
                                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[i][e], c.b[jn][e], acc[i][jn], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (G32_PRIO) __builtin_amdgcn_s_setprio(0);
            G32_T(t2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of the next stage has landed ...
            G32_T(t3);
            __syncthreads();                                   // ... and every wave's; this buffer is free for stage + 2
#ifdef G32_STAMP
            G32_T(t4);
            d_dma += t1 - t0; d_mfma += t2 - t1; d_vm += t3 - t2; d_bar += t4 - t3; ++n_st;
#endif
        }
#ifdef G32_STAMP
        if (tid == 0 && blockIdx.x == gridDim.x / 2 + 8) {
            g32_stamps[0] = d_dma; g32_stamps[1] = d_mfma; g32_stamps[2] = d_vm; g32_stamps[3] = d_bar; g32_stamps[4] = n_st;
        }
#endif
    }
    G32_DEST();
    if constexpr (MF == 32) {
        store_tile(g, acc, m0, n0, wr, wc, h, r, Cb, ldcb, atomic_);
    } else {
        // C/D layout of the 16 x 16 form: col = lane & 15, row = 4 (lane >> 4) + reg
        const bool atomic = atomic_, out_lower = g.flags & DSVGP_GEMM_OUT_LOWER;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int m = m0 + wr * 64 + i * 16 + 4 * h + c;
                    const int n = n0 + wc * 64 + j * 16 + r;
                    if (m >= g.M || n >= g.N) continue;
                    if (out_lower && n > m) continue;
                    const float v = g.alpha * acc[i][j][c];
                    float* dst = Cb + (int64_t)m * ldcb + n;
                    if (atomic) atomicAdd(dst, v);
                    else *dst = v;
                }
    }
}

template <int BK>
int dispatch32(hipStream_t st, const G32& a, dim3 grid, bool akc, bool bkc) {
    if (akc && bkc) hipLaunchKernelGGL((gemm32_kernel<BK, true, true>), grid, dim3(256), 0, st, a);
    else if (akc) hipLaunchKernelGGL((gemm32_kernel<BK, true, false>), grid, dim3(256), 0, st, a);
    else if (bkc) hipLaunchKernelGGL((gemm32_kernel<BK, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm32_kernel<BK, false, false>), grid, dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : 1000 + (int)e;
}

}  // namespace

#ifdef G32_STAMP
int g32_read_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g32_stamps), sizeof(unsigned long long) * 8); }
#endif

// returns 1 if the product was taken, 0 if the caller must use gemm.hip, > 1 on a launch error.
// Taken: plain fp32 products (no triangular operands, no Cin, no kscale, no second output) with >= 128 x 128 x 512
// of work per tile; OUT_LOWER (square tile grids) zero-fills the strict upper triangle like gemm.hip does.
int launch_gemm32(hipStream_t st, const GemmArgs& g) {
    const int fl = g.flags;
    if (fl & ~(DSVGP_GEMM_TRANS_A | DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER | DSVGP_GEMM_K_PADDED | DSVGP_GEMM_C_ZEROED | DSVGP_GEMM_UPPER_UNDEF)) return 0;
    if (g.batch != 1 || g.splitk != 1 || g.Cin || g.kscale || g.C32 || !g.C || g.beta != 0.0) return 0;
    if (g.M < 512 || g.N < 512 || g.K < 512) return 0;
    if (g.lda % 4 || g.ldb % 4 || ((uintptr_t)g.A % 16) || ((uintptr_t)g.B % 16)) return 0;     // 16-byte vector loads
    {   // the clamped vector loads of ragged edges stay inside the rows: ld covers the minor extent rounded up to 4
        const int64_t a_minor = (fl & DSVGP_GEMM_TRANS_A) ? g.M : g.K, b_minor = (fl & DSVGP_GEMM_TRANS_B) ? g.K : g.N;
        if (g.lda < (a_minor + 3) / 4 * 4 || g.ldb < (b_minor + 3) / 4 * 4) return 0;
        const int64_t a_rows = (fl & DSVGP_GEMM_TRANS_A) ? g.K : g.M, b_rows = (fl & DSVGP_GEMM_TRANS_B) ? g.N : g.K;
        if (a_rows * g.lda >= (int64_t)1 << 31 || b_rows * g.ldb >= (int64_t)1 << 31) return 0;     // 32-bit row offsets
    }
    const bool out_lower = fl & DSVGP_GEMM_OUT_LOWER;
    G32 a{};
    a.A = (const float*)g.A; a.B = (const float*)g.B; a.C = (float*)g.C;
    a.lda = g.lda; a.ldb = g.ldb; a.ldc = g.ldc;
    a.M = g.M; a.N = g.N; a.K = g.K; a.flags = fl; a.alpha = (float)g.alpha;
    a.tiles_m = cdiv(g.M, TM); a.tiles_n = cdiv(g.N, TN);
    if (out_lower) {
        const int tri = a.tiles_m < a.tiles_n ? a.tiles_m : a.tiles_n;
        a.ntiles = tri * (tri + 1) / 2 + (a.tiles_m > tri ? (a.tiles_m - tri) * a.tiles_n : 0);
    } else {
        a.ntiles = a.tiles_m * a.tiles_n;
    }
    // split-K: few rounds of tiles and a long K (the Gram product over the minibatch axis; the [B', M'] x [M', M'] products of
    // CIQ: 1152 tiles = 4.5 per CU).  Cost model: per-CU makespan, ceil(units / 256) x (slice length + fixed cost); measured
    // against fixed slice counts: C5's K_ZZ products 1 / 2 / 3 / 4 / 6 slices -> 80.9 / 77.1 / 78.7 / 77.7 / 78.7 ms per step
    // (the model picks 2), the Gram product of C4 5 slices as before
    int sk = 1;
    if (a.ntiles < G32_SK_BELOW && g.K >= G32_SK_MINK) {
        double best = 1e300;
        const int maxsk = g.K / 256 < 64 ? g.K / 256 : 64;
        for (int c = 1; c <= maxsk; ++c) {
            const double tcost = (double)cdiv((int64_t)a.ntiles * c, G32_SK_SLOTS) * ((double)g.K / c + 384.0);
            if (tcost < best * 0.999) { best = tcost; sk = c; }
        }
    }
#if G32_XCDK
    if (out_lower && sk > 1 && g.K >= 8 * 512) sk = 8;
#endif
#ifdef G32_SK      // probe: fixed slice count for the products of G32_SK_MINT..G32_SK_BELOW tiles
    if (a.ntiles < G32_SK_BELOW && a.ntiles >= G32_SK_MINT && g.K >= 1024 && g.K < 16384) sk = G32_SK;
#endif
    sk = slab_slices(g, sk, sizeof(float));         // deterministic mode: as many slices as the caller's scratch holds
    a.splitk = sk;
    a.kslice = cdiv(cdiv(g.K, sk), 32) * 32;
    a.splitk = cdiv(g.K, a.kslice);
    a.slab = (g.slab && a.splitk > 1) ? (float*)g.slab : nullptr;
    if (!a.slab && (a.splitk > 1 || (out_lower && !(fl & DSVGP_GEMM_UPPER_UNDEF))) && !(fl & DSVGP_GEMM_C_ZEROED)) {
        // atomics accumulate onto zeros / the strict upper triangle is defined as zero
        // (contiguous rows: one linear fill -- the 2-D fill kernel of the runtime takes 104 us for 36 MB, the linear one ~10)
        hipError_t e = zero_block(a.C, sizeof(float), a.ldc, a.M, a.N, st);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    const dim3 grid(cdiv((int64_t)a.ntiles * a.splitk, 8) * 8);
    const bool akc = !(fl & DSVGP_GEMM_TRANS_A), bkc = (fl & DSVGP_GEMM_TRANS_B) != 0;
#if G32_DMA
    // LDS-DMA kernel: a k-contiguous operand needs K % 4 == 0 or caller-zeroed padding up to it (the chunk that straddles K
    // is read as it lies in memory)
    if (!((akc || bkc) && g.K % 4 != 0 && !(g.flags & DSVGP_GEMM_K_PADDED))) {
        if (akc && bkc) hipLaunchKernelGGL((gemm32_dma_kernel<G32_MF, true, true, G32_DMA_BK>), grid, dim3(256), 0, st, a);
        else if (akc) hipLaunchKernelGGL((gemm32_dma_kernel<G32_MF, true, false, G32_DMA_BK>), grid, dim3(256), 0, st, a);
        else if (bkc) hipLaunchKernelGGL((gemm32_dma_kernel<G32_MF, false, true, G32_DMA_BK>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm32_dma_kernel<G32_MF, false, false, G32_DMA_BK>), grid, dim3(256), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return 1000 + (int)e;
    } else
#endif
    {
        const int rc = dispatch32<G32_BK>(st, a, grid, akc, bkc);
        if (rc != 1) return rc;
    }
    if (a.slab) {
        const int rc = launch_splitk_reduce(st, 0, a.slab, a.splitk, a.M, a.N, a.C, a.ldc, nullptr, 0, out_lower ? 1 : 0, 0);
        if (rc) return rc;
    }
    return 1;
}
