// Lean fp64 MFMA GEMM for operands that are both mn-contiguous (A stored [K, M], B stored [K, N]):
//   C = alpha op(A) op(B),  64 x 64 tiles, 4 waves of 32 x 32, 16-deep stages through ONE LDS buffer (20 KB),
// written for register economy (<= 80 VGPRs: 6 workgroups = 24 waves per CU) instead of per-wave tile size: the
// latency of a stage is hidden by the other workgroups of the CU, not by software pipelining inside one.
// Same conventions as gemm.hip (triangular operands trim the K range and mask the diagonal stages, OUT_LOWER skips /
// zeroes the tiles above the diagonal, optional fp32 copy of the result).  No Cin; split-K only in the few-tile regime (below): the caller (launch_gemm)
// sends products with >= GEMM64_MIN_TILES tiles here and keeps gemm.hip for everything else.  Two ways of dealing tiles
// to the 8 XCDs: >= G64_BALANCED_BELOW tiles (the [M', B'] solves) walk 16-column bands per XCD for L2 reuse; fewer tiles
// (the M' x M' x M' class, ~1.5 rounds over the resident workgroups) are dealt in chunks ordered by decreasing K range so
// that every XCD gets the same mix of long and short tiles.
// Measured at C4 (M' = 3000): forward solve 3.58 ms = 61.9 TF (0.79 of the fp64 MFMA peak; 4.09 ms on gemm.hip).
//
// Kernels of this file (round 5: the register-staged ones are the fallbacks of the software-pipelined LDS-DMA forms):
//   gemm64_kernel<TB, A_KC, B_KC>   64 x 64 tiles, register-staged, one LDS buffer, two barriers per stage: what the pipelined form does not take
//                                   (a K-contiguous FLOAT right operand, odd leading dimensions / unaligned bases, dsvgp_ctx::lean_classic)
//   gemm64w_kernel<TB, TW>          64 x TW tiles, register-staged: row-range pieces with LDS padding (the flag-128 experiment)
//   gemm64p_kernel<TB, TW>          64 x 192 (float B) / 64 x 128 (double B) tiles, LDS-DMA stages into two buffers, one barrier per stage:
//                                   the [M', B'] class from 8192 tiles of 64 x 64 up -- the forward solve: 3.30 ms = 67.0 TF at C4 (0.85 of the
//                                   data-sheet peak, 1.00 of the card's sustained rate)
//   gemm64l_kernel<TB, A_KC, B_KC>  64 x 64 tiles, LDS-DMA stages, six / five workgroups per CU: the mid-size solves, [Q' | a], the Cholesky
//                                   backward's products, the float64 model mode's Gram / dense products
#include "common.h"

namespace {

using acc4 = double __attribute__((ext_vector_type(4)));
constexpr int T = 64, BK = 16, LDS_STRIDE = T + 16;       // row stride = 16 (mod 32) doubles: conflict-free b64 fragment reads

struct G64 {
    const double* A; const void* B; double* C; float* C32;
    int64_t lda, ldb, ldc, ldc32;
    int M, N, K, tiles_m, tiles_n, flags, balanced;
    int tri_off;         // wide kernel, triangular A: row m of this product is row tri_off + m of the triangle
    int kchunk;          // > 0: split-K, blockIdx.y-th chunk of this many k of the tile's (trimmed) range; fp64 atomics onto zeros
    double* slab;        // deterministic mode: chunk y stores its tiles to slab[y][M][N] instead (summed in a fixed order afterwards)
    double alpha;
};

#ifndef G64_MINW
#define G64_MINW 6
#endif
#ifndef G64_PRIO
#define G64_PRIO 0
#endif
#ifndef G64_CHUNK
#define G64_CHUNK 8
#endif
#ifndef G64_BALANCED_BELOW
#define G64_BALANCED_BELOW 8192
#endif
#ifndef G64_KCHUNK
#define G64_KCHUNK 768      // split-K chunk (512 / 768 / 1024 measured within 0.3 %) of the few-tile (M' x M' x M') products: the longest tile's 188 dependent stages (K = 3000) set
#endif                      // the duration of a launch whose ~1100-2200 tiles are all resident at once
#ifndef G64_STORE_SWZ
#define G64_STORE_SWZ 0     // 1: bank-conflict-free order of the two 16-byte LDS stores of a staged thread (probe)
#endif
#ifndef G64_KC
#define G64_KC 1            // k-contiguous operands taken (A_KC / B_KC staging)
#endif
#if G64_STORE_SWZ && G64_KC
#error "G64_STORE_SWZ=1 stores ra / rb in the mn-contiguous [k][m] layout only: build the probe with -DG64_KC=0"
#endif
#ifndef G64_BAND
#define G64_BAND 16         // tile columns per band (probed 2 / 4 / 8 / 12 / 16 / 20 / 24 / 32: 56.5 / 57.1 / 60.8 / 62.8 / 63.4 / 57.0 / 62.6 / 54.2 TF) of the XCD-local walk
#endif

// A_KC / B_KC: the operand is K-CONTIGUOUS instead (A stored [M, K] / B stored [N, K], no triangular structure): a thread loads four
// consecutive k of one row with 16-byte loads and scatters them down a column of the [k][m] LDS image -- the same MFMA loop, so
// that L-bar = -tril([Q' | a] [G ; b^T]) (row-major [Q' | a]) and the fp64 model mode's Gram / dense products run here too
// instead of on the 128 x 128 kernel of gemm.hip
template <typename TB, bool A_KC = false, bool B_KC = false>
__global__ __launch_bounds__(256, G64_MINW) void gemm64_kernel(const G64 g) {
    __shared__ double As[BK * LDS_STRIDE];
    __shared__ double Bs[BK * LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int fl = g.flags;
    const int triA = (fl & DSVGP_GEMM_A_LOWER) ? 1 : ((fl & DSVGP_GEMM_A_UPPER) ? 2 : 0);
    const int triB = (fl & DSVGP_GEMM_B_LOWER) ? 2 : ((fl & DSVGP_GEMM_B_UPPER) ? 1 : 0);
    const bool out_lower = fl & DSVGP_GEMM_OUT_LOWER;
    int tm, tn;
    if (g.balanced) {
        // few tiles (a couple of rounds over the resident workgroups) with triangular K ranges: chunks of G64_CHUNK column
        // tiles of ONE tile row (they share the A panel and the K range), rows in order of decreasing K range, chunks dealt
        // round-robin to the XCDs -- every XCD gets the same mix of long and short tiles, longest first
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int per_row = (g.tiles_n + G64_CHUNK - 1) / G64_CHUNK;
        const int chunk = (j / G64_CHUNK) * 8 + xcd;
        if (chunk >= per_row * g.tiles_m) return;
        int rr, cc;
        if (triB && !triA) {
            // the K range depends on the tile COLUMN (triangular B): walk the chunks column-major, longest columns first, so
            // that the round-robin hands every XCD every column chunk (row-major order correlates chunk % 8 with the column)
            const int cr = chunk / g.tiles_m;
            rr = chunk - cr * g.tiles_m;
            cc = (triB == 2) ? cr : per_row - 1 - cr;
            tm = rr;
        } else {
            rr = chunk / per_row;
            cc = chunk - rr * per_row;
            tm = (triA == 1) ? g.tiles_m - 1 - rr : rr;
        }
        tn = cc * G64_CHUNK + j % G64_CHUNK;
        if (tn >= g.tiles_n) return;
    } else {
        // XCD-aware order: consecutive blocks of one XCD (b, b + 8, ...) walk a G64_BAND-wide band of tile columns row by row
        int t = (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
        if (t >= g.tiles_m * g.tiles_n) return;
        const int band = t / (G64_BAND * g.tiles_m), r = t - band * G64_BAND * g.tiles_m;
        const int wcols = min(G64_BAND, g.tiles_n - band * G64_BAND);
        tm = r / wcols;
        tn = band * G64_BAND + r - tm * wcols;
        if (triA == 1) tm = g.tiles_m - 1 - tm;              // longest K ranges first
    }
    const int m0 = tm * T, n0 = tn * T;
    if (out_lower && n0 >= m0 + T) {                         // strictly above the diagonal: defined as zero
        if (g.kchunk) return;                                // (split-K: the caller zeroed the whole output)
        for (int e = tid; e < T * T; e += 256) {
            const int m = m0 + e / T, n = n0 + e % T;
            if (m < g.M && n < g.N) {
                if (g.C) g.C[(int64_t)m * g.ldc + n] = 0.0;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = 0.f;
            }
        }
        return;
    }
    int klo = 0, khi = g.K;
    if (triA == 1) khi = min(khi, m0 + T);
    if (triA == 2) klo = max(klo, (m0 / BK) * BK);
    if (triB == 2) klo = max(klo, (n0 / BK) * BK);
    if (triB == 1) khi = min(khi, n0 + T);
    if (g.kchunk) {
        klo += (int)blockIdx.y * g.kchunk;                   // (klo is a multiple of BK, kchunk too)
        khi = min(khi, klo + g.kchunk);
        if (klo >= khi && !g.slab) return;                   // (deterministic mode: an empty chunk still stores its zeros to the slab)
    }

    // staging: thread -> (k = tid / 16, 4 consecutive columns at 4 (tid % 16)) of the 16 x 64 stage of each operand
    const int sk = tid >> 4, sc = (tid & 15) * 4;
    const bool hs = (tid >> 2) & 1;
    const int kr = tid >> 2, kq = (tid & 3) * 4;           // k-contiguous operand: row kr of the tile, k = kq .. kq + 3 of the stage
    const double* __restrict__ Ap = A_KC ? g.A + (int64_t)min(m0 + kr, g.M - 1) * g.lda + kq : g.A + m0 + sc;
    const TB* __restrict__ Bp = B_KC ? (const TB*)g.B + (int64_t)min(n0 + kr, g.N - 1) * g.ldb + kq : (const TB*)g.B + n0 + sc;
    const bool a_in = m0 + T <= g.M, b_in = n0 + T <= g.N;
    double ra[4], rb[4];
    auto fetch = [&](int k0) {
        if constexpr (A_KC) {
            const bool rin = m0 + kr < g.M;
            if (rin && k0 + kq + 3 < g.K) {
                const double2 v0 = *reinterpret_cast<const double2*>(Ap + k0);
                const double2 v1 = *reinterpret_cast<const double2*>(Ap + k0 + 2);
                ra[0] = v0.x; ra[1] = v0.y; ra[2] = v1.x; ra[3] = v1.y;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[e] = (rin && k0 + kq + e < g.K) ? Ap[k0 + e] : 0.0;
            }
        }
        if constexpr (B_KC) {
            const bool rin = n0 + kr < g.N;
            if (rin && k0 + kq + 3 < g.K) {
                if (sizeof(TB) == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(Bp + k0);
                    rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
                } else {
                    const double2 v0 = *reinterpret_cast<const double2*>(Bp + k0);
                    const double2 v1 = *reinterpret_cast<const double2*>(Bp + k0 + 2);
                    rb[0] = v0.x; rb[1] = v0.y; rb[2] = v1.x; rb[3] = v1.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = (rin && k0 + kq + e < g.K) ? (double)Bp[k0 + e] : 0.0;
            }
        }
        const int k = k0 + sk;
        const bool kin = k < g.K;
        // interior of the tile / of the triangle: unmasked vector loads; otherwise element-wise with the masks
        const bool a_fast = a_in && kin && (triA == 0 || (triA == 1 ? k0 + BK - 1 <= m0 : k0 >= m0 + T - 1));
        const bool b_fast = b_in && kin && (triB == 0 || (triB == 1 ? k0 + BK - 1 <= n0 : k0 >= n0 + T - 1));
        if constexpr (A_KC) {
        } else if (a_fast) {
            const double2 v0 = *reinterpret_cast<const double2*>(Ap + (int64_t)k * g.lda);
            const double2 v1 = *reinterpret_cast<const double2*>(Ap + (int64_t)k * g.lda + 2);
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v1.x; ra[3] = v1.y;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + sc + e;
                bool ok = kin && m < g.M;
                if (triA == 1) ok = ok && k <= m;
                if (triA == 2) ok = ok && k >= m;
                ra[e] = ok ? Ap[(int64_t)k * g.lda + e] : 0.0;
            }
        }
        if constexpr (B_KC) {
        } else if (b_fast) {
            if (sizeof(TB) == 4) {
                const float4 v = *reinterpret_cast<const float4*>(Bp + (int64_t)k * g.ldb);
                rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
            } else {
                const double2 v0 = *reinterpret_cast<const double2*>(Bp + (int64_t)k * g.ldb);
                const double2 v1 = *reinterpret_cast<const double2*>(Bp + (int64_t)k * g.ldb + 2);
                rb[0] = v0.x; rb[1] = v0.y; rb[2] = v1.x; rb[3] = v1.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + sc + e;
                bool ok = kin && n < g.N;
                if (triB == 1) ok = ok && k <= n;
                if (triB == 2) ok = ok && k >= n;
                rb[e] = ok ? (double)Bp[(int64_t)k * g.ldb + e] : 0.0;
            }
        }
    };
    acc4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc4{0, 0, 0, 0};

    if (klo < khi) {
        fetch(klo);
        for (int k0 = klo; k0 < khi; k0 += BK) {
            double* as = As + sk * LDS_STRIDE + sc;
            double* bs = Bs + sk * LDS_STRIDE + sc;
#if G64_STORE_SWZ
            // a thread's 32 bytes go out as two 16-byte stores; with every lane storing its first half first, lanes L and L + 4 of
            // an 8-lane store group hit the same four banks (lane stride 32 B): lanes 4..7 store their halves in the opposite order
            {
                const int o0 = hs ? 2 : 0, o1 = hs ? 0 : 2;
                *reinterpret_cast<double2*>(as + o0) = hs ? double2{ra[2], ra[3]} : double2{ra[0], ra[1]};
                *reinterpret_cast<double2*>(as + o1) = hs ? double2{ra[0], ra[1]} : double2{ra[2], ra[3]};
                *reinterpret_cast<double2*>(bs + o0) = hs ? double2{rb[2], rb[3]} : double2{rb[0], rb[1]};
                *reinterpret_cast<double2*>(bs + o1) = hs ? double2{rb[0], rb[1]} : double2{rb[2], rb[3]};
            }
#else
            if constexpr (A_KC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[(kq + e) * LDS_STRIDE + kr] = ra[e];
            } else {
                *reinterpret_cast<double2*>(as) = double2{ra[0], ra[1]};
                *reinterpret_cast<double2*>(as + 2) = double2{ra[2], ra[3]};
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[(kq + e) * LDS_STRIDE + kr] = rb[e];
            } else {
                *reinterpret_cast<double2*>(bs) = double2{rb[0], rb[1]};
                *reinterpret_cast<double2*>(bs + 2) = double2{rb[2], rb[3]};
            }
#endif
            __syncthreads();
            if (k0 + BK < khi) fetch(k0 + BK);               // in flight under the 16 MFMAs of this stage
            if (G64_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int kq = kk * 4 + (lane >> 4);
                double a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[kq * LDS_STRIDE + wr * 32 + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = Bs[kq * LDS_STRIDE + wc * 32 + j * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (G64_PRIO) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * r;     // f64 C/D layout: row = (lane >> 4) + 4 reg
                const int n = n0 + wc * 32 + j * 16 + (lane & 15);
                if (m >= g.M || n >= g.N) continue;
                double v = g.alpha * acc[i][j][r];
                if (g.kchunk) {
                    if (out_lower && n > m) continue;
                    if (g.slab) g.slab[((int64_t)blockIdx.y * g.M + m) * g.N + n] = v;    // deterministic mode (summed in order afterwards)
#ifdef G64_ABL_NOATOMIC      // (timing ablation only: wrong results)
                    else g.C[(int64_t)m * g.ldc + n] = v;
#else
                    else atomicAdd(&g.C[(int64_t)m * g.ldc + n], v);
#endif
                    continue;
                }
                if (out_lower && n > m) v = 0.0;
                if (g.C) g.C[(int64_t)m * g.ldc + n] = v;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = (float)v;
            }
}

// -------------------------------------------------------------------------------------------------
// Wide variant for the [M', B'] class (>= 8192 tiles of 64 x 64, no split-K, dense mn-contiguous B, optional triangular A: the
// forward panel solve A = L^-1 K_ZX): 64 x TW tiles, 4 waves of 32 x TW/2 (2 x TW/32 MFMA tiles), 16-deep stages through one
// LDS buffer like above.  Against the 64 x 64 tile a stage carries 2-3x the MFMAs between its two barriers, an MFMA needs
// 3/4 - 2/3 of the LDS fragment reads and the tile 2/3 - 5/9 of the operand bytes per flop; the tile ROW stays 64 so that the
// K range of a triangular A is trimmed as finely as before.  Measured at C4 on one box (forward solve, ms): 64 x 64 3.72-3.79,
// TW = 128 (120 VGPRs, 4 workgroups per CU) 3.56-3.62, TW = 192 (164 VGPRs, 3 per CU) 3.51-3.58, TW = 256 (216 VGPRs, 2 per
// CU) 4.3-4.9; bands of 8 tile columns per XCD (4: +0.03-0.1, 10: +0.4, 16: +0.7).  A float right operand uses TW = 192, a double
// one (float64 model mode; 168 VGPRs would spill) TW = 128.
// -------------------------------------------------------------------------------------------------
#ifndef G64_WIDE
#define G64_WIDE 1
#endif
#ifndef G64W_BAND
#define G64W_BAND 8
#endif
#ifndef G64W_MIN_TILES
#define G64W_MIN_TILES G64_BALANCED_BELOW       // tiles of 64 x 64 from which the wide kernel takes the product
#endif
#ifndef G64W_TW
#define G64W_TW 192
#endif
template <typename TB, int TW>
__global__ __launch_bounds__(256, TW > 128 ? 3 : 4) void gemm64w_kernel(const G64 g) {
    constexpr int LDS_STRIDE_W = TW + 16, NJ = TW / 32, NB = TW / 16;       // MFMA column tiles per wave; B values staged per thread
    __shared__ double As[BK * LDS_STRIDE];
    __shared__ double Bs[BK * LDS_STRIDE_W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int fl = g.flags;
    const int triA = (fl & DSVGP_GEMM_A_LOWER) ? 1 : ((fl & DSVGP_GEMM_A_UPPER) ? 2 : 0);
    int t = (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
    if (t >= g.tiles_m * g.tiles_n) return;
    const int band = t / (G64W_BAND * g.tiles_m), r = t - band * G64W_BAND * g.tiles_m;
    const int wcols = min(G64W_BAND, g.tiles_n - band * G64W_BAND);
    int tm = r / wcols;
    const int tn = band * G64W_BAND + r - tm * wcols;
    if (triA == 1) tm = g.tiles_m - 1 - tm;              // longest K ranges first
    const int m0 = tm * T, n0 = tn * TW;
    const int toff = g.tri_off;
    int klo = 0, khi = g.K;
    if (triA == 1) khi = min(khi, m0 + toff + T);
    if (triA == 2) klo = max(klo, ((m0 + toff) / BK) * BK);
    const int sk = tid >> 4, sc = (tid & 15) * 4, sc8 = (tid & 15) * NB;
    const double* __restrict__ Ap = g.A + m0 + sc;
    const TB* __restrict__ Bp = (const TB*)g.B + n0 + sc8;
    const bool a_in = m0 + T <= g.M, b_in = n0 + TW <= g.N;
    double ra[4];
    TB rb[NB];                   // (a float right operand waits in its own width: 12 registers less at TW = 192)
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
        const bool kin = k < g.K;
        const bool a_fast = a_in && kin && (triA == 0 || (triA == 1 ? k0 + BK - 1 <= m0 + toff : k0 >= m0 + toff + T - 1));
        if (a_fast) {
            const double2 v0 = *reinterpret_cast<const double2*>(Ap + (int64_t)k * g.lda);
            const double2 v1 = *reinterpret_cast<const double2*>(Ap + (int64_t)k * g.lda + 2);
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v1.x; ra[3] = v1.y;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + sc + e;
                bool ok = kin && m < g.M;
                if (triA == 1) ok = ok && k <= m + toff;
                if (triA == 2) ok = ok && k >= m + toff;
                ra[e] = ok ? Ap[(int64_t)k * g.lda + e] : 0.0;
            }
        }
        if (b_in && kin) {
            if (sizeof(TB) == 4) {
#pragma unroll
                for (int e = 0; e < NB; e += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(Bp + (int64_t)k * g.ldb + e);
                    rb[e] = (TB)v.x; rb[e + 1] = (TB)v.y; rb[e + 2] = (TB)v.z; rb[e + 3] = (TB)v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NB; e += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(Bp + (int64_t)k * g.ldb + e);
                    rb[e] = (TB)v.x; rb[e + 1] = (TB)v.y;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < NB; ++e) rb[e] = (kin && n0 + sc8 + e < g.N) ? Bp[(int64_t)k * g.ldb + e] : (TB)0;
        }
    };
    acc4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = acc4{0, 0, 0, 0};
    if (klo < khi) {
        fetch(klo);
        for (int k0 = klo; k0 < khi; k0 += BK) {
            double* as = As + sk * LDS_STRIDE + sc;
            double* bs = Bs + sk * LDS_STRIDE_W + sc8;
            *reinterpret_cast<double2*>(as) = double2{ra[0], ra[1]};
            *reinterpret_cast<double2*>(as + 2) = double2{ra[2], ra[3]};
#pragma unroll
            for (int e = 0; e < NB; e += 2) *reinterpret_cast<double2*>(bs + e) = double2{(double)rb[e], (double)rb[e + 1]};
            __syncthreads();
            if (k0 + BK < khi) fetch(k0 + BK);               // in flight under the 32 MFMAs of this stage
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int kq = kk * 4 + (lane >> 4);
                double a[2], b[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[kq * LDS_STRIDE + wr * 32 + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = Bs[kq * LDS_STRIDE_W + wc * (TW / 2) + j * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * rr;
                const int n = n0 + wc * (TW / 2) + j * 16 + (lane & 15);
                if (m >= g.M || n >= g.N) continue;
                const double v = g.alpha * acc[i][j][rr];
                if (g.C) g.C[(int64_t)m * g.ldc + n] = v;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = (float)v;
            }
}

// -------------------------------------------------------------------------------------------------
// Software-pipelined form of the wide kernel (round 5) for a FLOAT right operand: the same 64 x 192 tile, 4 waves of 32 x 96, 16-deep
// stages -- but the stages that lie entirely inside the operands arrive by LDS-DMA (global_load_lds_dwordx4: no register staging, no
// ds_write) into TWO stage buffers, one barrier per stage: stage s + 1 is requested right behind the barrier that opens stage s
// and has that stage's 48 MFMAs per wave to land.  The right operand stays FLOAT in LDS (12 KB per stage instead of 26 KB as
// doubles) and is widened at the fragment read (v_cvt_f64_f32 beside the MFMAs).  The stages at the diagonal of a triangular A /
// a ragged K end keep the masked register-staged path of gemm64w_kernel (at most five of a tile's stages).
// LDS-DMA writes 64 lanes x 16 bytes contiguously, so the images are unpadded and the bank conflicts of the fragment reads
// are removed through the SOURCE addresses: A image [16][64] doubles, the odd k rows with their 16-double halves of each 32
// swapped (column c ^ 16); B image [16][192] floats, the odd k rows rotated by 48 columns -- in both, the two k rows that one
// 32-lane read group touches then fall on disjoint halves of the banks.
// -------------------------------------------------------------------------------------------------
#ifndef G64_PIPE
#define G64_PIPE 1
#endif
typedef __attribute__((address_space(3))) void* g64_lds_ptr_t;
__device__ __forceinline__ void g64_dma16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// TB = float: TW = 192 (B stage 12 KB, three workgroups per CU by LDS, 156 registers).  TB = double (float64 model mode): TW = 128 -- a k row of
// the stage is exactly one 1 KB piece, odd k rows with column c ^ 16 like the A image; 48 KB, three workgroups per CU.
// The K range of a tile is walked as  masked stages | DMA stages | masked stages  (lower A: the diagonal is at the END of the range, upper A:
// at its START; a ragged K end is masked too).
template <typename TB, int TW>
__global__ __launch_bounds__(256, sizeof(TB) == 4 ? 2 : 3) void gemm64p_kernel(const G64 g) {
    constexpr bool BF = sizeof(TB) == 4;
    static_assert((BF && TW == 192) || (!BF && TW == 128), "image permutations are derived for these two shapes");
    constexpr int LDS_STRIDE_W = TW + 16, NJ = TW / 32, NB = TW / 16;
    constexpr int A_STAGE = BK * T * 8, B_STAGE = BK * TW * (int)sizeof(TB);          // bytes of one DMA stage image
    // one LDS block: the two DMA stage pairs, or (masked stages) the padded images of the register-staged path
    constexpr int SLOW_BYTES = (BK * LDS_STRIDE + BK * LDS_STRIDE_W) * 8, FAST_BYTES = 2 * (A_STAGE + B_STAGE);
    __shared__ __attribute__((aligned(16))) unsigned char lds[FAST_BYTES > SLOW_BYTES ? FAST_BYTES : SLOW_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int fl = g.flags;
    const int triA = (fl & DSVGP_GEMM_A_LOWER) ? 1 : ((fl & DSVGP_GEMM_A_UPPER) ? 2 : 0);
    int t = (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
    if (t >= g.tiles_m * g.tiles_n) return;
    const int band = t / (G64W_BAND * g.tiles_m), r = t - band * G64W_BAND * g.tiles_m;
    const int wcols = min(G64W_BAND, g.tiles_n - band * G64W_BAND);
    int tm = r / wcols;
    const int tn = band * G64W_BAND + r - tm * wcols;
    if (triA == 1) tm = g.tiles_m - 1 - tm;              // longest K ranges first
    const int m0 = tm * T, n0 = tn * TW;
    const int toff = g.tri_off;
    int klo = 0, khi = g.K;
    if (triA == 1) khi = min(khi, m0 + toff + T);
    if (triA == 2) klo = max(klo, ((m0 + toff) / BK) * BK);
    // DMA stages [f0, f1): k < K, strictly inside the triangle, and (a DMA piece is 16 bytes: with an odd M / an N that is no multiple of the
    // piece the last piece of a row reaches past the operand's last column -- inside the row's leading dimension, except in the very last k
    // row) not the operands' last k row in that case
    int f0 = klo, f1 = (g.K / BK) * BK;
    if ((g.M & 1) || (g.N & (BF ? 3 : 1))) f1 = min(f1, ((g.K - 1) / BK) * BK);
    if (triA == 1) f1 = min(f1, ((m0 + toff + 1) / BK) * BK);
    if (triA == 2) f0 = max(f0, ((m0 + toff + T - 1 + BK - 1) / BK) * BK);
    f1 = min(f1, khi);
    if (f0 >= f1) { f0 = khi; f1 = khi; }                   // (no DMA stage: one masked range)

    acc4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = acc4{0, 0, 0, 0};
    const int gq = lane >> 4, ml = lane & 15;

    // ---- masked, register-staged stages over [ka, kb), as gemm64w_kernel (the padded images take the LDS block)
    auto masked_range = [&](int ka, int kb) {
        double* As = (double*)lds;
        double* Bs = As + BK * LDS_STRIDE;
        const int sk = tid >> 4, sc = (tid & 15) * 4, sc8 = (tid & 15) * NB;
        const double* __restrict__ Ap = g.A + m0 + sc;
        const TB* __restrict__ Bp = (const TB*)g.B + n0 + sc8;
        double ra[4];
        TB rb[NB];
        auto fetch = [&](int k0) {
            const int k = k0 + sk;
            const bool kin = k < g.K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + sc + e;
                bool ok = kin && m < g.M;
                if (triA == 1) ok = ok && k <= m + toff;
                if (triA == 2) ok = ok && k >= m + toff;
                ra[e] = ok ? Ap[(int64_t)k * g.lda + e] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) rb[e] = (kin && n0 + sc8 + e < g.N) ? Bp[(int64_t)k * g.ldb + e] : (TB)0;
        };
        fetch(ka);
        for (int k0 = ka; k0 < kb; k0 += BK) {
            double* as = As + sk * LDS_STRIDE + sc;
            double* bs = Bs + sk * LDS_STRIDE_W + sc8;
            *reinterpret_cast<double2*>(as) = double2{ra[0], ra[1]};
            *reinterpret_cast<double2*>(as + 2) = double2{ra[2], ra[3]};
#pragma unroll
            for (int e = 0; e < NB; e += 2) *reinterpret_cast<double2*>(bs + e) = double2{(double)rb[e], (double)rb[e + 1]};
            __syncthreads();
            if (k0 + BK < kb) fetch(k0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int kq = kk * 4 + gq;
                double a[2], b[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[kq * LDS_STRIDE + wr * 32 + i * 16 + ml];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = Bs[kq * LDS_STRIDE_W + wc * (TW / 2) + j * 16 + ml];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    };

    if (klo < f0) masked_range(klo, f0);
    if (f0 < f1) {
        const unsigned lds0 = (unsigned)(uintptr_t)(g64_lds_ptr_t)&lds[0];
        const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
        // ---- DMA sources.  A: wave w brings k rows 4 w .. 4 w + 3 of a stage as two 1 KB pieces (two rows each): lane L = 32 par + h
        // writes doubles 2 h, 2 h + 1 of row 2 piece + par, taken from column (2 h) ^ (16 par).  Columns past M (ragged last tile row)
        // are clamped to the row's last pair (the one that holds column M - 1): they feed accumulator rows that are never stored.
        const int par = lane >> 5, hcol = (2 * (lane & 31)) ^ (16 * par);
        const int acol = min(m0 + hcol, ((g.M - 1) & ~1));
        const double* asrc = g.A + (int64_t)(f0 + 4 * wave_u + par) * g.lda + acol;
        // B, float: the [16][192] float image is 12 pieces of 1 KB (256 consecutive floats; 192 = 0 mod 4: a lane's four floats never straddle
        // a row), wave w pieces 3 w .. 3 w + 2: lane L of piece q writes image floats f = 256 q + 4 L .. + 3 = row f / 192, positions
        // p = f mod 192, taken from column (p - 48 (row & 1)) mod 192; columns past N are clamped likewise.  (global_load_lds_dwordx3
        // -- one 768-byte row per instruction -- does NOT pack: it writes its 12 bytes at a 16-byte lane stride.)
        // B, double: the [16][128] image is one 1 KB piece per k row, wave w rows 4 w .. 4 w + 3: lane L writes doubles 2 L, 2 L + 1, taken
        // from column (2 L) ^ (16 (row & 1)).
        constexpr int NBP = BF ? 3 : 4;
        const TB* bsrc[NBP];
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            if constexpr (BF) {
                const int f = 256 * (3 * wave_u + i) + 4 * lane, row = f / TW, pos = f % TW;
                const int c = (pos + TW - 48 * (row & 1)) % TW;
                bsrc[i] = (const TB*)g.B + (int64_t)(f0 + row) * g.ldb + min(n0 + c, (g.N - 1) & ~3);
            } else {
                const int row = 4 * wave_u + i, c = (2 * lane) ^ (16 * (row & 1));
                bsrc[i] = (const TB*)g.B + (int64_t)(f0 + row) * g.ldb + min(n0 + c, (g.N - 1) & ~1);
            }
        }
        const int64_t a2 = 2 * g.lda, astage = (int64_t)BK * g.lda, bstage = (int64_t)BK * g.ldb;
        auto dma = [&](int buf) {
            const unsigned da = lds0 + buf * A_STAGE + wave_u * 2048, db = lds0 + 2 * A_STAGE + buf * B_STAGE + wave_u * (NBP * 1024);
            g64_dma16(asrc, da);
            g64_dma16(asrc + a2, da + 1024);
#pragma unroll
            for (int i = 0; i < NBP; ++i) { g64_dma16(bsrc[i], db + i * 1024); bsrc[i] += bstage; }
            asrc += astage;
        };
        // ---- fragment addresses: lane (gq, ml) reads k row 4 kk + gq; the odd rows' permutation folded into lane bases
        const int odd = gq & 1;
        const double* abase[2];          // i = 0, 1
        abase[0] = (const double*)lds + gq * T + wr * 32 + 16 * odd + ml;
        abase[1] = (const double*)lds + gq * T + wr * 32 + 16 - 16 * odd + ml;
        const TB* bbase = (const TB*)(lds + 2 * A_STAGE) + gq * TW;
        int boff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            boff[j] = BF ? (wc * (TW / 2) + 48 * odd + ml + 16 * j) % TW : ((wc * (TW / 2) + 16 * j + ml) ^ (16 * odd));
        dma(0);
        const int nst = (f1 - f0) / BK;
        for (int s = 0; s < nst; ++s) {
            // this wave's pieces of stage s have landed.  An asm statement: the loop holds no memory operation the compiler knows of, and
            // its wait-count pass drops a builtin s_waitcnt it believes redundant (measured: wrong products from K = 256 on)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                // ... everybody's; and everybody has left stage s - 1's buffer
            if (s + 1 < nst) dma((s + 1) & 1);
            const double* ab0 = abase[0] + (s & 1) * (A_STAGE / 8);
            const double* ab1 = abase[1] + (s & 1) * (A_STAGE / 8);
            const TB* bb = bbase + (s & 1) * (B_STAGE / (int)sizeof(TB));
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[2], b[NJ];
                a[0] = ab0[kk * 4 * T];
                a[1] = ab1[kk * 4 * T];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = (double)bb[kk * 4 * TW + boff[j]];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                    // the DMA images are dead: the padded images of the masked stages take the block
    }
    if (f1 < khi) masked_range(f1, khi);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + wr * 32 + i * 16 + gq + 4 * rr;
                const int n = n0 + wc * (TW / 2) + j * 16 + ml;
                if (m >= g.M || n >= g.N) continue;
                const double v = g.alpha * acc[i][j][rr];
                if (g.C) g.C[(int64_t)m * g.ldc + n] = v;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = (float)v;
            }
}

// -------------------------------------------------------------------------------------------------
// The lean 64 x 64 kernel in the same software-pipelined form (round 5), for the mid-size [M', B'] solves (C3, the data-parallel
// shares) and the [Q' | a] solve: float right operand, mn-contiguous operands, optional triangular A (lower OR upper), no
// triangular B, no split-K.  Stage = A 16 x 64 doubles (8 KB, two DMA pieces per wave) + B 16 x 64 FLOATS (4 KB, one piece per
// wave); two stage buffers = 24 KB, six workgroups per CU as before; one barrier per 16-MFMA stage instead of two, no ds_write.
// Images as gemm64p_kernel's (A: odd k rows with the 16-double halves of each 32 swapped; B: odd k rows with column c ^ 16).
// The K range of a tile is walked as  masked stages | DMA stages | masked stages:  the masked (register-staged) stages are the
// ones at the diagonal of a triangular A -- at the END of the range for a lower A, at its START for an upper one -- and a ragged
// K end.
// -------------------------------------------------------------------------------------------------
#ifndef G64_LEAN_PIPE
#define G64_LEAN_PIPE 1
#endif
#ifndef G64_SMALL_PIPE
#define G64_SMALL_PIPE 1            // 1: products with 64 .. 1023 tiles (small problems) on the four-buffer form of gemm64l_kernel instead of gemm.hip's split-K
#endif
#ifndef G64_LEAN_PIPE_KC
#define G64_LEAN_PIPE_KC 1          // 1: k-contiguous operands too (double right operand; the float64 model mode's Gram / dense products)
#endif
#ifndef G64_LEAN_PIPE_D
#define G64_LEAN_PIPE_D 1           // 1: a double right operand too (the Cholesky backward's products; the float64 model mode)
#endif
#ifndef G64_LEAN_PIPE_UPPER
#define G64_LEAN_PIPE_UPPER 1       // 0: an upper-triangular A (the [Q' | a] solve) stays on gemm64_kernel
#endif
// TB = double (round 5, the Cholesky backward's products and the float64 model mode): the B image is built like the A image (16 KB per stage, 32 KB:
// five workgroups per CU).  A triangular B trims / masks the K range by the tile COLUMN the same way; split-K (kchunk) walks its chunk of the range
// and accumulates with fp64 atomics (or stores to the deterministic slab) as gemm64_kernel does.
// A_KC / B_KC (double operands, no triangular structure: the float64 model mode's Gram and dense products): the operand is K-CONTIGUOUS
// (stored [M, K] / [N, K]); its stage image is [64 rows][16 k] doubles -- 128-byte rows of eight 16-byte chunks, chunk c of row R at position
// c ^ ((R >> 1) & 7) (a 1 KB piece = 8 rows; the 16 rows of a read group then fall on 16 distinct slots of the 256-byte bank row) -- and a
// stage advances along the rows (+ 16 doubles) instead of down the matrix.
// NBUF = 4 (the FEW-tile products of small problems, M' of a few hundred: at most one workgroup per CU, nothing beside it to hide a stage's
// latency): four stage buffers, the DMA of stage s + 3 is requested when stage s opens, so that a tile's chain of K / 16 dependent stages runs
// at the pace of its MFMAs instead of at the pace of a global load -- what split-K over fp64 atomics bought these products before.  A fifth
// buffer takes the dummy requests that keep the count of DMA instructions in flight constant at the end of the range (the counted wait,
// vmcnt((NBUF - 2) x pieces), would otherwise fall through before the last stages have landed).
template <typename TB, bool A_KC = false, bool B_KC = false, int NBUF = 2>
__global__ __launch_bounds__(256, NBUF > 2 ? 1 : (sizeof(TB) == 4 ? G64_MINW : 5)) void gemm64l_kernel(const G64 g) {
    constexpr bool BF = sizeof(TB) == 4;
    static_assert(!B_KC || !BF, "a k-contiguous float operand stays on gemm64_kernel (its 64-byte rows read 2-way conflicted)");
    static_assert(NBUF == 2 || NBUF == 4, "two stage buffers, or four + a dummy");
    constexpr int NBT = NBUF > 2 ? NBUF + 1 : NBUF;                 // buffers in LDS (the last one: dummy target)
    constexpr int A_STAGE = BK * T * 8, B_STAGE = BK * T * (int)sizeof(TB);          // 8 KB + 4 / 8 KB
    constexpr int SLOW_BYTES = 2 * BK * LDS_STRIDE * 8, FAST_BYTES = NBT * (A_STAGE + B_STAGE);
    __shared__ __attribute__((aligned(16))) unsigned char lds[FAST_BYTES > SLOW_BYTES ? FAST_BYTES : SLOW_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int fl = g.flags;
    const int triA = (fl & DSVGP_GEMM_A_LOWER) ? 1 : ((fl & DSVGP_GEMM_A_UPPER) ? 2 : 0);
    const int triB = (fl & DSVGP_GEMM_B_LOWER) ? 2 : ((fl & DSVGP_GEMM_B_UPPER) ? 1 : 0);
    const bool out_lower = fl & DSVGP_GEMM_OUT_LOWER;
    int tm, tn;
    if (g.balanced) {                                       // (the walks of gemm64_kernel)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int per_row = (g.tiles_n + G64_CHUNK - 1) / G64_CHUNK;
        const int chunk = (j / G64_CHUNK) * 8 + xcd;
        if (chunk >= per_row * g.tiles_m) return;
        int rr, cc;
        if (triB && !triA) {
            const int cr = chunk / g.tiles_m;
            rr = chunk - cr * g.tiles_m;
            cc = (triB == 2) ? cr : per_row - 1 - cr;
            tm = rr;
        } else {
            rr = chunk / per_row;
            cc = chunk - rr * per_row;
            tm = (triA == 1) ? g.tiles_m - 1 - rr : rr;
        }
        tn = cc * G64_CHUNK + j % G64_CHUNK;
        if (tn >= g.tiles_n) return;
    } else {
        int t = (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
        if (t >= g.tiles_m * g.tiles_n) return;
        const int band = t / (G64_BAND * g.tiles_m), r = t - band * G64_BAND * g.tiles_m;
        const int wcols = min(G64_BAND, g.tiles_n - band * G64_BAND);
        tm = r / wcols;
        tn = band * G64_BAND + r - tm * wcols;
        if (triA == 1) tm = g.tiles_m - 1 - tm;
    }
    const int m0 = tm * T, n0 = tn * T;
    if (out_lower && n0 >= m0 + T) {                         // strictly above the diagonal: defined as zero
        if (g.kchunk) return;                                // (split-K: the caller zeroed the whole output)
        for (int e = tid; e < T * T; e += 256) {
            const int m = m0 + e / T, n = n0 + e % T;
            if (m < g.M && n < g.N) {
                if (g.C) g.C[(int64_t)m * g.ldc + n] = 0.0;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = 0.f;
            }
        }
        return;
    }
    int klo = 0, khi = g.K;
    if (triA == 1) khi = min(khi, m0 + T);
    if (triA == 2) klo = max(klo, (m0 / BK) * BK);
    if (triB == 2) klo = max(klo, (n0 / BK) * BK);
    if (triB == 1) khi = min(khi, n0 + T);
    if (g.kchunk) {
        klo += (int)blockIdx.y * g.kchunk;                   // (klo is a multiple of BK, kchunk too)
        khi = min(khi, klo + g.kchunk);
        if (klo >= khi && !g.slab) return;                   // (deterministic mode: an empty chunk still stores its zeros to the slab)
    }
    // DMA stages [f0, f1): k < K (and not the operands' very last k row when a 16-byte piece can reach past their last column),
    // strictly inside the triangles: below the tile's first row / column (lower A, upper B), from its last row / column on (upper A, lower B)
    int f1 = (g.K / BK) * BK;
    if ((!A_KC && (g.M & 1)) || (!B_KC && (g.N & (BF ? 3 : 1)))) f1 = min(f1, ((g.K - 1) / BK) * BK);
    int f0 = klo;
    if (triA == 1) f1 = min(f1, ((m0 + 1) / BK) * BK);
    if (triA == 2) f0 = max(f0, ((m0 + T - 1 + BK - 1) / BK) * BK);
    if (triB == 1) f1 = min(f1, ((n0 + 1) / BK) * BK);
    if (triB == 2) f0 = max(f0, ((n0 + T - 1 + BK - 1) / BK) * BK);
    f1 = min(f1, khi);
    if (f0 >= f1) { f0 = max(klo, khi); f1 = f0; }          // (no DMA stage: one masked range)

    acc4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc4{0, 0, 0, 0};
    const int gq = lane >> 4, ml = lane & 15;

    // ---- masked, register-staged stages over [ka, kb) (the padded images take the LDS block)
    auto masked_range = [&](int ka, int kb) {
        double* As = (double*)lds;
        double* Bs = As + BK * LDS_STRIDE;
        const int sk = tid >> 4, sc = (tid & 15) * 4;
        const int kr = tid >> 2, kq4 = (tid & 3) * 4;       // k-contiguous operand: row kr of the tile, k = kq4 .. kq4 + 3 of the stage
        const double* __restrict__ Ap = A_KC ? g.A + (int64_t)min(m0 + kr, g.M - 1) * g.lda + kq4 : g.A + m0 + sc;
        const TB* __restrict__ Bp = B_KC ? (const TB*)g.B + (int64_t)min(n0 + kr, g.N - 1) * g.ldb + kq4 : (const TB*)g.B + n0 + sc;
        double ra[4];
        TB rb[4];
        auto fetch = [&](int k0) {
            const int k = k0 + sk;
            const bool kin = k < g.K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (A_KC) {
                    ra[e] = (m0 + kr < g.M && k0 + kq4 + e < g.K) ? Ap[k0 + e] : 0.0;
                } else {
                    const int m = m0 + sc + e;
                    bool ok = kin && m < g.M;
                    if (triA == 1) ok = ok && k <= m;
                    if (triA == 2) ok = ok && k >= m;
                    ra[e] = ok ? Ap[(int64_t)k * g.lda + e] : 0.0;
                }
                if constexpr (B_KC) {
                    rb[e] = (n0 + kr < g.N && k0 + kq4 + e < g.K) ? Bp[k0 + e] : (TB)0;
                } else {
                    const int n = n0 + sc + e;
                    bool okb = kin && n < g.N;
                    if (triB == 1) okb = okb && k <= n;
                    if (triB == 2) okb = okb && k >= n;
                    rb[e] = okb ? Bp[(int64_t)k * g.ldb + e] : (TB)0;
                }
            }
        };
        fetch(ka);
        for (int k0 = ka; k0 < kb; k0 += BK) {
            double* as = As + sk * LDS_STRIDE + sc;
            double* bs = Bs + sk * LDS_STRIDE + sc;
            if constexpr (A_KC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[(kq4 + e) * LDS_STRIDE + kr] = ra[e];
            } else {
                *reinterpret_cast<double2*>(as) = double2{ra[0], ra[1]};
                *reinterpret_cast<double2*>(as + 2) = double2{ra[2], ra[3]};
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[(kq4 + e) * LDS_STRIDE + kr] = (double)rb[e];
            } else {
                *reinterpret_cast<double2*>(bs) = double2{(double)rb[0], (double)rb[1]};
                *reinterpret_cast<double2*>(bs + 2) = double2{(double)rb[2], (double)rb[3]};
            }
            __syncthreads();
            if (k0 + BK < kb) fetch(k0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int kq = kk * 4 + gq;
                double a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[kq * LDS_STRIDE + wr * 32 + i * 16 + ml];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = Bs[kq * LDS_STRIDE + wc * 32 + j * 16 + ml];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    };

    if (klo < f0) masked_range(klo, min(f0, khi));
    if (f0 < f1) {
        const unsigned lds0 = (unsigned)(uintptr_t)(g64_lds_ptr_t)&lds[0];
        const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
        // A (and a double B): wave w brings k rows 4 w .. 4 w + 3 as two 1 KB pieces (gemm64p_kernel); a float B: ONE 1 KB piece = rows
        // 4 w .. 4 w + 3 of 64 floats: lane L writes floats 4 (L & 15) .. + 3 of row 4 w + (L >> 4), taken from column (4 (L & 15)) ^ (16 (row & 1))
        const int par = lane >> 5, hcol = (2 * (lane & 31)) ^ (16 * par);
        // (k-contiguous operand: wave w brings tile rows 16 w .. 16 w + 15 as two 1 KB pieces of 8 rows: lane L writes chunk position L & 7 of
        //  row 16 w + 8 piece + (L >> 3), taken from chunk (L & 7) ^ ((row >> 1) & 7) of that row; rows past the operand's last: clamped)
        const int krow = 16 * wave_u + (lane >> 3), kch = (lane & 7) ^ ((krow >> 1) & 7), kch2 = (lane & 7) ^ (((krow + 8) >> 1) & 7);
        const double* asrc;
        const double* asrc2;
        if constexpr (A_KC) {
            asrc = g.A + (int64_t)min(m0 + krow, g.M - 1) * g.lda + f0 + 2 * kch;
            asrc2 = g.A + (int64_t)min(m0 + krow + 8, g.M - 1) * g.lda + f0 + 2 * kch2;
        } else {
            asrc = g.A + (int64_t)(f0 + 4 * wave_u + par) * g.lda + min(m0 + hcol, ((g.M - 1) & ~1));
            asrc2 = asrc + 2 * g.lda;
        }
        const TB* bsrc;
        const TB* bsrc2 = nullptr;
        if constexpr (B_KC) {
            bsrc = (const TB*)g.B + (int64_t)min(n0 + krow, g.N - 1) * g.ldb + f0 + 2 * kch;
            bsrc2 = (const TB*)g.B + (int64_t)min(n0 + krow + 8, g.N - 1) * g.ldb + f0 + 2 * kch2;
        } else if constexpr (BF) {
            const int brow = 4 * wave_u + (lane >> 4), bcol = (4 * (lane & 15)) ^ (16 * (brow & 1));
            bsrc = (const TB*)g.B + (int64_t)(f0 + brow) * g.ldb + min(n0 + bcol, (g.N - 1) & ~3);
        } else {
            bsrc = (const TB*)g.B + (int64_t)(f0 + 4 * wave_u + par) * g.ldb + min(n0 + hcol, ((g.N - 1) & ~1));
            bsrc2 = bsrc + 2 * g.ldb;
        }
        const int64_t astage = A_KC ? (int64_t)BK : (int64_t)BK * g.lda, bstage = B_KC ? (int64_t)BK : (int64_t)BK * g.ldb;
        auto dma = [&](int buf) {
            const unsigned da = lds0 + buf * A_STAGE + wave_u * 2048, db = lds0 + NBT * A_STAGE + buf * B_STAGE + wave_u * (BF ? 1024 : 2048);
            g64_dma16(asrc, da);
            g64_dma16(asrc2, da + 1024);
            g64_dma16(bsrc, db);
            if constexpr (!BF) { g64_dma16(bsrc2, db + 1024); bsrc2 += bstage; }
            asrc += astage; asrc2 += astage; bsrc += bstage;
        };
        // (NBUF > 2) the first stage's sources: what a dummy request re-reads, into the dummy buffer
        const double* const asrc_0 = asrc;
        const double* const asrc2_0 = asrc2;
        const TB* const bsrc_0 = bsrc;
        const TB* const bsrc2_0 = bsrc2;
        auto dma_dummy = [&]() {
            const unsigned da = lds0 + (NBT - 1) * A_STAGE + wave_u * 2048, db = lds0 + NBT * A_STAGE + (NBT - 1) * B_STAGE + wave_u * (BF ? 1024 : 2048);
            g64_dma16(asrc_0, da);
            g64_dma16(asrc2_0, da + 1024);
            g64_dma16(bsrc_0, db);
            if constexpr (!BF) g64_dma16(bsrc2_0, db + 1024);
        };
        const int odd = gq & 1;
        // fragment addresses (elements into a stage image): mn-contiguous -- row 4 kk + gq at + kk * 4 * T; k-contiguous -- row m of 16 doubles,
        // chunk (2 kk + (gq >> 1)) ^ ((ml >> 1) & 7), half gq & 1: four lane offsets koff[kk]
        int koff[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) koff[kk] = 2 * ((2 * kk + (gq >> 1)) ^ ((ml >> 1) & 7)) + (gq & 1);
        const double* ab0 = A_KC ? (const double*)lds + (wr * 32 + ml) * BK : (const double*)lds + gq * T + wr * 32 + 16 * odd + ml;
        const double* ab1 = A_KC ? (const double*)lds + (wr * 32 + 16 + ml) * BK : (const double*)lds + gq * T + wr * 32 + 16 - 16 * odd + ml;
        const TB* bb0 = B_KC ? (const TB*)(lds + NBT * A_STAGE) + (wc * 32 + ml) * BK : (const TB*)(lds + NBT * A_STAGE) + gq * T + wc * 32 + 16 * odd + ml;
        const TB* bb1 = B_KC ? (const TB*)(lds + NBT * A_STAGE) + (wc * 32 + 16 + ml) * BK : (const TB*)(lds + NBT * A_STAGE) + gq * T + wc * 32 + 16 - 16 * odd + ml;
        const int nst = (f1 - f0) / BK;
        constexpr int DEPTH = NBUF - 1, NI = BF ? 3 : 4;     // stages in flight; DMA instructions per stage and wave
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            if (q < nst) dma(q);
            else dma_dummy();
        }
        for (int st = 0; st < nst; ++st) {
            // this wave's pieces of stage st have landed: all but the (DEPTH - 1) younger stages' instructions (asm: see gemm64p_kernel)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * NI) : "memory");
            __syncthreads();
            if (st + DEPTH < nst) dma((st + DEPTH) % NBUF);      // (the buffer of stage st - 1: everybody has left it)
            else if (NBUF > 2) dma_dummy();
            const int ao = (st % NBUF) * (A_STAGE / 8), bo = (st % NBUF) * (B_STAGE / (int)sizeof(TB));
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[2], b[2];
                a[0] = ab0[ao + (A_KC ? koff[kk] : kk * 4 * T)];
                a[1] = ab1[ao + (A_KC ? koff[kk] : kk * 4 * T)];
                b[0] = (double)bb0[bo + (B_KC ? koff[kk] : kk * 4 * T)];
                b[1] = (double)bb1[bo + (B_KC ? koff[kk] : kk * 4 * T)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (NBUF > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (dummy requests still in flight)
        __syncthreads();                                    // the DMA images are dead
    }
    if (f1 < khi) masked_range(max(f1, klo), khi);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + gq + 4 * r;
                const int n = n0 + wc * 32 + j * 16 + ml;
                if (m >= g.M || n >= g.N) continue;
                double v = g.alpha * acc[i][j][r];
                if (g.kchunk) {
                    if (out_lower && n > m) continue;
                    if (g.slab) g.slab[((int64_t)blockIdx.y * g.M + m) * g.N + n] = v;    // deterministic mode (summed in order afterwards)
                    else atomicAdd(&g.C[(int64_t)m * g.ldc + n], v);
                    continue;
                }
                if (out_lower && n > m) v = 0.0;
                if (g.C) g.C[(int64_t)m * g.ldc + n] = v;
                if (g.C32) g.C32[(int64_t)m * g.ldc32 + n] = (float)v;
            }
}

// fp32 copy of a split-K result (the atomics accumulate in fp64 only): rows over blockIdx.y, two columns per thread through
// 16-byte loads / 8-byte stores where the rows allow (no per-element 64-bit division: 78 -> ~25 us for the 3000 x 3001 [Q' | a])
template <bool VEC>
__global__ __launch_bounds__(256) void cvt_f64_f32_kernel(const double* __restrict__ C, int64_t ldc, float* __restrict__ C32,
                                                          int64_t ldc32, int M, int N) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c0 >= N) return;
    for (int r = blockIdx.y; r < M; r += gridDim.y) {
        const double* src = C + (int64_t)r * ldc + c0;
        float* dst = C32 + (int64_t)r * ldc32 + c0;
        if (VEC && c0 + 1 < N) {
            const double2 v = *reinterpret_cast<const double2*>(src);
            *reinterpret_cast<float2*>(dst) = float2{(float)v.x, (float)v.y};
        } else {
            dst[0] = (float)src[0];
            if (c0 + 1 < N) dst[1] = (float)src[1];
        }
    }
}

// dst (double) = src (float), row by row (two columns per thread)
__global__ __launch_bounds__(256) void widen_f32_f64_kernel(const float* __restrict__ src, int64_t lds_, double* __restrict__ dst,
                                                            int64_t ldd, int M, int N) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c0 >= N) return;
    for (int r = blockIdx.y; r < M; r += gridDim.y) {
        const float* s = src + (int64_t)r * lds_ + c0;
        double* d = dst + (int64_t)r * ldd + c0;
        d[0] = (double)s[0];
        if (c0 + 1 < N) d[1] = (double)s[1];
    }
}

}  // namespace

// dst = (double) src for an M x N block
void launch_widen_f32_f64(hipStream_t st, const float* src, int64_t ld, double* dst, int64_t ldd, int M, int N) {
    const dim3 grid(cdiv(N, 512), M < 2048 ? M : 2048);
    hipLaunchKernelGGL(widen_f32_f64_kernel, grid, dim3(256), 0, st, src, ld, dst, ldd, M, N);
}

// C32 = (float) C: the fp32 copy of a split-K result that was accumulated in fp64
void launch_cvt_f64_f32(hipStream_t st, const double* C, int64_t ldc, float* C32, int64_t ldc32, int M, int N) {
    const bool vec = ldc % 2 == 0 && ldc32 % 2 == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)C32 % 8) == 0;
    const dim3 grid(cdiv(N, 512), M < 2048 ? M : 2048);
    if (vec) hipLaunchKernelGGL(cvt_f64_f32_kernel<true>, grid, dim3(256), 0, st, C, ldc, C32, ldc32, M, N);
    else hipLaunchKernelGGL(cvt_f64_f32_kernel<false>, grid, dim3(256), 0, st, C, ldc, C32, ldc32, M, N);
}

// returns 1 if the product was taken, 0 if the caller must use gemm.hip, > 1 on a launch error
int launch_gemm64(hipStream_t st, const GemmArgs& g) {
    const int fl = g.flags;
    // operand layouts: mn-contiguous (TRANS_A / no TRANS_B) or, without triangular structure, k-contiguous (A_KC / B_KC staging)
    const bool a_kc = !(fl & DSVGP_GEMM_TRANS_A), b_kc = fl & DSVGP_GEMM_TRANS_B;
    if (a_kc && (fl & (DSVGP_GEMM_A_LOWER | DSVGP_GEMM_A_UPPER))) return 0;
    if (b_kc && (fl & (DSVGP_GEMM_B_LOWER | DSVGP_GEMM_B_UPPER))) return 0;
#if !G64_KC
    if (a_kc || b_kc) return 0;
#endif
    if (g.batch != 1 || g.splitk != 1 || g.Cin || g.kscale || (fl & DSVGP_GEMM_KEEP_UPPER)) return 0;
    if (g.small64 && (a_kc || b_kc || g.N < 4 || g.M < 2 || g.slab)) return 0;            // (the few-tile form: mn-contiguous operands only, not in deterministic mode)
    const bool bf = fl & DSVGP_GEMM_B_IS_FLOAT;
    // 16-byte vector loads: even leading dimensions (multiples of 4 for a float B) and aligned bases
    if (g.lda % 2 || ((uintptr_t)g.A % 16) || (bf ? (g.ldb % 4 || (uintptr_t)g.B % 16) : (g.ldb % 2 || (uintptr_t)g.B % 16))) return 0;
    G64 a{};
    a.A = (const double*)g.A; a.B = g.B; a.C = (double*)g.C; a.C32 = g.C32;
    a.lda = g.lda; a.ldb = g.ldb; a.ldc = g.ldc; a.ldc32 = g.ldc32;
    a.M = g.M; a.N = g.N; a.K = g.K; a.flags = fl; a.alpha = g.alpha;
    a.tiles_m = cdiv(g.M, T); a.tiles_n = cdiv(g.N, T);
    a.tri_off = g.tri_off;
    const int total = a.tiles_m * a.tiles_n;
    a.balanced = total < G64_BALANCED_BELOW;
#if G64_WIDE
    if ((total >= G64W_MIN_TILES || g.wide64) && !a_kc && !b_kc && !(fl & (DSVGP_GEMM_B_LOWER | DSVGP_GEMM_B_UPPER | DSVGP_GEMM_OUT_LOWER)) &&      // (no split-K, no atomics: also in deterministic mode)
        (bf ? g.ldb % 4 == 0 : true)) {
        constexpr int TWD = G64W_TW > 128 ? 128 : G64W_TW;          // (double right operand)
        a.tiles_n = cdiv(g.N, bf ? G64W_TW : TWD);
        const dim3 gridw(cdiv(a.tiles_m * a.tiles_n, 8) * 8);
        const int pad = g.lds_pad > 0 ? g.lds_pad : 0;
        if (pad) {          // (static + dynamic LDS beyond 64 KB needs the attribute; a host-side table write per call)
            const hipError_t ea = bf ? hipFuncSetAttribute((const void*)gemm64w_kernel<float, G64W_TW>, hipFuncAttributeMaxDynamicSharedMemorySize, pad)
                                     : hipFuncSetAttribute((const void*)gemm64w_kernel<double, TWD>, hipFuncAttributeMaxDynamicSharedMemorySize, pad);
            if (ea != hipSuccess) return 1000 + (int)ea;
        }
#if G64_PIPE && G64W_TW == 192
        // (the pipelined form: rows of both operands addressable as 16-byte pieces -- checked above; N >= 4 and M >= 2 for its clamped edge
        //  addresses; not as a row-range piece with LDS padding)
        if (!pad && g.N >= 4 && g.M >= 2) {
            if (bf) hipLaunchKernelGGL((gemm64p_kernel<float, 192>), gridw, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((gemm64p_kernel<double, 128>), gridw, dim3(256), 0, st, a);
        } else
#endif
        if (bf) hipLaunchKernelGGL((gemm64w_kernel<float, G64W_TW>), gridw, dim3(256), pad, st, a);
        else hipLaunchKernelGGL((gemm64w_kernel<double, TWD>), gridw, dim3(256), pad, st, a);
        hipError_t ew = hipGetLastError();
        return ew == hipSuccess ? 1 : 1000 + (int)ew;
    }
#endif
    if (g.tri_off || g.wide64) return DSVGP_EINVAL;                 // (only the wide kernel takes row-range pieces)
    // split-K for the few-tile products with a long K: every tile is resident at once, so the launch lasts as long as its
    // longest tile's chain of K / 16 dependent stages; chunks of G64_KCHUNK accumulate with fp64 atomics onto a zeroed output
    a.kchunk = (G64_KCHUNK > 0 && a.balanced && g.K >= 2 * G64_KCHUNK && g.C && !g.small64) ? G64_KCHUNK : 0;
    int ysplit = 1;
    a.slab = nullptr;
    if (a.kchunk) {
        ysplit = cdiv(g.K, a.kchunk);
        if (g.slab) {                          // deterministic mode: one slab per K chunk (fewer, longer chunks if the scratch is small)
            const int fit = slab_slices(g, ysplit, sizeof(double));
            if (fit < ysplit) { a.kchunk = fit > 1 ? cdiv(cdiv(g.K, fit), BK) * BK : 0; ysplit = a.kchunk ? cdiv(g.K, a.kchunk) : 1; }
            if (a.kchunk) a.slab = (double*)g.slab;
        }
        if (a.kchunk && !a.slab && !(fl & DSVGP_GEMM_C_ZEROED)) {
            hipError_t ez = zero_block(a.C, sizeof(double), a.ldc, a.M, a.N, st);
            if (ez != hipSuccess) return 1000 + (int)ez;
        }
    }
    const dim3 grid(a.balanced ? 8 * G64_CHUNK * cdiv(a.tiles_m * cdiv(a.tiles_n, G64_CHUNK), 8) : cdiv(total, 8) * 8, ysplit);
#if G64_SMALL_PIPE
    if (g.small64) {          // (few tiles: the four-buffer form; the caller -- launch_gemm -- has checked the layouts it takes)
        if (bf) hipLaunchKernelGGL((gemm64l_kernel<float, false, false, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm64l_kernel<double, false, false, 4>), grid, dim3(256), 0, st, a);
        hipError_t es = hipGetLastError();
        return es == hipSuccess ? 1 : 1000 + (int)es;
    }
#endif
    // (k-contiguous operands: the pipelined form takes them when the right operand is double -- G64_LEAN_PIPE_KC -- and the rows are
    //  16-byte addressable: even leading dimensions and aligned bases, checked above)
    const bool lkc = G64_LEAN_PIPE && G64_LEAN_PIPE_KC && !g.lean_classic && g.N >= 4 && g.M >= 2;
    if (a_kc && b_kc) {
        if (bf) hipLaunchKernelGGL((gemm64_kernel<float, true, true>), grid, dim3(256), 0, st, a);
        else if (lkc) hipLaunchKernelGGL((gemm64l_kernel<double, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm64_kernel<double, true, true>), grid, dim3(256), 0, st, a);
    } else if (a_kc) {
        if (lkc && bf) hipLaunchKernelGGL((gemm64l_kernel<float, true, false>), grid, dim3(256), 0, st, a);
        else if (lkc) hipLaunchKernelGGL((gemm64l_kernel<double, true, false>), grid, dim3(256), 0, st, a);
        else if (bf) hipLaunchKernelGGL((gemm64_kernel<float, true, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm64_kernel<double, true, false>), grid, dim3(256), 0, st, a);
    } else if (b_kc) {
        if (bf) hipLaunchKernelGGL((gemm64_kernel<float, false, true>), grid, dim3(256), 0, st, a);
        else if (lkc) hipLaunchKernelGGL((gemm64l_kernel<double, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm64_kernel<double, false, true>), grid, dim3(256), 0, st, a);
    }
#if G64_LEAN_PIPE
    // (the pipelined form of the lean kernel: mn-contiguous operands, N >= 4 and M >= 2 for its clamped edge addresses; G64_LEAN_PIPE_D: a double B too)
    else if (g.N >= 4 && g.M >= 2 && !g.lean_classic && (bf || G64_LEAN_PIPE_D) && (G64_LEAN_PIPE_UPPER || !(fl & DSVGP_GEMM_A_UPPER))) {
        if (bf) hipLaunchKernelGGL(gemm64l_kernel<float>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(gemm64l_kernel<double>, grid, dim3(256), 0, st, a);
    }
#endif
    else if (bf) hipLaunchKernelGGL(gemm64_kernel<float>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm64_kernel<double>, grid, dim3(256), 0, st, a);
    if (a.slab) {
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return 1000 + (int)e0;
        const int rc = launch_splitk_reduce(st, 1, a.slab, ysplit, a.M, a.N, a.C, a.ldc, a.C32, a.ldc32, (fl & DSVGP_GEMM_OUT_LOWER) ? 1 : 0, 0);
        return rc ? rc : 1;
    }
    if (a.kchunk && a.C32) launch_cvt_f64_f32(st, a.C, a.ldc, a.C32, a.ldc32, a.M, a.N);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : 1000 + (int)e;
}
