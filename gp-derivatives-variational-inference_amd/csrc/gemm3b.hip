// fp32-equivalent GEMM on the bf16 matrix pipe (round 4, OPT-IN: flag 32 of dsvgp_elbo_step_f32 / `bench.py --split-bf16`; the
// default step keeps v_mfma_f32_32x32x2_f32).
//
// Every fp32 operand is cut into three bf16 planes  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): the two
// subtractions are exact in fp32, the remainder after three planes is below 2^-26 |x| -- finer than fp32 itself), and a product
// keeps the six terms of (h + m + l)(h' + m' + l') down to 2^-16:
//       a b  ~=  h h' + (h m' + m h') + (h l' + m m' + l h')            (dropped: m l' + l m' + l l' < 2^-24 |a b|)
// Each term is a bf16 x bf16 product (exact in fp32) accumulated in fp32 by v_mfma_f32_32x32x16_bf16, which retires 32 x 32 x 16
// multiply-adds in 32 cycles where the fp32 MFMA needs 8 x 64: six of them are 192 against 512 cycles per 32 x 32 x 16 block.
// What the bf16 pipe asks for in return is operand bandwidth: 6 bytes per element instead of 4 in 0.375 of the time, i.e. 4x the
// LDS fill rate of the fp32 kernel at equal tiles -- hence 256 x 256 output tiles (2.7x the flops per staged byte of 128 x 128) and
// operand planes split ONCE per step by a streaming pass (dsvgp_split3_bf16), not in the GEMM's inner loop.
//
// Kernel: C[M, N] = alpha sum_k A[m, k] B[n, k], both operands k-contiguous plane triples (a [K, N] operand is split through a
// transposing pass).  One workgroup of 8 waves (4 x 2, wave tile 64 x 128 = 2 x 4 MFMA tiles, 128 accumulator registers) per
// 256 x 256 tile, 16-deep stages (one MFMA K step) brought in by LDS-DMA into three 48 KB buffers (two stages in flight); a plane image is
// [256 rows][16 k] bf16 = 32 bytes per row with the two 16-byte halves of a row swapped on rows 8..15 (mod 16), so that the
// ds_read_b128 of a 16-lane group (16 consecutive rows, one half each) covers all 64 banks once.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using acc16 = float __attribute__((ext_vector_type(16)));
constexpr int T3 = 256, BK3 = 16;
constexpr int IMG3 = T3 * BK3 * 2;              // bytes of one plane image of one operand (8 KB)
constexpr int STAGE3 = 6 * IMG3;                // A planes 0..2, B planes 0..2 (48 KB)

constexpr int G3_NBUF = 3;
#ifndef G3_CHAIN_K
#define G3_CHAIN_K 2048
#endif
#ifndef G3_SK_SLOTS
#define G3_SK_SLOTS 256
#endif

struct G3 {
    const unsigned short* A; const unsigned short* B; float* C;
    int64_t ldc, pa, pb;                        // plane strides (elements)
    int ra, rb;                                 // rows of the operands as split (the K-blocked plane layout needs them)
    int M, N, K, tiles_m, tiles_n, flags, splitk, kslice, ntiles;
    float alpha;
};

typedef __attribute__((address_space(3))) void* lds_ptr3_t;
__device__ __forceinline__ void lds_dma16_3b(const unsigned short* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

__global__ __launch_bounds__(512, 1) void gemm3b_kernel(const G3 g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int h = lane >> 5, r = lane & 31;
    // ---- which (tile, K slice): consecutive blocks of one XCD take consecutive units; lower-triangular outputs enumerate tn <= tm
    const int u = (blockIdx.x >> 3) + (blockIdx.x & 7) * ((gridDim.x + 7) >> 3);
    if (u >= g.ntiles * g.splitk) return;
    const int slice = u / g.ntiles, t = u - slice * g.ntiles;
    int tm, tn;
    if (g.flags & DSVGP_GEMM_OUT_LOWER) {
        // tiles with tn <= tm, row by row: the triangle while tm < tiles_n, full rows of tiles_n tiles below it (tiles_m > tiles_n:
        // the extra row b^T of [G ; b^T] opening a tile row of its own when M' is a multiple of 256)
        const int tri = min(g.tiles_m, g.tiles_n), t0 = tri * (tri + 1) / 2;
        if (t < t0) {
            tm = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
            while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
            while (tm * (tm + 1) / 2 > t) --tm;
            tn = t - tm * (tm + 1) / 2;
        } else {
            tm = tri + (t - t0) / g.tiles_n;
            tn = (t - t0) % g.tiles_n;
        }
    } else {                                      // bands of 8 tile columns, row by row inside a band
        const int band = t / (8 * g.tiles_m), q = t - band * 8 * g.tiles_m;
        const int wcols = min(8, g.tiles_n - band * 8);
        tm = q / wcols;
        tn = band * 8 + q - tm * wcols;
    }
    const int m0 = tm * T3, n0 = tn * T3;
    const int kbeg = slice * g.kslice, kend = min(g.K, kbeg + g.kslice);

    // ---- DMA sources: 48 instructions per stage (6 images x 8 blocks of 32 rows), instruction 6 w + i belongs to wave w
    const unsigned short* src[6];
    unsigned dst[6];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr3_t)&lds3[0];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = wave * 6 + i, img = q >> 3, blk = q & 7;            // img 0..2: A planes, 3..5: B planes
        const int R = blk * 32 + (lane >> 1), c = (lane & 1) ^ ((R >> 3) & 1);
        // K-blocked planes: element (row, k) of a plane lies at ((k / 16) rows + row) 16 + k % 16 -- the 32 rows x 32 bytes one
        // instruction moves are ONE contiguous KB of memory (row-major planes gave 32-byte pieces 6 KB apart: half of every 64-byte
        // request wasted, the kernel stayed at the fp32 kernel's ~4 TB/s of LDS fills)
        if (img < 3) src[i] = g.A + (int64_t)img * g.pa + ((int64_t)(kbeg / BK3) * g.ra + min(m0 + R, g.M - 1)) * BK3 + 8 * c;
        else src[i] = g.B + (int64_t)(img - 3) * g.pb + ((int64_t)(kbeg / BK3) * g.rb + min(n0 + R, g.N - 1)) * BK3 + 8 * c;
        dst[i] = __builtin_amdgcn_readfirstlane(lds_base + img * IMG3 + blk * 1024);
    }
    const int64_t step_a = (int64_t)g.ra * BK3, step_b = (int64_t)g.rb * BK3;
    auto dma = [&](int buf) {        // (the next stage's transfers; the sources advance by one stage per call)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            lds_dma16_3b(src[i], dst[i] + (unsigned)buf * STAGE3);
            src[i] += ((wave * 6 + i) >> 3) < 3 ? step_a : step_b;
        }
    };

    acc16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[i][j][c] = 0.f;

    // per-lane byte offset of its 16-byte fragment inside a 32-row block of a plane image
    const int loff = r * 32 + 16 * (h ^ ((r >> 3) & 1));
    const int aoff = (wr * 64) * 32 + loff, boff = 3 * IMG3 + (wc * 128) * 32 + loff;

    // three 48 KB buffers: the DMA of stage s + 2 is issued when stage s starts, so a transfer has two stage times to land
    // (with two buffers the wave stalled on `vmcnt(0)` at the end of every stage: 160 TF-equivalent on the dense C4 product)
    const int nst = (kend - kbeg + BK3 - 1) / BK3;
    if (nst > 0) {
        dma(0);
        if (nst > 1) dma(1);
        for (int st = 0; st < nst; ++st) {
            // this wave's transfers of stage st have landed (the 6 of stage st + 1 may still be in flight) ...
            if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();            // ... and every wave's; every wave has also read its fragments of stage st - 1
            if (st + 2 < nst) dma((st + 2) % G3_NBUF);
            const unsigned char* S = lds3 + (st % G3_NBUF) * STAGE3;
            bf16x8 a[2][3], b[4][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const bf16x8*>(S + p * IMG3 + aoff + i * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(S + p * IMG3 + boff + j * 1024);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // smallest terms first: (h l' + m m' + l h'), (h m' + m h'), h h'
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
    }
    // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (c & 3) + 8 (c >> 2) + 4 (lane >> 5)
    const bool out_lower = g.flags & DSVGP_GEMM_OUT_LOWER, atomic = g.splitk > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int m = m0 + wr * 64 + i * 32 + (c & 3) + 8 * (c >> 2) + 4 * h;
                const int n = n0 + wc * 128 + j * 32 + r;
                if (m >= g.M || n >= g.N) continue;
                if (out_lower && n > m) continue;                // (the launcher zero-fills m < n)
                const float v = g.alpha * acc[i][j][c];
                float* dstp = g.C + (int64_t)m * g.ldc + n;
                if (atomic) atomicAdd(dstp, v);
                else *dstp = v;
            }
}

// ---- the split pass: fp32 -> three bf16 planes in the K-BLOCKED layout the GEMM streams: element (row, k) of a plane at
// ((k / 16) rows_out + row) 16 + k % 16, K padded with zeros to a multiple of 16.  One 64 (rows_out) x 64 (k) tile per workgroup
// through LDS: the loads run along the contiguous dimension of src (k for a plain split, rows_out for the transposed one), every
// store instruction of a wave writes one contiguous KB (32 rows x 32 bytes of one 16-deep K block).
__device__ __forceinline__ void split3(float x, __bf16& hh, __bf16& mm, __bf16& ll) {
    hh = (__bf16)x;
    const float r1 = x - (float)hh;
    mm = (__bf16)r1;
    ll = (__bf16)(r1 - (float)mm);
}
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, int64_t ld, int R, int Cc, unsigned short* __restrict__ planes,
                                                     int rows_out, int64_t pstride, int Kp) {
    __shared__ float tile[64][65];                                  // [rows_out][k]
    const int ro0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        if (TRANSPOSE) {                                            // src[k][rows_out]: consecutive threads along rows_out
            const int kk = e >> 6, rr = e & 63;
            tile[rr][kk] = (k0 + kk < R && ro0 + rr < Cc) ? src[(int64_t)(k0 + kk) * ld + ro0 + rr] : 0.f;
        } else {                                                    // src[rows_out][k]: consecutive threads along k
            const int rr = e >> 6, kk = e & 63;
            tile[rr][kk] = (ro0 + rr < R && k0 + kk < Cc) ? src[(int64_t)(ro0 + rr) * ld + k0 + kk] : 0.f;
        }
    }
    __syncthreads();
    // thread -> (K block kb of the tile, row rr, 16-byte half ch): 128 (row, half) pairs per K block, two K blocks per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int kb = pass * 2 + (threadIdx.x >> 7), rr = (threadIdx.x & 127) >> 1, ch = threadIdx.x & 1;
        const int kg = k0 + kb * 16;
        if (kg >= Kp || ro0 + rr >= rows_out) continue;
        bf16x8 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { __bf16 a, b, c; split3(tile[rr][kb * 16 + ch * 8 + e], a, b, c); ph[e] = a; pm[e] = b; pl[e] = c; }
        unsigned short* d = planes + ((int64_t)(kg / 16) * rows_out + ro0 + rr) * 16 + ch * 8;
        *reinterpret_cast<bf16x8*>(d) = ph;
        *reinterpret_cast<bf16x8*>(d + pstride) = pm;
        *reinterpret_cast<bf16x8*>(d + 2 * pstride) = pl;
    }
}

}  // namespace

// K rounded up to the stage depth: the leading dimension of a plane
extern "C" int dsvgp_split3_kpad(int K) { return (K + BK3 - 1) / BK3 * BK3; }
// bytes of the three planes of a [rows_out, K] operand
extern "C" size_t dsvgp_split3_bytes(int rows_out, int K) {
    if (rows_out <= 0 || K <= 0) return 0;
    return (size_t)3 * rows_out * dsvgp_split3_kpad(K) * sizeof(unsigned short) + 256;
}
// planes <- the bf16 triple of src[R, Cc] (transpose = 0: rows_out = R, K = Cc) or of its transpose (1: rows_out = Cc, K = R);
// plane p starts at planes + p * rows_out * kpad(K) elements, K-blocked (element (row, k) at ((k / 16) rows_out + row) 16 + k % 16),
// K padding zero-filled.
extern "C" int dsvgp_split3_bf16(dsvgp_ctx* ctx, const float* src, int64_t ld, int R, int Cc, int transpose, void* planes) {
    if (!ctx || !src || !planes || R <= 0 || Cc <= 0 || ld < Cc || ((uintptr_t)planes & 15)) return DSVGP_EINVAL;
    unsigned short* P = (unsigned short*)planes;
    const int rows_out = transpose ? Cc : R, K = transpose ? R : Cc, Kp = dsvgp_split3_kpad(K);
    const dim3 grid(cdiv(rows_out, 64), cdiv(Kp, 64));
    if (transpose) hipLaunchKernelGGL(split3_kernel<true>, grid, dim3(256), 0, ctx->stream, src, ld, R, Cc, P, rows_out, (int64_t)rows_out * Kp, Kp);
    else hipLaunchKernelGGL(split3_kernel<false>, grid, dim3(256), 0, ctx->stream, src, ld, R, Cc, P, rows_out, (int64_t)rows_out * Kp, Kp);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// C[M, N] = alpha A B^T from plane triples as written by dsvgp_split3_bf16 (A: rows 0..M-1 of an operand split with a_rows rows,
// B: rows 0..N-1 of one split with b_rows rows);
// flags: DSVGP_GEMM_OUT_LOWER (only n <= m computed, the strict upper triangle zero-filled; K is then split over the workgroups).
extern "C" int dsvgp_gemm3b(dsvgp_ctx* ctx, int flags, int M, int N, int K, float alpha, const void* Aplanes, int a_rows, const void* Bplanes,
                            int b_rows, float* C, int64_t ldc) {
    if (!ctx || !Aplanes || !Bplanes || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N || a_rows < M || b_rows < N ||
        (flags & ~DSVGP_GEMM_OUT_LOWER))
        return DSVGP_EINVAL;
    {   // per call: the attribute belongs to the CURRENT device's copy of the kernel (a process-wide "done" flag would leave a second
        // GPU of the process, or a concurrent first call, without it); a host-side table write, no device work
        hipError_t e = hipFuncSetAttribute((const void*)gemm3b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G3_NBUF * STAGE3);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    G3 a{};
    const int Kp = dsvgp_split3_kpad(K);
    a.A = (const unsigned short*)Aplanes; a.B = (const unsigned short*)Bplanes; a.C = C;
    a.ldc = ldc; a.pa = (int64_t)a_rows * Kp; a.pb = (int64_t)b_rows * Kp; a.ra = a_rows; a.rb = b_rows;
    a.M = M; a.N = N; a.K = Kp; a.flags = flags; a.alpha = alpha;
    a.tiles_m = cdiv(M, T3); a.tiles_n = cdiv(N, T3);
    const bool out_lower = flags & DSVGP_GEMM_OUT_LOWER;
    if (out_lower) {
        const int tri = a.tiles_m < a.tiles_n ? a.tiles_m : a.tiles_n;
        a.ntiles = tri * (tri + 1) / 2 + (a.tiles_m > tri ? (a.tiles_m - tri) * a.tiles_n : 0);
    } else {
        a.ntiles = a.tiles_m * a.tiles_n;
    }
    // split-K where the tiles alone would leave CUs idle (the Gram product: 78 tiles, K = 24576): per-CU makespan model
    int sk = 1;
    if (a.ntiles < 2 * G3_SK_SLOTS && Kp >= 1024) {
        double best = 1e300;
        const int maxsk = Kp / 256 < 64 ? Kp / 256 : 64;
        for (int c = 1; c <= maxsk; ++c) {
            const double tcost = (double)cdiv((int64_t)a.ntiles * c, G3_SK_SLOTS) * ((double)Kp / c + 256.0);
            if (tcost < best * 0.999) { best = tcost; sk = c; }
        }
    }
    // accuracy: an accumulator takes six MFMA additions per 16 k, and the bf16 MFMA's internal sum is not the correctly rounded fmaf
    // chain of the fp32 instruction -- over the 8192-long slices the model picks for the Gram product its error was 2-3x that of the
    // fp32 kernel (4.3e-6 against 1.8e-6 of the largest entry).  Long contractions are cut into slices of <= G3_CHAIN_K, whose
    // partial sums meet in fp32 atomics
    if (Kp > 2 * G3_CHAIN_K && sk < cdiv(Kp, G3_CHAIN_K)) sk = cdiv(Kp, G3_CHAIN_K);
    a.kslice = cdiv(cdiv(Kp, sk), BK3) * BK3;
    a.splitk = cdiv(Kp, a.kslice);
    if (a.splitk > 1 || out_lower) {
        hipError_t e = zero_block(C, sizeof(float), ldc, M, N, ctx->stream);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    const dim3 grid(cdiv((int64_t)a.ntiles * a.splitk, 8) * 8);
    hipLaunchKernelGGL(gemm3b_kernel, grid, dim3(512), G3_NBUF * STAGE3, ctx->stream, a);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
