// Blocked right-looking Cholesky of the fp64 K_ZZ (algo 1 of dsvgp_potrf), one fused MFMA launch per 64-column block.
// reference: psd_safe_cholesky(K_ZZ.double()) -> torch.cholesky (DirectionalGradVariationalStrategy.py:72-75).
//
// rocSOLVER's dpotrf takes 8.1 ms at M' = 3000 (serial single-workgroup panel kernels + many small launches); a first
// blocked version here (diagonal-block kernel + panel GEMM + trailing GEMM per block column, 141 launches) took 4.5 ms.
// This version: 48 launches, 1.355 ms at M' = 3000 for the factor WITH the fused inverse (1.62 at 3300, 0.222 at 600; 1.41 / 1.67 / 0.252 with the
// column-by-column chain of the 16 x 16 diagonal sub-blocks, POTRF_F16_BLK = 0; tools/potrf_inv_probe.py; per-launch
// trace tools/potrf_inv_trace.sh; in-kernel stamps of the diagonal workgroup tools/potrf_clock.sh; anatomy in DESIGN.md section 5).
// Measured without effect on the chain (+-1 %, removed again): one Newton step behind v_rsq_f64 instead of two, multipliers folded
// once per column of the 16-column chain, operand reads of the 64^3 products pipelined in chunks, 8-wave workgroups (POTRF_NW).
#include "common.h"

#include <type_traits>

namespace {

constexpr int NBC = 64;

#ifndef POTRF_NW
#define POTRF_NW 4          // waves per workgroup of the step launches: 4; 8 (each 64 x 64 tile product split 32 x 16 per wave) measured slower at M' = 3000 (1.78-1.93 vs 1.70 ms); at 600 5 % faster in round 2, 28 % slower on the round-4 kernel (0.319 vs 0.249 ms)
#endif
#ifndef POTRF_NEWTON
#define POTRF_NEWTON 2
#endif
#ifndef POTRF_FOLD
#define POTRF_FOLD 0        // 1: both rank-1 updates of factor16_wave through w = a_ij / a_jj (10 instead of 16 fp64 operations per column).
#endif                      // Measured (round 4): n = 3000 1.400 / 1.406 -> 1.396 / 1.414 ms, n = 600 0.250 -> 0.252: the column is latency-, not issue-bound
// Update / inverse tiles of one tile row are dealt to workgroups in STRIPS of up to `strip` block columns: T = A_ik W_k is formed
// once per strip (1 + strip products for strip tiles instead of 2 per tile, A_ik and W_k loaded once).  The launcher picks the
// strip length per launch from the tile count: long strips only where the launch is bound by its tiles, not by the diagonal
// workgroup (a strip of 4 lasts about as long as the diagonal workgroup's load -> update -> factor chain).
#ifndef POTRF_STRIP_T6
#define POTRF_STRIP_T6 900      // tiles in the launch above which strips of 6 are used (8: slower, 1.56 vs 1.41 ms at n = 3000)
#endif
#ifndef POTRF_STRIP_T4
#define POTRF_STRIP_T4 512      // ... strips of 4
#endif
#ifndef POTRF_STRIP_T2
#define POTRF_STRIP_T2 256      // ... strips of 2
#endif
#ifndef POTRF_STRIP_V6
#define POTRF_STRIP_V6 6        // strip lengths of the three classes (probes)
#endif
#ifndef POTRF_STRIP_V4
#define POTRF_STRIP_V4 4
#endif
#ifndef POTRF_STRIP_V2
#define POTRF_STRIP_V2 2
#endif
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < POTRF_NEWTON; ++it) {      // y += (y / 2) (1 - d y^2): three dependent operations per step (y / 2 runs beside d y)
        const double h = 0.5 * y;
        y = fma(h, fma(-(d * y), y, 1.0), y);
    }
    return y;
}

// -------------------------------------------------------------------------------------------------
// ONE launch per block column.
//   With X_k = inv(L_kk) and W_k = X_k^T X_k = inv(A_kk) the rank-64 update of step k needs no solved panel:
//       A_ij -= L_ik L_jk^T = (A_ik W_k) A_jk^T                  (i >= j > k)
//   so kernel k does, per 64 x 64 trailing tile, T = A_ik W_k and A_ij -= T A_jk^T out of LDS (column k of A is
//   read-only in that launch: no in-place hazard), and the workgroup of tile (k+1, k+1) goes on to factor its
//   updated tile: L_{k+1,k+1}, X_{k+1}, W_{k+1}.  The solved panels L_ik = A_ik X_k^T are formed afterwards by ONE
//   batched launch.  Critical path per block column: load + 2 products + the in-LDS factorisation.
//
//   In-LDS factorisation of a 64 x 64 tile by 256 threads: 4 x 4 grid of 16 x 16 sub-blocks; the diagonal
//   sub-block is factored AND inverted by ONE wave out of registers (lane = row i, column group g; columns /
//   rows exchanged through 16-entry LDS vectors, pivot by v_readlane: no workgroup barrier on the 16-column
//   chain), the sub-panel, the trailing sub-blocks and the running inverse [L | I] -> [I | X] are 16x16x16
//   fp64 MFMA products.
// -------------------------------------------------------------------------------------------------
using acc4 = double __attribute__((ext_vector_type(4)));
#ifdef POTRF_DEBUG
#ifndef POTRF_DEBUG_K
#define POTRF_DEBUG_K 20      // block column whose critical workgroup is stamped
#endif
#if !defined(POTRF_CRIT_STRIPS) || POTRF_CRIT_STRIPS
#define CHOL_STRIPS_STAMPS 1  // (slots 16 .. 27 belong to the strips chain's own stamps)
#else
#define CHOL_STRIPS_STAMPS 0
#endif
__device__ unsigned long long chol_dbg[64];
// stamps pinned in place (the scalar s_memtime would otherwise be scheduled ahead of the MFMAs it is meant to follow) and kept
// in registers until the end of the kernel (a global store per stamp would sit in vmcnt and stretch the waits that follow it)
#define CHOL_STAMP_DECL unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CHOL_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[slot]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define CHOL_STAMP_FLUSH do { if (b == 0 && k == POTRF_DEBUG_K) { if (tid == 0) { for (int q_ = 0; q_ < 8; ++q_) chol_dbg[q_] = st_[q_]; } \
        if ((tid & 63) == 0 && !CHOL_STRIPS_STAMPS) { chol_dbg[16 + (tid >> 6)] = st_[8]; chol_dbg[20 + (tid >> 6)] = st_[9]; chol_dbg[24 + (tid >> 6)] = st_[10]; } } } while (0)
#else
#define CHOL_STAMP_DECL
#define CHOL_STAMP(slot) do { } while (0)
#define CHOL_STAMP_FLUSH do { } while (0)
#endif
// per-workgroup phase trace of ONE step launch (tools/potrf_wgtrace.py; -DPOTRF_TRACE -DPOTRF_DEBUG_K=k): thread 0 of every workgroup
// stamps s_memtime into an LDS array at phase boundaries and copies it out on exit; slot 22 = role / strip, slot 23 = hardware id
#ifdef POTRF_TRACE
#ifndef POTRF_DEBUG_K
#define POTRF_DEBUG_K 20
#endif
constexpr int TR_SLOTS = 24, TR_MAXWG = 4096;
__device__ unsigned long long chol_trace[TR_MAXWG * TR_SLOTS];
#define TR_DECL __shared__ unsigned long long trs_[TR_SLOTS]; if (threadIdx.x < TR_SLOTS) trs_[threadIdx.x] = 0
#define TR(slot) do { if (k == POTRF_DEBUG_K && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); \
        trs_[slot] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define TR_VAL(slot, v) do { if (k == POTRF_DEBUG_K && threadIdx.x == 0) trs_[slot] = (unsigned long long)(v); } while (0)
#define TR_FLUSH do { if (k == POTRF_DEBUG_K && threadIdx.x == 0 && blockIdx.x < TR_MAXWG) { \
        trs_[23] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); \
        trs_[21] = __builtin_amdgcn_s_memtime(); \
        for (int q_ = 0; q_ < TR_SLOTS; ++q_) chol_trace[(size_t)blockIdx.x * TR_SLOTS + q_] = trs_[q_]; } } while (0)
#define PIPE_TR(slot) do { if (trs && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); \
        trs[slot] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PIPE_TR(slot) do { } while (0)
#define TR_DECL
#define TR(slot) do { } while (0)
#define TR_VAL(slot, v) do { } while (0)
#define TR_FLUSH do { } while (0)
#endif
constexpr int LDT = 66;                       // LDS row stride of a 64 x 64 tile (doubles)

__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// 64 x 64 x 64 product by 4 waves (wave -> 32 x 32 outputs):  acc[i][j] = sum_q Aop[m][q] * Bop(q, n)
//   B_NK: Bop(q, n) = Bs[n][q]   (B given as [n][k]);  else Bop(q, n) = Bs[q][n]
//   ACC_INIT: the caller has loaded the accumulators (acc += product)
template <bool B_NK, int NW = 4, bool ACC_INIT = false>
__device__ __forceinline__ void tile_product(const double (*As)[LDT], const double (*Bs)[LDT], int lane, int wr, int wc,
                                             acc4 (&acc)[2][8 / NW]) {
    constexpr int NJ = 8 / NW;            // 16-column sub-tiles per wave: 2 (4 waves, 32 x 32 each) or 1 (8 waves, 32 x 16 each)
    if constexpr (!ACC_INIT) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = acc4{0, 0, 0, 0};
    }
    // all operands of the wave first, then back-to-back MFMAs: the loops are
    // fully unrolled (a rolled loop moves the accumulators AGPR <-> VGPR and drains the MFMA pipe every trip)
    double a[2][16], b[NJ][16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int kq = 4 * ks + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i][ks] = As[wr * 32 + i * 16 + (lane & 15)][kq];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = wc * (16 * NJ) + j * 16 + (lane & 15);
            b[j][ks] = B_NK ? Bs[n][kq] : Bs[kq][n];
        }
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
}

#ifndef POTRF_IL
#define POTRF_IL 0          // 1: the strip products of the update / inverse tiles issue the NEXT block column's global loads between their MFMAs.
#endif                      // Measured (round 5, profiles/r05_b_potrf_tile_roles.txt): needs POTRF_MINW = 1 (at two waves per SIMD it spills 23 registers:
                            // 1.48 ms); with it n = 3000 1.33 -> 1.27 ms, n = 3300 1.60 -> 1.54, but the STEP does not move (C4 12.81 / 12.78 against
                            // 12.83 / 12.79, C3 7.055 against 7.06: beside the side stream's kernels the launches are bound elsewhere).  Off.
// The same 64 x 64 x 64 product by 4 waves (acc += / = A B), scheduled for a wave that is ALONE on its SIMD (the tile-bound
// launches run one strip workgroup per CU): operand fragments are read two k-steps ahead of the MFMAs that use them, and after
// every k-step (4 MFMAs = 256 cycles of the matrix pipe, of which the SIMD's issue port is busy for ~32) the caller's hook
// pf(ks), ks = 0 .. 15, issues one sixteenth of the NEXT tile's global loads.  With those 32 loads issued in front of the product
// (round 2 .. 4) a strip column cost 7.0k cycles against 4.8k for the last column of a strip, which prefetches nothing
// (profiles/r05_b_potrf_wgtrace_before.txt): the vector-memory issue of one wave, ~70 cycles per load, ran with the matrix pipe idle.
template <bool B_NK, bool ACC_INIT, typename F>
__device__ __forceinline__ void tile_product_il(const double (*As)[LDT], const double (*Bs)[LDT], int lane, int wr, int wc,
                                                acc4 (&acc)[2][2], F&& pf) {
    if constexpr (!ACC_INIT) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = acc4{0, 0, 0, 0};
    }
    const int g = lane >> 4, m = lane & 15;
    double a[16][2], b[16][2];
    auto rd = [&](int ks) {
        const int kq = 4 * ks + g;
#pragma unroll
        for (int i = 0; i < 2; ++i) a[ks][i] = As[wr * 32 + i * 16 + m][kq];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j] = B_NK ? Bs[wc * 32 + j * 16 + m][kq] : Bs[kq][wc * 32 + j * 16 + m];
    };
    rd(0);
    rd(1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + 2 < 16) rd(ks + 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        pf(ks);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // the four fragment reads of k-step ks + 2
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // the four MFMAs of k-step ks
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // two of the next tile's global loads
    }
}

// -------------------------------------------------------------------------------------------------
// Software-pipelined strip (round 5; the PIPE instantiation of chol_step_kernel: ONE workgroup per CU, three LDS tiles).
// A tile-bound launch of the chain is one strip workgroup per CU running  T = A_ik W_k  and then, per block column of its strip,
// C_c -= T B_c^T:  measured (profiles/r05_b_potrf_tile_roles.txt) 9.4k cycles per column of which 4.1k are the product's MFMAs --
// the next column's 32 loads are issued in front of the product (2.2k), its operand goes to LDS between two barriers (1.5k), the
// result is stored behind it (0.9k).  Here all of that rides INSIDE the MFMA stream of the product:
//   product(c)  reads -T (S[0]) and B_c (S[2 - c % 2]); in front of it the wave issues its 8 LDS-DMA instructions of B_{c+1} (into
//               the other B buffer: global -> LDS, no registers, no ds_write); after each of its first 8 k-steps (4 MFMAs) it
//               issues two loads of C_{c+1} and two stores of the result of column c - 1;
//   then ONE drain + barrier closes the column (everything was issued >= 8 k-steps earlier).
// Register sets rotate by NAME, not by copying (a copy of a register a load is still filling waits for the load): a C set is loaded
// in stage c - 1, accumulates in stage c, is stored in stage c + 1 -- three sets; with the two B buffers the loop body is six stages.
// (First version, measured: B rows staged through registers one stage further ahead -- 356 registers, spills whose reloads wait
// vmcnt(0); selects on loaded values became branches around the loads and cut the scheduling regions.)
// An LDS-DMA instruction writes 64 x 16 bytes CONTIGUOUSLY (two rows of a tile), so B's LDS image is rows in PAIRS with a stride of
// 132 doubles per pair, and the 16-byte granules of the odd row of a pair are stored at column c ^ 18 (applied to the SOURCE
// address): bsw() below; the fragment reads of both operand orientations are then conflict-free (32 lanes, 32 distinct bank pairs).
// B_NK: B_c given as [n][k] (update role: rows of A_jk) or [k][n] (inverse role: R_kj).  ident_after: the block column BEHIND the
// ncols pipelined ones has B = I and old C = 0 (inverse role, j == k: always the strip's last): its result is -T itself, stored from
// LDS at the end, no product.  last_diag: the last column is a diagonal tile (only n <= m is stored).  All 64 rows of every C tile
// are stored (callers: interior tile rows only), ldB must be even (16-byte source granules).  On entry: T = the product A_ik W_k,
// S[0] / S[1] still being read by the other waves (the barrier is here), cs[0] = column 0's C elements (requested by the caller),
// B_0 already in S[2] (pipe_dma_tile, issued by the caller before its own loads).
// -------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* potrf_lds_ptr_t;
constexpr int BPAIR = 132;                                  // doubles per row pair of a DMA-filled B image (64 x 66 = 32 x 132)
__device__ __forceinline__ int bsw(int row, int col) { return (row >> 1) * BPAIR + (row & 1) * 64 + (col ^ ((row & 1) * 18)); }
// One LDS-DMA instruction as an asm statement: no VGPR destination, and WE count it (s_waitcnt vmcnt(0) before the barrier that
// ends the stage) -- through the builtin hipcc parks a vmcnt(0) in front of the next LDS read (see gemm32.hip).  M0 (the LDS
// destination base) is compiler-reserved: saved, set and restored inside the statement.
__device__ __forceinline__ void potrf_lds_dma16(const double* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// the 64 x 64 tile at Bt (row stride ldB) into the B image Bs: wave w fills row pairs 8 w .. 8 w + 7
__device__ __forceinline__ void pipe_dma_tile(double (*Bs)[LDT], const double* __restrict__ Bt, int64_t ldB) {
    const int lane = threadIdx.x & 63;
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int par = lane >> 5, col = ((2 * lane) & 63) ^ (par * 18);
    const double* src = Bt + (int64_t)(16 * wave_u + par) * ldB + col;
    const unsigned dst = (unsigned)(uintptr_t)(potrf_lds_ptr_t)&Bs[0][0] + wave_u * (8 * BPAIR * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) potrf_lds_dma16(src + (int64_t)(2 * i) * ldB, dst + i * (BPAIR * 8));
}

// The lane's two base addresses into a B image (the swizzle folded into them, so that every fragment read of a product is
// base + compile-time offset):  B_NK: lane (g, m) reads rows wc 32 + 16 j + m, column 4 ks + g -- for an odd row the column is
// (4 ks) ^ 16 + (g ^ 2): base[0] serves the k-steps with bit 2 clear, base[1] those with it set, + 4 ks + 1056 j doubles;
// [k][n]: rows 4 ks + g, columns wc 32 + 16 j + m -- for an odd row 16 (j ^ 1) + (m ^ 2): base[j], + 264 ks doubles.
template <bool B_NK>
__device__ __forceinline__ void pipe_bases(int lane, int wc, int (&base)[2]) {
    const int g = lane >> 4, m = lane & 15;
    if constexpr (B_NK) {
        const int odd = m & 1, r = ((wc * 32 + m) >> 1) * BPAIR + odd * 64 + (g ^ (2 * odd));
        base[0] = r + 16 * odd;
        base[1] = r - 16 * odd;
    } else {
        const int odd = g & 1, r = (g >> 1) * BPAIR + odd * 64 + wc * 32 + (m ^ (2 * odd));
        base[0] = r + 16 * odd;
        base[1] = r + 16 - 16 * odd;
    }
}

template <bool B_NK, bool HAS_PREV, typename F>
__device__ __forceinline__ void pipe_product(const double (*Ts)[LDT], const double* __restrict__ Bs, const int (&base)[2], int lane,
                                             int wr, double (&accd)[2][2][4], F&& hook) {
    const int g = lane >> 4, m = lane & 15;
    acc4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc4{accd[i][j][0], accd[i][j][1], accd[i][j][2], accd[i][j][3]};
    const double* ta = &Ts[wr * 32 + m][g];
    const double* b0 = Bs + base[0];
    const double* b1 = Bs + base[1];
    double a[16][2], b[16][2];
    auto rd = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[ks][i] = ta[i * 16 * LDT + 4 * ks];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j] = B_NK ? ((ks & 4) ? b1 : b0)[4 * ks + j * 8 * BPAIR] : (j ? b1 : b0)[ks * 2 * BPAIR];
    };
    rd(0);
    rd(1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + 2 < 16) rd(ks + 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        if (ks < 8) hook(ks, std::integral_constant<bool, HAS_PREV>{});
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // fragment reads of k-step ks + 2
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // the four MFMAs of k-step ks
        if (ks < 8) {
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                  // two loads of C_{c+1}
            if (HAS_PREV) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);    // two stores of column c - 1
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) accd[i][j][q] = acc[i][j][q];
}

template <bool B_NK>
__device__ __forceinline__ void strip_pipe(double (*S)[64][LDT], const double* __restrict__ Bg, int64_t ldB, int64_t dB,
                                           double* __restrict__ Cg, int64_t ldC, int ncols, bool ident_after, bool last_diag,
                                           acc4 (&T)[2][2], double (&cs)[3][2][2][4], unsigned long long* trs = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int ml0 = wr * 32 + (lane >> 4), nl0 = wc * 32 + (lane & 15);
    const int64_t step4C = 4 * ldC;
    __syncthreads();                                        // every wave is done reading A_ik / W_k for T
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[0][wr * 32 + i * 16 + (lane >> 4) + 4 * q][wc * 32 + j * 16 + (lane & 15)] = -T[i][j][q];
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the caller's B_0 (LDS-DMA, not counted by the compiler) and C_0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int base[2];
    pipe_bases<B_NK>(lane, wc, base);
    PIPE_TR(2);                                             // (trace: T formed and back in LDS)
    auto stage = [&](auto s_, int c) {
        constexpr int s = decltype(s_)::value;
        double (&acc)[2][2][4] = cs[s % 3];          // (plain doubles, not acc4: element writes into an array of vectors kept it in scratch)
        double (&nxt)[2][2][4] = cs[(s + 1) % 3];
        double (&prev)[2][2][4] = cs[(s + 2) % 3];
        // column c + 1 (clamped to the strip's last column: a harmless re-read at its end)
        const int c1 = min(c + 1, ncols - 1);
        pipe_dma_tile(S[1 + (s & 1)], Bg + (int64_t)c1 * dB, ldB);
        const double* pc = Cg + (int64_t)c1 * 64;
        double* pd = Cg + (int64_t)(c - 1) * 64;
        auto hook = [&](int ks, auto has_prev) {            // ks = 0..7 = the row step t of this thread's C elements
#pragma unroll
            for (int j = 0; j < 2; ++j) nxt[ks >> 2][j][ks & 3] = pc[j * 16];
            pc += step4C;
            if constexpr (decltype(has_prev)::value) {
#pragma unroll
                for (int j = 0; j < 2; ++j) pd[j * 16] = prev[ks >> 2][j][ks & 3];     // (unpredicated: a branch would cut the scheduling region)
                pd += step4C;
            }
        };
        const double* Bcur = &S[2 - (s & 1)][0][0];
        if (c < 5) PIPE_TR(3 + 3 * c);
        if (s != 0 || c > 0) pipe_product<B_NK, true>(S[0], Bcur, base, lane, wr, acc, hook);
        else pipe_product<B_NK, false>(S[0], Bcur, base, lane, wr, acc, hook);
        if (c < 5) PIPE_TR(4 + 3 * c);
        // this wave's share of B_{c+1} has landed (and C_{c+1}, and the stores).  The BUILTIN, not an asm statement: the compiler's own
        // count of its loads restarts here -- otherwise the next stage's first MFMAs wait "vmcnt(3)" for accumulators that are long
        // there, and that wait now includes five of the eight DMA instructions just issued.
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (and one the wait-count pass cannot weaken: the DMA is invisible to it)
        __syncthreads();                                    // B_{c+1} is complete; every wave is done with B_c
        if (c < 5) PIPE_TR(5 + 3 * c);
    };
    // the strip's last column leaves from the stage that computed it (one store loop per register set: choosing the set behind the
    // loop made its address a phi of three allocas, and the sets stayed in scratch)
    auto store_last = [&](double (&res)[2][2][4]) {
        double* pd = Cg + (int64_t)(ncols - 1) * 64;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ml = ml0 + 4 * t, nl = nl0 + j * 16;
                if (!(last_diag && nl > ml)) pd[j * 16] = res[t >> 2][j][t & 3];
            }
            pd += step4C;
        }
    };
    for (int c = 0; c < ncols; c += 6) {
        stage(std::integral_constant<int, 0>{}, c);
        if (c + 1 >= ncols) { store_last(cs[0]); break; }
        stage(std::integral_constant<int, 1>{}, c + 1);
        if (c + 2 >= ncols) { store_last(cs[1]); break; }
        stage(std::integral_constant<int, 2>{}, c + 2);
        if (c + 3 >= ncols) { store_last(cs[2]); break; }
        stage(std::integral_constant<int, 3>{}, c + 3);
        if (c + 4 >= ncols) { store_last(cs[0]); break; }
        stage(std::integral_constant<int, 4>{}, c + 4);
        if (c + 5 >= ncols) { store_last(cs[1]); break; }
        stage(std::integral_constant<int, 5>{}, c + 5);
        if (c + 6 >= ncols) { store_last(cs[2]); break; }
    }
    if (ident_after) {                                      // C_ncols = 0 - T I: this thread's elements of -T, back from LDS
        double* pd = Cg + (int64_t)ncols * 64;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int j = 0; j < 2; ++j) pd[j * 16] = S[0][ml0 + 4 * t][nl0 + j * 16];
            pd += step4C;
        }
    }
}

#ifndef POTRF_F16_BLK
#define POTRF_F16_BLK 1     // 1: the 16-column chain of the diagonal sub-block in four 4-column blocks, rank-4 updates on the fp64 MFMA; 0: column by column
#endif
#if POTRF_F16_BLK
// one wave: Cholesky factor L_d (in place, upper part zeroed) and inverse Xd = L_d^-1 of the 16 x 16 block at F[o.., o..], FOUR
// COLUMNS PER EXCHANGE, the updates as two v_mfma_f64_16x16x4 per block.
//   The column-by-column form (POTRF_F16_BLK = 0) publishes one column and one row through LDS per column and applies a rank-1
//   update from VALU code: 16 write -> read round trips and ~35 instructions per column on the single wave whose chain bounds
//   every launch of the factorisation.  Here, per block of four columns j0 .. j0+3:
//     * every lane publishes ONE element of D (column j0 + g of its row) and one of Y', reads the 4 x 4 pivot block (the same ten
//       numbers in every lane), the four block entries of its own row and the four block rows of Y' in its own column;
//     * factors the pivot block (four dependent rsqrt), substitutes its own row against it -> M[i][0..3] = L[i][j0..j0+3], and
//       eliminates the four rows of Y' inside the block -> Z[0..3];
//     * D -= M M^T and Y' -= (M diag(1/L_kk), strictly below the diagonal) Z are ONE MFMA each, and both take their operands from
//       the lane's own registers: with a[t] = D[i][4t+g] (i = lane & 15, g = lane >> 4) the 16x16x4 A operand wants lane (m, k)
//       to hold M[m][k] and the B operand lane (n, k) to hold M[n][k] -- the same register, M[i][g]; its C/D layout (lane (c, g'),
//       register q <-> element (g' + 4q, c)) is the transposed position of D[c][4q+g'], which a SYMMETRIC update term leaves
//       correct; Y' is kept as yt[t] = Y'[4t+g][i] (row in (t, g), column on the lane), the C/D layout itself.
//   Y': forward elimination of the identity with UNSCALED rows (Y'[r] = L_rr X[r]); scaled by 1 / L_rr at the end.
//   colblk = [16][4] (columns j0 .. j0+3 of D), rowblk = [4][16] (rows j0 .. j0+3 of Y').
//   from_regs (wave-uniform): the block arrives in a_in, a_in[t] = D[i][4t+g] (the critical tile's leading block comes straight out of
//   the MFMA accumulators of its rank-64 update, see crit_tile_update) instead of from F.
__device__ __forceinline__ void factor16_wave(double (*F)[LDT], int o, double (*Xd)[17], double* colblk, double* rowblk,
                                              int lane, int* info, int gidx0, int nvalid, bool from_regs = false,
                                              acc4 a_in = acc4{0, 0, 0, 0}) {
    const int i = lane & 15, g = lane >> 4;
    acc4 a, yt;
    if (from_regs) a = a_in;
    else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = F[o + i][o + 4 * t + g];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) yt[t] = (4 * t + g == i) ? 1.0 : 0.0;
    double rrow[4];                  // 1 / L_rr of rows r = 4t + g
    int bad = -1;                    // first non-positive pivot (wave-uniform)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const int j0 = 4 * jb;
        colblk[i * 4 + g] = a[jb];                      // D[i][j0 + g]
        rowblk[g * 16 + i] = yt[jb];                    // Y'[j0 + g][i]
        __builtin_amdgcn_wave_barrier();
        const double P00 = colblk[(j0 + 0) * 4 + 0];    // (the same addresses in every lane: broadcast reads)
        const double P10 = colblk[(j0 + 1) * 4 + 0], P11 = colblk[(j0 + 1) * 4 + 1];
        const double P20 = colblk[(j0 + 2) * 4 + 0], P21 = colblk[(j0 + 2) * 4 + 1], P22 = colblk[(j0 + 2) * 4 + 2];
        const double P30 = colblk[(j0 + 3) * 4 + 0], P31 = colblk[(j0 + 3) * 4 + 1], P32 = colblk[(j0 + 3) * 4 + 2],
                     P33 = colblk[(j0 + 3) * 4 + 3];
        const double p0 = colblk[i * 4 + 0], p1 = colblk[i * 4 + 1], p2 = colblk[i * 4 + 2], p3 = colblk[i * 4 + 3];
        const double y0 = rowblk[0 * 16 + i], y1 = rowblk[1 * 16 + i], y2 = rowblk[2 * 16 + i], y3 = rowblk[3 * 16 + i];
        __builtin_amdgcn_wave_barrier();
        // the 4 x 4 pivot block
        const double r0 = rsqrt_nr(P00);
        const double l10 = P10 * r0, l20 = P20 * r0, l30 = P30 * r0;
        const double d1 = fma(-l10, l10, P11);
        const double r1 = rsqrt_nr(d1);
        const double l21 = fma(-l20, l10, P21) * r1, l31 = fma(-l30, l10, P31) * r1;
        const double d2 = fma(-l21, l21, fma(-l20, l20, P22));
        const double r2 = rsqrt_nr(d2);
        const double l32 = fma(-l31, l21, fma(-l30, l20, P32)) * r2;
        const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, P33)));
        const double r3 = rsqrt_nr(d3);
        bad = (bad < 0 && !(P00 > 0.0)) ? j0 : bad;
        bad = (bad < 0 && !(d1 > 0.0)) ? j0 + 1 : bad;
        bad = (bad < 0 && !(d2 > 0.0)) ? j0 + 2 : bad;
        bad = (bad < 0 && !(d3 > 0.0)) ? j0 + 3 : bad;
        // this lane's row against the pivot block: M[i][0..3] = L[i][j0 .. j0+3]; it keeps column g (zero above the diagonal)
        const double m0 = p0 * r0;
        const double m1 = fma(-m0, l10, p1) * r1;
        const double m2 = fma(-m1, l21, fma(-m0, l20, p2)) * r2;
        const double m3 = fma(-m2, l32, fma(-m1, l31, fma(-m0, l30, p3))) * r3;
        const double mraw = (g == 0) ? m0 : ((g == 1) ? m1 : ((g == 2) ? m2 : m3));
        const double rg = (g == 0) ? r0 : ((g == 1) ? r1 : ((g == 2) ? r2 : r3));
        const double mg = (i >= j0 + g) ? mraw : 0.0;
        const double sg = (i > j0 + g) ? mraw * rg : 0.0;       // L[i][j0+g] / L[j0+g][j0+g], strictly below the diagonal
        // rows j0 .. j0+3 of Y' after the elimination inside the block, in this lane's column; it keeps row g
        const double w10 = l10 * r0, w20 = l20 * r0, w30 = l30 * r0, w21 = l21 * r1, w31 = l31 * r1, w32 = l32 * r2;
        const double z1 = fma(-w10, y0, y1);
        const double z2 = fma(-w21, z1, fma(-w20, y0, y2));
        const double z3 = fma(-w32, z2, fma(-w31, z1, fma(-w30, y0, y3)));
        const double zg = (g == 0) ? y0 : ((g == 1) ? z1 : ((g == 2) ? z2 : z3));
        a = __builtin_amdgcn_mfma_f64_16x16x4f64(-mg, mg, a, 0, 0, 0);          // D -= M M^T (every column; the block's own ...
        a[jb] = mg;                                                              // ... is final: L)
        yt = __builtin_amdgcn_mfma_f64_16x16x4f64(-sg, zg, yt, 0, 0, 0);        // Y' -= S Z (rows of the block end up as Z)
        rrow[jb] = rg;
    }
    if (lane == 0 && bad >= 0 && o + bad < nvalid && *info == 0) *info = gidx0 + o + bad + 1;   // LAPACK convention
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = 4 * t + g;                                    // column of D held in a[t]; row of Y' held in yt[t]
        F[o + i][o + c] = (c > i) ? 0.0 : a[t];
        Xd[c][i] = (i > c) ? 0.0 : yt[t] * rrow[t];
    }
}
#else
// one wave: Cholesky factor L_d (in place, upper part zeroed) and inverse Xd = L_d^-1 of the 16 x 16 block at F[o.., o..].
// Lane (i = lane & 15, g = lane >> 4) holds a[t] = D[i][4g+t] and y[t] = Y[i][4g+t] (Y: forward elimination of the
// identity, row i scaled by 1/L_ii at the end).  The single wave is instruction-issue bound, so per-element predicates
// are replaced by zeros in the exchanged vectors: column j is published with rows < j zeroed, hence
//   li = L[i][j] = 0 for i < j  and  lc = L[c][j] = 0 for c < j,   a[i][c] -= li * lc   needs no mask
// (rows <= j only collect junk above the diagonal, which is never read), likewise Y[i][:] -= ls * Y[j][:] / L_jj with
// ls = li for i > j, 0 otherwise.
__device__ __forceinline__ void factor16_wave(double (*F)[LDT], int o, double (*Xd)[17], double* colbuf, double* rowbuf,
                                              int lane, int* info, int gidx0, int nvalid, bool = false, acc4 = acc4{0, 0, 0, 0}) {
    const int i = lane & 15, g = lane >> 4;
    double a[4], y[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a[t] = F[o + i][o + 4 * g + t];
        y[t] = (i == 4 * g + t) ? 1.0 : 0.0;
    }
    double ri = 1.0;                 // 1 / L_ii of this lane's row
    int bad = -1;                    // first non-positive pivot (wave-uniform)
    // lanes that do not own the published column / row write to a dummy slot (no exec-mask branches)
    double* const cdst = colbuf + i;
    double* const cdummy = colbuf + 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int gj = j >> 2, tj = j & 3;
        const double d = readlane_f64(a[tj], j + 16 * gj);          // pivot a_jj
        *((g == gj) ? cdst : cdummy) = (i >= j) ? a[tj] : 0.0;
        {
            double* rdst = (i == j) ? (rowbuf + 4 * g) : (rowbuf + 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < 4; ++t) rdst[t] = y[t];
        }
        __builtin_amdgcn_wave_barrier();
        const double ci = colbuf[i];
        double cc[4], rr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { cc[t] = colbuf[4 * g + t]; rr[t] = rowbuf[4 * g + t]; }
        __builtin_amdgcn_wave_barrier();
        bad = (bad < 0 && !(d > 0.0)) ? j : bad;
        const double rinv = rsqrt_nr(d);
        const double li = ci * rinv;                                // L[i][j]  (0 above the diagonal, sqrt(a_jj) on it)
        ri = (i == j) ? rinv : ri;
#if POTRF_FOLD
        // both rank-1 updates through w = L[i][j] / L[j][j] = a_ij / a_jj: ten fp64 operations per column instead of sixteen (the
        // multiplier is folded once per lane instead of once per element)
        const double w = li * rinv;
        const double ws = (i == j) ? 0.0 : w;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = fma(-w, cc[t], a[t]);
            y[t] = fma(-ws, rr[t], y[t]);
        }
#else
        const double ls = (i == j) ? 0.0 : li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = fma(-li, cc[t] * rinv, a[t]);
            y[t] = fma(-ls, rr[t] * rinv, y[t]);
        }
#endif
        if (g == gj) a[tj] = li;
    }
    if (lane == 0 && bad >= 0 && o + bad < nvalid && *info == 0) *info = gidx0 + o + bad + 1;   // LAPACK convention
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = 4 * g + t;
        F[o + i][o + c] = (c > i) ? 0.0 : a[t];
        Xd[i][c] = (c > i) ? 0.0 : y[t] * ri;
    }
}

#endif      // POTRF_F16_BLK

// 16 x 16 x 16 product on one wave:  P[m][n] = sum_q Aop(m, q) Bop(q, n);  operands through pointers + strides
//   Aop(m, q) = Ab[m * lda_ + q];   Bop(q, n) = B_NK ? Bb[n * ldb_ + q] : Bb[q * ldb_ + n]
#ifndef POTRF_CRIT_PFORM
#define POTRF_CRIT_PFORM 1     // 1: the critical tile's rank-64 update through P = A_ik X_k^T (40 + 16 MFMAs on wave 0's path instead of 64 + 16)
#endif
#ifndef POTRF_PROD16_SPLIT
#define POTRF_PROD16_SPLIT 1
#endif
template <bool B_NK>
__device__ __forceinline__ acc4 prod16(const double* Ab, int lda_, const double* Bb, int ldb_, int lane) {
#if POTRF_PROD16_SPLIT
    // two accumulators: one dependent chain of four fp64 MFMAs waits out every instruction's latency (these 16 x 16 x 16 products sit on the
    // serial path of factor64_lds: the panel block and the next diagonal block's update of wave 0)
    acc4 acc0{0, 0, 0, 0}, acc1{0, 0, 0, 0};
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kq = 4 * t + (lane >> 4), r = lane & 15;
        a[t] = Ab[r * lda_ + kq];
        b[t] = B_NK ? Bb[r * ldb_ + kq] : Bb[kq * ldb_ + r];
    }
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
    return acc0 + acc1;
#else
    acc4 acc{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
        const int kq = kk + (lane >> 4), r = lane & 15;
        const double a = Ab[r * lda_ + kq];
        const double b = B_NK ? Bb[r * ldb_ + kq] : Bb[kq * ldb_ + r];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
#endif
}

// Factor F (64 x 64, lower, in LDS) -> L in place; Y <- X = L^-1.  256 threads.  While wave 0 runs the serial 16-column
// chain of the next diagonal sub-block, waves 1..3 do everything that is already final: the other trailing sub-blocks,
// the global stores of row-block kb of X and column-block kb of L, and the running sum W = X^T X = sum_kb X_kb^T X_kb
// (lower blocks, mirrored on store) -- so the critical path is 4 x (factor16 + one panel product + one update product).
//
// Y is never initialised: its 16 x 16 blocks are ASSIGNED at their first touch (block (ib, kb) of the elimination of the identity
// starts as -L_{ib,kb} X_{kb,kb}: the identity has nothing there), blocks above the diagonal are never touched and leave as zeros
// in the stores to Xg.  So Y may alias a tile that other waves are still reading until the barrier behind the first chain
// (the critical tile's update keeps A_ik there, crit_tile_update).
// first_from_regs: wave 0 starts the first 16-column chain from a0 (see factor16_wave) instead of from F.
__device__ __forceinline__ void factor64_lds(double (*F)[LDT], double (*Y)[LDT], double (*Xd)[17], double* colbuf,
                                             double* rowbuf, int tid, int* info, int gidx0, int nvalid,
                                             double* __restrict__ Ag, int64_t lda, double* __restrict__ Xg,
                                             double* __restrict__ Wg, bool first_from_regs = false,
                                             acc4 a0 = acc4{0, 0, 0, 0}) {
    const int lane = tid & 63, wave = tid >> 6;
    const int mrow = (lane >> 4), ncol = lane & 15;
    // (written for waves 0..3; in an 8-wave workgroup waves 4..7 only take part in the barriers)
    // No barrier between the caller's writes of F and the first 16-column chain: wave 0 wrote the 16 x 16 block it starts
    // with ITSELF (the caller guarantees that) or holds it in registers, so it goes straight on while waves 1..3 finish their
    // parts of F; the barrier behind the first chain closes that.
    acc4 wacc[4];                        // waves 1..3: lower blocks idx = (wave - 1) + 3 s of W
#pragma unroll
    for (int sI = 0; sI < 4; ++sI) wacc[sI] = acc4{0, 0, 0, 0};
#pragma unroll 1      // one copy of the 16-column chain: trips 2..4 hit the instruction cache
    for (int kb = 0; kb < 4; ++kb) {
        const int o = kb * 16;
#ifdef POTRF_DEBUG
        if (tid == 0 && kb == 0 && gidx0 == (POTRF_DEBUG_K + 1) * 64) chol_dbg[8] = __builtin_amdgcn_s_memtime();
#endif
        if (wave == 0) factor16_wave(F, o, Xd, colbuf, rowbuf, lane, info, gidx0, nvalid, first_from_regs && kb == 0, a0);
#ifdef POTRF_DEBUG
        if (tid == 0 && kb == 0 && gidx0 == (POTRF_DEBUG_K + 1) * 64) chol_dbg[9] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
        // (b) 4 tasks, one per wave: panel blocks ib > kb, row-block kb of X (cb < kb), and X_{kb,kb} = Xd
        if (wave < 4) {
            const int t = wave;
            if (t < 3 - kb) {                    // L_{ib,kb} = F_{ib,kb} Xd^T
                const int ib = kb + 1 + t;
                acc4 r = prod16<true>(&F[ib * 16][o], LDT, &Xd[0][0], 17, lane);
#pragma unroll
                for (int q = 0; q < 4; ++q) F[ib * 16 + mrow + 4 * q][o + ncol] = r[q];
            } else if (t < 3) {                  // X_{kb,cb} = Xd Y_{kb,cb},  cb = t - (3 - kb)  in [0, kb)
                const int cb = t - (3 - kb);
                acc4 r = prod16<false>(&Xd[0][0], 17, &Y[o][cb * 16], LDT, lane);
#pragma unroll
                for (int q = 0; q < 4; ++q) Y[o + mrow + 4 * q][cb * 16 + ncol] = r[q];
            } else {                             // X_{kb,kb} = Xd
#pragma unroll
                for (int q = 0; q < 4; ++q) Y[o + mrow + 4 * q][o + ncol] = Xd[mrow + 4 * q][ncol];
            }
        }
        __syncthreads();
#ifdef POTRF_DEBUG
        if (tid == 0 && kb == 0 && gidx0 == (POTRF_DEBUG_K + 1) * 64) chol_dbg[10] = __builtin_amdgcn_s_memtime();
#endif
        // (c) trailing sub-blocks:  F_{ib,jb} -= L_{ib,kb} L_{jb,kb}^T (kb < jb <= ib),  Y_{ib,cb} -= L_{ib,kb} X_{kb,cb} (cb <= kb).
        // Look-ahead: wave 0 updates only the NEXT diagonal sub-block (from the panel block it produced itself) and goes
        // straight on to factor it; waves 1..3 touch neither that sub-block nor Xd / colbuf / rowbuf; the barrier after
        // the next factor16 closes the phase.
        if (wave >= 4) {
        } else if (wave == 0) {
            if (kb < 3) {
                const int ib = kb + 1;
                acc4 r = prod16<true>(&F[ib * 16][o], LDT, &F[ib * 16][o], LDT, lane);
#pragma unroll
                for (int q = 0; q < 4; ++q) F[ib * 16 + mrow + 4 * q][ib * 16 + ncol] -= r[q];
            }
        } else {
            int task = 0;
            for (int ib = kb + 1; ib < 4; ++ib) {
                for (int jb = kb + 1; jb <= ib; ++jb) {
                    if (ib == kb + 1 && jb == kb + 1) continue;          // wave 0
                    if (1 + (task++ % 3) != wave) continue;
                    acc4 r = prod16<true>(&F[ib * 16][o], LDT, &F[jb * 16][o], LDT, lane);
#pragma unroll
                    for (int q = 0; q < 4; ++q) F[ib * 16 + mrow + 4 * q][jb * 16 + ncol] -= r[q];
                }
                for (int cb = 0; cb <= kb; ++cb) {
                    if (1 + (task++ % 3) != wave) continue;
                    acc4 r = prod16<false>(&F[ib * 16][o], LDT, &Y[o][cb * 16], LDT, lane);
                    if (cb == kb) {                  // first touch of block (ib, kb): the identity is zero there
#pragma unroll
                        for (int q = 0; q < 4; ++q) Y[ib * 16 + mrow + 4 * q][cb * 16 + ncol] = -r[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) Y[ib * 16 + mrow + 4 * q][cb * 16 + ncol] -= r[q];
                    }
                }
            }
            // final data of this sub-block step: rows o..o+15 of X, columns o..o+15 of L -> global
            const int t3 = tid - 64;
            for (int e = t3; e < 16 * 64; e += 192) {
                const int r = o + (e >> 6), c = e & 63;
                Xg[r * 64 + c] = (c < o + 16) ? Y[r][c] : 0.0;            // (blocks above the diagonal: never touched)
            }
            for (int e = t3; e < (64 - o) * 16; e += 192) {
                const int r = o + (e >> 4), c = o + (e & 15);
                if (r < nvalid && c <= r) Ag[(int64_t)r * lda + c] = F[r][c];
            }
            // W += X_kb^T X_kb on the lower blocks (mb >= nb) owned by this wave; X_{kb,cb} = 0 for cb > kb
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const int idx = (wave - 1) + 3 * sI;
                const int mb = (idx >= 6) ? 3 : ((idx >= 3) ? 2 : ((idx >= 1) ? 1 : 0)), nb = idx - mb * (mb + 1) / 2;
                if (idx < 10 && mb <= kb) {
#pragma unroll
                    for (int kq0 = 0; kq0 < 16; kq0 += 4) {
                        const int kq = o + kq0 + (lane >> 4);
                        wacc[sI] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kq][mb * 16 + ncol], Y[kq][nb * 16 + ncol], wacc[sI], 0, 0, 0);
                    }
                }
            }
        }
#ifdef POTRF_DEBUG
        if (tid == 0 && kb == 0 && gidx0 == (POTRF_DEBUG_K + 1) * 64) chol_dbg[11] = __builtin_amdgcn_s_memtime();
        if (tid == 0 && gidx0 == (POTRF_DEBUG_K + 1) * 64) chol_dbg[12 + kb] = __builtin_amdgcn_s_memtime();      // end of sub-step kb
#endif
    }
    if (wave > 0 && wave < 4) {
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            const int idx = (wave - 1) + 3 * sI;
            const int mb = (idx >= 6) ? 3 : ((idx >= 3) ? 2 : ((idx >= 1) ? 1 : 0)), nb = idx - mb * (mb + 1) / 2;
            if (idx < 10) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mb * 16 + mrow + 4 * q, nn = nb * 16 + ncol;
                    Wg[m * 64 + nn] = wacc[sI][q];
                    Wg[nn * 64 + m] = wacc[sI][q];
                }
            }
        }
    }
}

// Fused inverse (Rw != nullptr): the same launches also run the forward elimination of the identity, block row by
// block row, so that Y = L^-1 is complete when the factorisation is (no separate trtri recursion):
//   R-tiles  (i > k, j <= k):  R_ij -= (A_ik W_k) R_kj,  R_kk = I     (the W_k form of  R_i -= L_ik Y_k,  Y_k = X_k R_k)
//   Y-tiles  (j <= k):         Y_kj  = X_k R_kj                        (row block k of L^-1 is final)
// Row k of R was completed by launch k-1 and is read-only here; these tiles ride on the CUs the latency-bound
// factorisation chain leaves idle.
template <int NW = 4, bool PIPE = false>
__device__ __forceinline__ void chol_inverse_tile(double (*S)[64][LDT], const double* __restrict__ A, int64_t lda, int n,
                                                  int k, int e, const double* __restrict__ Xws,
                                                  const double* __restrict__ Wws, double* __restrict__ Rw, int64_t ldr,
                                                  double* __restrict__ Y, int64_t ldy, int nblk,
                                                  double* __restrict__ YT, int strip = 1
#ifdef POTRF_TRACE
                                                  , unsigned long long* trs_ = nullptr
#endif
                                                  ) {
    constexpr int WC = NW / 2, NJ = 8 / NW, NU = 64 / NW;     // column groups of waves, 16-column sub-tiles per wave, rows per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / WC, wc = wave % WC;
    const int nt = nblk - (k + 1), spr = (k + strip) / strip, nR = nt * spr;     // strips per tile row of R
    const int k0 = k * 64;
    const bool ytile = e >= nR;
    int j = ytile ? (e - nR) : (e % spr) * strip;                    // (first) block column
    const int ncols = ytile ? 1 : min(strip, k + 1 - j);
    const int i0 = ytile ? k0 : (k + 1 + e / spr) * 64;
    int j0 = j * 64;
    const double* Lk = ytile ? (Xws + (size_t)k * 4096) : (Wws + (size_t)k * 4096);
    if constexpr (PIPE && NW == 4) {
        if (!ytile) {                                   // R strip, software-pipelined (R is padded to whole blocks: no clamping there)
            const int ml0 = wr * 32 + (lane >> 4), nl0 = wc * 32 + (lane & 15);
            const int64_t rstep = (int64_t)4 * ldr;
            double cs[3][2][2][4];
            double* const Cg = Rw + (int64_t)(i0 + ml0) * ldr + j0 + nl0;
            const double* const Bg = Rw + (int64_t)k0 * ldr + j0;
            const bool ident_after = k - j < ncols;               // the strip ends with block column j == k: R_kk = I, old R_ik = 0
            const int np = ident_after ? ncols - 1 : ncols;       // block columns that take a product
            {
                double ra[16], rw[16];
                const int64_t arow0 = (int64_t)(i0 + (tid >> 6)) * lda, alast = (int64_t)(n - 1) * lda, astep = (int64_t)4 * lda;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int r = (tid >> 6) + 4 * u, c = tid & 63;
                    ra[u] = A[((i0 + r < n) ? arow0 + u * astep : alast) + k0 + c];
                    rw[u] = Lk[r * 64 + c];
                }
#if !POTRF_PIPE_LATE0
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) cs[0][t >> 2][jj][t & 3] = Cg[t * rstep + jj * 16];
                pipe_dma_tile(S[2], Bg, ldr);                      // B_0 = R_kj of the strip's first column (np == 0: unused); issued last, as above
#endif
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int r = (tid >> 6) + 4 * u;
                    S[0][r][tid & 63] = (i0 + r < n) ? ra[u] : 0.0;
                    S[1][r][tid & 63] = rw[u];
                }
            }
            __syncthreads();
#if POTRF_PIPE_LATE0
            pipe_dma_tile(S[2], Bg, ldr);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) cs[0][t >> 2][jj][t & 3] = Cg[t * rstep + jj * 16];
#endif
            acc4 acc[2][2];
#ifdef POTRF_TRACE
            if (trs_) { TR(1); TR_VAL(20, (ncols << 16) | (e / spr)); }
#endif
            tile_product<true, 4>(S[0], S[1], lane, wr, wc, acc);            // T = A_ik W_k (W symmetric: [n][k] == [k][n])
#ifdef POTRF_TRACE
            strip_pipe<false>(S, Bg, ldr, (int64_t)64, Cg, ldr, np, ident_after, false, acc, cs, k == POTRF_DEBUG_K ? trs_ : nullptr);
#else
            strip_pipe<false>(S, Bg, ldr, (int64_t)64, Cg, ldr, np, ident_after, false, acc, cs);
#endif
            return;
        }
    }
    // two LDS tiles only (two workgroups per CU): R_kj waits in registers until T = A_ik W_k has been formed
    double rc[NU];
    double ro[2][NJ][4];      // R-tile: the old R_ij, requested with the operands (not as 16 load -> wait -> store round trips at the end)
    double* rdst = Rw + (int64_t)(i0 + wr * 32 + (lane >> 4)) * ldr + j0 + wc * (16 * NJ) + (lane & 15);
    {
        double ra[NU], rb[NU];
        const int64_t arow0 = (int64_t)(i0 + (tid >> 6)) * lda, alast = (int64_t)(n - 1) * lda, astep = (int64_t)NW * lda;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int r = (tid >> 6) + NW * u, c = tid & 63;
            ra[u] = ytile ? 0.0 : A[((i0 + r < n) ? arow0 + u * astep : alast) + k0 + c];
            rb[u] = Lk[r * 64 + c];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {           // (behind the operands of the first product)
            const int r = (tid >> 6) + NW * u, c = tid & 63;
            rc[u] = (j == k) ? ((r == c) ? 1.0 : 0.0) : Rw[(int64_t)(k0 + r) * ldr + j0 + c];
        }
        if (!ytile && j != k) {              // (R is padded to whole blocks: no clamping)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
#ifdef POTRF_ABL_NOC
                    for (int q = 0; q < 4; ++q) ro[i][jj][q] = 1e-3 * q;
#else
                    for (int q = 0; q < 4; ++q) ro[i][jj][q] = rdst[(int64_t)(i * 16 + 4 * q) * ldr + jj * 16];
#endif
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ro[i][jj][q] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int r = (tid >> 6) + NW * u, c = tid & 63;
            S[0][r][c] = ytile ? rb[u] : ((i0 + r < n) ? ra[u] : 0.0);      // Y-tile: X_k is the left operand already
            S[1][r][c] = ytile ? rc[u] : rb[u];
        }
    }
    __syncthreads();
#ifdef POTRF_TRACE
    if (trs_) { TR(1); TR_VAL(20, ((ytile ? 0 : ncols) << 16) | (ytile ? 0xffff : (e / spr))); }
#endif
    acc4 acc[2][NJ];
    if (!ytile) {
#if defined(POTRF_ABL_P1) || defined(POTRF_ABL_NOP)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = acc4{1e-3, 1e-3, 1e-3, 1e-3};
#else
#if POTRF_IL
        if constexpr (NW == 4) tile_product_il<true, false>(S[0], S[1], lane, wr, wc, acc, [](int) {});
        else
#endif
        tile_product<true, NW>(S[0], S[1], lane, wr, wc, acc);          // T = A_ik W_k
#endif
#ifdef POTRF_TRACE
        if (trs_) TR(2);
#endif
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {       // -T to LDS, accumulators restart from the old R_ij: the second product leaves R_ij - T R_kj
                    S[0][wr * 32 + i * 16 + (lane >> 4) + 4 * q][wc * (16 * NJ) + jj * 16 + (lane & 15)] = -acc[i][jj][q];
                    acc[i][jj][q] = ro[i][jj][q];
                }
#pragma unroll 1
        for (int cc = 0; cc < ncols; ++cc) {         // the strip: -T stays in S[0], R_kj and the old R_ij change per block column
            if (cc > 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][jj][q] = ro[i][jj][q];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) S[1][(tid >> 6) + NW * u][tid & 63] = rc[u];
            __syncthreads();
#ifdef POTRF_TRACE
            if (trs_ && cc < 5) TR(3 + 3 * cc);
#endif
            double* const rcur = rdst;
#if POTRF_IL && !defined(POTRF_ABL_NOP) && !defined(POTRF_ABL_NOC)
            if constexpr (NW == 4) {
                if (cc + 1 < ncols) {                // next block column: its loads go out BETWEEN the MFMAs of this product
                    ++j; j0 += 64; rdst += 64;
                    if (j == k) {                    // (the identity block of R's row k: nothing to load)
#pragma unroll
                        for (int u = 0; u < NU; ++u) rc[u] = (((tid >> 6) + NW * u) == (tid & 63)) ? 1.0 : 0.0;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                                for (int q = 0; q < 4; ++q) ro[i][jj][q] = 0.0;
                        tile_product_il<false, true>(S[0], S[1], lane, wr, wc, acc, [](int) {});
                    } else {
                        const double* prc = Rw + (int64_t)(k0 + (tid >> 6)) * ldr + j0 + (tid & 63);
                        const double* pro = rdst;
                        const int64_t step4 = (int64_t)4 * ldr;
                        auto pf = [&](int ks) {
                            rc[ks] = *prc;
                            prc += step4;
                            const int t = ks & 7;
                            ro[t >> 2][ks >> 3][t & 3] = pro[(ks >> 3) * 16];
                            pro += (t == 7) ? -7 * step4 : step4;
                        };
                        tile_product_il<false, true>(S[0], S[1], lane, wr, wc, acc, pf);
                    }
                } else {
                    tile_product_il<false, true>(S[0], S[1], lane, wr, wc, acc, [](int) {});
                }
                goto inv_product_done;
            }
#endif
            if (cc + 1 < ncols) {                    // next block column: requested now, consumed after this product
                ++j; j0 += 64; rdst += 64;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int r = (tid >> 6) + NW * u, c = tid & 63;
                    rc[u] = (j == k) ? ((r == c) ? 1.0 : 0.0) : Rw[(int64_t)(k0 + r) * ldr + j0 + c];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#ifdef POTRF_ABL_NOC
                            ro[i][jj][q] = 1e-3 * q;
#else
                            ro[i][jj][q] = (j == k) ? 0.0 : rdst[(int64_t)(i * 16 + 4 * q) * ldr + jj * 16];
#endif
            }
#ifndef POTRF_ABL_NOP
            tile_product<false, NW, true>(S[0], S[1], lane, wr, wc, acc);   // R_ij - T R_kj
#endif
#if POTRF_IL && !defined(POTRF_ABL_NOP) && !defined(POTRF_ABL_NOC)
inv_product_done:
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#ifdef POTRF_ABL_NOC
                        if (acc[i][jj][q] == 123.456)
#endif
                        rcur[(int64_t)(i * 16 + 4 * q) * ldr + jj * 16] = acc[i][jj][q];
#ifdef POTRF_TRACE
            if (trs_ && cc < 5) TR(4 + 3 * cc);
#endif
#ifdef POTRF_TRACE
            if (trs_ && cc < 5) TR(5 + 3 * cc);
#endif
            if (cc + 1 < ncols) __syncthreads();     // every wave is done reading R_kj out of S[1]
        }
    } else {
#if POTRF_IL
        if constexpr (NW == 4) tile_product_il<false, false>(S[0], S[1], lane, wr, wc, acc, [](int) {});
        else
#endif
        tile_product<false, NW>(S[0], S[1], lane, wr, wc, acc);         // X_k R_kj
        if (YT == nullptr) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = k0 + wr * 32 + i * 16 + (lane >> 4) + 4 * q, c = j0 + wc * (16 * NJ) + jj * 16 + (lane & 15);
                        if (m < n && c < n) Y[(int64_t)m * ldy + c] = acc[i][jj][q];
                    }
        } else {
            // the tile also goes out transposed (L^-T for the forward solves, which stream an mn-contiguous left operand):
            // through LDS, so that both images are written in full rows -- no separate transpose pass over L^-1
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        S[0][wr * 32 + i * 16 + (lane >> 4) + 4 * q][wc * (16 * NJ) + jj * 16 + (lane & 15)] = acc[i][jj][q];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int r = (tid >> 6) + NW * u, c = tid & 63;
                if (k0 + r < n && j0 + c < n) Y[(int64_t)(k0 + r) * ldy + j0 + c] = S[0][r][c];
                if (j0 + r < n && k0 + c < n) YT[(int64_t)(j0 + r) * ldy + k0 + c] = S[0][c][r];
            }
        }
    }
}

#ifndef POTRF_CRIT_SLIVER
#define POTRF_CRIT_SLIVER 1     // 1: the critical tile's rank-64 update in 16-row strips out of registers (crit_tile_update); 0: as every other tile
#endif
// The critical workgroup's rank-64 update  D = C - (A W) A^T  of the NEXT diagonal tile (A = A_{k+1,k}, W = W_k, C = A_{k+1,k+1}),
// ordered for the serial chain that follows it instead of for throughput.  Every other tile forms T = A W with four waves of
// 32 x 32, sends -T through LDS and runs a second 64^3 product (two barriers, ~10.6k cycles + 3.3k of LDS moves) before the
// factorisation can start.  Here wave w owns the 16-row strip w of the tile:
//   * U = T_w^T = W A_w^T (64 x 16: four accumulator blocks, K = 64).  Its C/D register layout (lane (c, g), register q <->
//     U[g + 4q][c] = T_w[c][g + 4q]) IS the A-operand layout of T_w for the next product -- T never goes through LDS;
//   * D[w][nb] = C[w][nb] - T_w A_nb^T only for the blocks nb <= w on or below the diagonal (16 / 32 / 48 / 64 MFMAs for wave 0..3);
//   * wave 0 therefore holds the leading 16 x 16 block after 64 + 16 MFMAs and starts the 16-column chain OUT OF ITS REGISTERS
//     (it loads C transposed, C[n][g + 4q]: the accumulator then holds D^T, whose register layout is the one factor16_wave wants,
//     lower triangle from valid data); waves 1..3 finish their strips in the shadow of that chain and meet it at its barrier.
// LDS: S[0] = A (read until the end of the second product, then the inverse's work tile Y -- never initialised, see factor64_lds),
// S[1] = W (dead after the one barrier between the two products, then F).  Returns wave 0's leading block in a0.
__device__ __forceinline__ void crit_tile_update(double (*S)[64][LDT], const double* __restrict__ A, int64_t lda, int n, int k,
                                                 const double* __restrict__ Wk, int tid, acc4& a0
#ifdef POTRF_DEBUG
                                                 , unsigned long long* st_
#endif
                                                 ) {
    const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int k0 = k * 64, i0 = (k + 1) * 64;
    const int64_t alast = (int64_t)(n - 1) * lda;
    acc4 cacc[4];
    {
        double ra[16], rw[16];
        const int64_t arow0 = (int64_t)(i0 + (tid >> 6)) * lda, astep = (int64_t)4 * lda;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = (tid >> 6) + 4 * u, c = tid & 63;
            rw[u] = Wk[r * 64 + c];
            ra[u] = A[((i0 + r < n) ? arow0 + u * astep : alast) + k0 + c];
        }
        // this strip's blocks of C, requested behind the operands and consumed after the first product
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb > wave) { cacc[nb] = acc4{0, 0, 0, 0}; continue; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // wave 0 (its one block is the tile's leading diagonal block): transposed, element (row g + 4q, column i) <- C[i][g + 4q]
                const int rr = (wave == 0) ? i : 16 * wave + g + 4 * q, cc = (wave == 0) ? g + 4 * q : 16 * nb + i;
                const int gr = min(i0 + rr, n - 1), gc = min(i0 + cc, n - 1);
                cacc[nb][q] = A[(int64_t)gr * lda + gc];
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = (tid >> 6) + 4 * u, c = tid & 63;
            S[0][r][c] = (i0 + r < n) ? ra[u] : 0.0;
            S[1][r][c] = rw[u];
        }
    }
    __syncthreads();
    CHOL_STAMP(1);
    // U = W A_w^T: A operand W[16 cb + m][k] (lane (m, k)), B operand A_w[n][k] (lane (n, k)); four independent accumulators
    // (POTRF_CRIT_PFORM: S[1] holds X_k instead -- A_ik W_k A_ik^T = (A_ik X_k^T)(A_ik X_k^T)^T -- and U = X A_w^T = P_w^T: X_k is lower
    //  triangular, so column block cb of P needs the k blocks 0 .. cb only: 40 MFMAs instead of 64)
    acc4 U[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) U[cb] = acc4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int kq = 4 * ks + g;
        const double bop = S[0][16 * wave + i][kq];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (POTRF_CRIT_PFORM && cb < (ks >> 2)) continue;
            U[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1][16 * cb + i][kq], bop, U[cb], 0, 0, 0);
        }
    }
    CHOL_STAMP(4);
    __syncthreads();                 // W is dead: S[1] becomes F
#if POTRF_CRIT_PFORM
    // P replaces A in S[0] (every wave is done with A): strip w's rows from this wave's accumulators (register q of lane (i, g) = P[16 w + i][16 cb + 4 q + g]);
    // the second products below then read P_nb where they read A_nb
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) S[0][16 * wave + i][16 * cb + 4 * q + g] = U[cb][q];
    __syncthreads();
#endif
    CHOL_STAMP(5);
    // D[w][nb] = C[w][nb] - T_w A_nb^T, T_w from the registers of U; the leading block (wave 0) on two accumulators (one
    // dependent chain of 16 fp64 MFMAs would wait out every instruction's latency)
    if (wave == 0) {
        acc4 d0 = cacc[0], d1 = acc4{0, 0, 0, 0};
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double bop = S[0][i][16 * cb + 4 * q + g];
                if (q & 1) d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[cb][q], bop, d1, 0, 0, 0);
                else d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[cb][q], bop, d0, 0, 0, 0);
            }
        CHOL_STAMP(7);
        // (symmetric update term: the transposed C makes this D^T) a0[t] = D[i][4t + g] for 4t + g <= i, zero above, identity padding
        // of a ragged last block
        const int nr = n - i0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = 4 * t + g;
            const bool in = i < nr && c <= i;
            a0[t] = in ? d0[t] + d1[t] : ((i == c && i >= nr) ? 1.0 : 0.0);
        }
    } else {
        double (*F)[LDT] = S[1];
        const int nr = n - i0;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb > wave) continue;
            acc4 d = cacc[nb];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[cb][q], S[0][16 * nb + i][16 * cb + 4 * q + g], d, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ml = 16 * wave + g + 4 * q, nl = 16 * nb + i;
                const bool in = ml < nr && nl <= ml;
                F[ml][nl] = in ? d[q] : ((ml == nl) ? 1.0 : 0.0);
            }
        }
    }
}

#ifndef POTRF_CRIT_STRIPS
#define POTRF_CRIT_STRIPS 1     // 1: the critical workgroup keeps the diagonal tile in REGISTERS, one 16-row strip per wave (factor64_strips below); 0: round 5's factor64_lds
#endif
#if POTRF_CRIT_STRIPS
// -------------------------------------------------------------------------------------------------
// Round 6: the critical workgroup's factorisation with the diagonal tile in registers and the inverse OFF the serial chain.
//
// Round 5 (factor64_lds): per 16-column block, ONE wave factors AND inverts the 16 x 16 diagonal block (factor16_wave, 4.4-4.7k cycles),
// then two barrier phases form the panel L_{ib,kb} = F_{ib,kb} X_d^T and the next diagonal block's update -- 6.7k cycles per block,
// 26.5k for the tile (of the critical workgroup's 37.6k).  The panel needed X_d only because it was written as an MFMA product.
// Here wave w owns rows 16 w .. 16 w + 15 of the tile for the WHOLE factorisation, in the register image
//     a[cb][t] = D[16 w + i][16 cb + 4 t + g]      (i = lane & 15, g = lane >> 4;  cb <= w)
// and block column kb is processed by all waves w >= kb together, four columns at a time:
//   * the DIAGONAL wave (w = kb) publishes the 16 x 4 column block of its strip, factors the 4 x 4 pivot block (four dependent
//     rsqrt), substitutes its rows against it and applies the rank-4 update D -= M M^T with ONE MFMA out of its own registers
//     (as factor16_wave did) -- but runs no forward elimination of the identity beside it: L only;
//   * the FOLLOWERS (w > kb) read the published block, redo the pivot factorisation (the same ten numbers in every lane), substitute
//     their own rows AND the diagonal block's rows (the B / A operands of the update D_w -= M_w M_d^T, computed transposed so that the
//     accumulator IS the a image), one block behind the diagonal wave and never waited for by it: the panel L_{w,kb} comes out of
//     the same substitution the diagonal block gets -- no inverse on the path;
//   * after the 16 columns wave w holds L_{w,kb}; it publishes the block (LDS, register image as it stands: readers use the same
//     lane map), applies  D_{w,cb} -= L_{w,kb} L_{cb,kb}^T  for cb = kb + 1 .. w (own block: both operands are its own registers),
//     and wave kb + 1 goes straight on as the next diagonal wave.
// Critical path: 4 x (16-column chain + 4 MFMAs), hopping from wave to wave.  X = L^-1 and W = X^T X are formed by the waves that
// are done with the chain: wave w inverts its diagonal block from registers (invert16_regs: the forward elimination of the old
// chain, now beside the NEXT wave's chain), wave c < r forms T_{r,c} = sum_j L_{r,j} X_{j,c} as soon as its inputs exist and
// X_{r,c} = -X_rr T_{r,c} when wave r has published X_rr; every wave accumulates its own blocks of W row by row.
// Synchronisation: LDS flags (one writer each, polled), no workgroup barrier after the update phase.  LDS: xs (packed lower
// blocks of X, row stride XB) and cpub (published column blocks) in the tile that held X_k; lpub (published L blocks) in the tile
// that holds P = A X_k^T, written only once every wave has left the update phase (FL_P flags).
// tools/crit_chain_emulator.py replays this routine lane by lane on the host (index algebra check).
// -------------------------------------------------------------------------------------------------
#ifndef POTRF_SPIN_SLEEP
#define POTRF_SPIN_SLEEP 1      // s_sleep argument between two polls of a flag word (0: none; measured: n = 600 0.194 -> 0.188 ms with 1, nothing more with 4)
#endif
constexpr int XB = 17;                                        // row stride (doubles) of a 16 x 16 block of X in LDS
constexpr int FL_C = 0, FL_P = 4, FL_L = 8, FL_X = 24, FL_N = 40;      // flag words: cflag[kb], pdone[w], lflag[r][j], xflag[r][c]
// (explicit LDS pointers: through generic pointers the volatile flag words became FLAT stores / loads with a vmcnt(0) wait behind each)
typedef __attribute__((address_space(3))) double LD;
typedef __attribute__((address_space(3))) volatile int LVI;
struct CritLds {
    LD* xs;               // 10 x 16 x XB
    LD* cpub;         // [4 kb][4 jb][80]: the pivot factors (10 words) and, from word 16, the block M of the diagonal wave, per 4-column block
    LD* lpub;             // [6][4 t][64 lanes]
    LD* own;              // [4 waves][64]
    LVI* fl;              // FL_N words, zeroed before use
#ifdef POTRF_DEBUG
    int dbg_gidx0;
    __attribute__((address_space(3))) unsigned long long* dbg;      // 64 stamp words (LDS: a stamp is one s_memtime + one ds_write of lane 0), copied to chol_dbg at the end
#endif
};
#ifdef POTRF_DEBUG
#define STRIPS_STAMP(slot) do { if (gidx0 == (POTRF_DEBUG_K + 1) * 64) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (lane == 0) S.dbg[slot] = t__; } } while (0)
#else
#define STRIPS_STAMP(slot) do { } while (0)
#endif
__device__ __forceinline__ int xblk(int r, int c) { return 16 * XB * (r * (r + 1) / 2 + c); }
__device__ __forceinline__ int lblk(int r, int j) { return 256 * (r * (r - 1) / 2 + j); }
// one writer per flag; LDS executes a wave's instructions in order, so the flag store behind the data stores needs no wait
__device__ __forceinline__ void fl_set(LVI* f, int v, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) *f = v;
}
__device__ __forceinline__ void fl_wait(LVI* f, int v, int* info) {
    int n = 0;
    while (*f < v) {
        if (++n > (1 << 22)) {            // (never in a correct run: every flag is set unconditionally; a bounded spin cannot hang the card)
            if (info) *info = -777;
            break;
        }
#if POTRF_SPIN_SLEEP
        __builtin_amdgcn_s_sleep(POTRF_SPIN_SLEEP);
#endif
    }
    asm volatile("" ::: "memory");
}
struct Piv4 { double r0, r1, r2, r3, l10, l20, l30, l21, l31, l32; };
#ifndef POTRF_RSQ_HALLEY
#define POTRF_RSQ_HALLEY 1      // 1: v_rsq_f64 + ONE third-order step (5 instructions); 0: two Newton steps (8 instructions, rsqrt_nr)
#endif
// 1 / sqrt(d) for the serial chain.  A single wave issues one fp64 instruction per ~5 cycles and a dependent one waits no longer than that
// (tools/lat_probe.cpp: dependent v_fma_f64 5.3, v_rsq_f64 16 cycles): the chain is bound by its INSTRUCTION COUNT.  v_rsq_f64 is good to
// 2.5e-8, so with e = 1 - d y^2 the step y (1 + e / 2 + 3 e^2 / 8) leaves 5 e^3 / 16 ~ 5e-24: full accuracy from five instructions.
__device__ __forceinline__ double rsqrt_chain(double d) {
#if POTRF_RSQ_HALLEY
    const double y = __builtin_amdgcn_rsq(d);
    const double gq = d * y;
    const double e = fma(-gq, y, 1.0);
    const double pq = fma(e, 0.375, 0.5);
    const double ye = y * e;
    return fma(ye, pq, y);
#else
    return rsqrt_nr(d);
#endif
}
// the 4 x 4 pivot block (lower entries P_rc): reciprocal square roots of the pivots and the block's L entries
template <bool CHECK>
__device__ __forceinline__ Piv4 pivot4(double P00, double P10, double P11, double P20, double P21, double P22, double P30, double P31, double P32,
                                       double P33, int j0, int& bad) {
    Piv4 v;
    v.r0 = rsqrt_chain(P00);
    v.l10 = P10 * v.r0; v.l20 = P20 * v.r0; v.l30 = P30 * v.r0;
    const double d1 = fma(-v.l10, v.l10, P11);
    v.r1 = rsqrt_chain(d1);
    v.l21 = fma(-v.l20, v.l10, P21) * v.r1; v.l31 = fma(-v.l30, v.l10, P31) * v.r1;
    const double d2 = fma(-v.l21, v.l21, fma(-v.l20, v.l20, P22));
    v.r2 = rsqrt_chain(d2);
    v.l32 = fma(-v.l31, v.l21, fma(-v.l30, v.l20, P32)) * v.r2;
    const double d3 = fma(-v.l32, v.l32, fma(-v.l31, v.l31, fma(-v.l30, v.l30, P33)));
    v.r3 = rsqrt_chain(d3);
    if constexpr (CHECK) {
        bad = (bad < 0 && !(P00 > 0.0)) ? j0 : bad;
        bad = (bad < 0 && !(d1 > 0.0)) ? j0 + 1 : bad;
        bad = (bad < 0 && !(d2 > 0.0)) ? j0 + 2 : bad;
        bad = (bad < 0 && !(d3 > 0.0)) ? j0 + 3 : bad;
    }
    return v;
}
// one row's four block entries against the pivot block: M[row][0..3]; returns column g
__device__ __forceinline__ double subst4(double p0, double p1, double p2, double p3, const Piv4& v, int g) {
    const double m0 = p0 * v.r0;
    const double m1 = fma(-m0, v.l10, p1) * v.r1;
    const double m2 = fma(-m1, v.l21, fma(-m0, v.l20, p2)) * v.r2;
    const double m3 = fma(-m2, v.l32, fma(-m1, v.l31, fma(-m0, v.l30, p3))) * v.r3;
    asm volatile("" :: "v"(m1), "v"(m2), "v"(m3));      // (all four in every lane: left to itself hipcc sinks them under nested exec-mask branches on g)
    return (g == 0) ? m0 : ((g == 1) ? m1 : ((g == 2) ? m2 : m3));
}
typedef double __attribute__((ext_vector_type(2))) d2v;      // (a builtin vector: HIP's double2 class cannot be copied out of address space 3 in the host pass)
typedef __attribute__((address_space(3))) d2v LD2;
// The diagonal wave of block column kb (ONE copy of this code serves the four block columns: the wave that runs it for kb + 1 finds it in
// the instruction cache -- as four template instances every chain started with ~1k cycles of instruction fetch, profiles/r06_a_*).
// cur = D_kk -> L_kk (zero above the diagonal), rks[jb] = the four 1 / L_cc of block jb (the same in every lane).  Per 4-column block: the
// strip's column block goes through the wave's own LDS words once (row i's four entries for the substitution; rows j0 .. j0+3 are the pivot
// block, read as broadcasts), P_00 alone comes by v_readlane so that the first 1 / sqrt starts at once; the pivot factors (cpub[kb][jb][0..9])
// and the block M (the followers' A operand, lane for lane: cpub[..][16 + lane]) are published with ONE flag per block (cflag = jb + 1; the last block:
// 4 behind the factors, 5 behind M),
// behind the MFMA of the rank-4 update (the publication rides under its latency).
__device__ __forceinline__ void strips_diag(const CritLds& S, int kb, acc4& cur, acc4 (&rks)[4], int lane, int* info, int gidx0, int nvalid) {
    const int i = lane & 15, g = lane >> 4;
    LD* own = S.own + 64 * kb;
    int unused = -1;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const int j0 = 4 * jb;
        LD* cb = S.cpub + (kb * 4 + jb) * 80;
        own[i * 4 + g] = cur[jb];                                    // D[i][j0 + g]
        __builtin_amdgcn_wave_barrier();
        const double P00 = readlane_f64(cur[jb], j0);
        const d2v p01 = *(const LD2*)(own + i * 4), p23 = *(const LD2*)(own + i * 4 + 2);
        const d2v R1 = *(const LD2*)(own + (j0 + 1) * 4), R2 = *(const LD2*)(own + (j0 + 2) * 4), R3 = *(const LD2*)(own + (j0 + 3) * 4),
                      R3b = *(const LD2*)(own + (j0 + 3) * 4 + 2);
        const double P22 = own[(j0 + 2) * 4 + 2];
        __builtin_amdgcn_sched_barrier(0);                           // (the exchange is issued HERE: its latency rides under the first 1 / sqrt)
        const Piv4 v = pivot4<false>(P00, R1.x, R1.y, R2.x, R2.y, P22, R3.x, R3.y, R3b.x, R3b.y, j0, unused);
        if (lane == 0) {
            cb[0] = v.r0; cb[1] = v.r1; cb[2] = v.r2; cb[3] = v.r3; cb[4] = v.l10; cb[5] = v.l20; cb[6] = v.l30; cb[7] = v.l21; cb[8] = v.l31;
            cb[9] = v.l32;
        }
        if (jb == 3) fl_set(S.fl + FL_C + kb, 4, lane);              // (the followers' last block needs the factors only: the next chain starts ~0.4k earlier)
        const double mraw = subst4(p01.x, p01.y, p23.x, p23.y, v, g);
        const double mg = (i >= j0 + g) ? mraw : 0.0;
        if (jb < 3) cur = __builtin_amdgcn_mfma_f64_16x16x4f64(-mg, mg, cur, 0, 0, 0);      // D -= M M^T (the last block has nothing behind it)
        cb[16 + lane] = mg;
        fl_set(S.fl + FL_C + kb, jb < 3 ? jb + 1 : 5, lane);         // (5: the last M block is out too -- the wave that forms the inverse reads it)
        cur[jb] = mg;
        rks[jb] = acc4{v.r0, v.r1, v.r2, v.r3};
        STRIPS_STAMP(32 + 4 * kb + jb);                              // (end of block jb)
    }
    // a pivot <= 0 (or NaN) makes its 1 / sqrt and everything behind it NaN: ONE test of the last one per chain; the column is looked up only then
    const double rl = rks[3][3];
    if (!(rl < 1.7e308)) {
        int bad = 15;
#pragma unroll
        for (int jb = 3; jb >= 0; --jb)
#pragma unroll
            for (int q = 3; q >= 0; --q)
                if (!(rks[jb][q] < 1.7e308)) bad = 4 * jb + q;
        if (lane == 0 && 16 * kb + bad < nvalid && *info == 0) *info = gidx0 + 16 * kb + bad + 1;      // LAPACK convention
    }
}
// A follower of block column kb: cur = D_{w,kb} -> L_{w,kb}; dg = this wave's own diagonal block D_ww.  It substitutes its own rows against the
// published pivot factors and applies, per 4-column block,
//   D_w^T -= M_d M_w^T   (A operand lane (m, k) = M_d[m][k]: the published block, lane for lane; B operand lane (n, k) = M_w[n][k]; register q of
//                         lane (n, g') receives row g' + 4 q of M_d against row n of M_w = the update of D_w[n][4 q + g']: the a image)
//   D_ww  -= M_w M_w^T   (both operands the same register; the trailing update of the wave's own diagonal block, four columns at a time: when
//                         the sixteenth column is in, the next diagonal wave starts its chain after ONE more MFMA)
// The flag poll and the factor reads of block jb + 1 are issued behind the MFMAs of block jb, in front of the exchange of the updated column
// block (which has to wait for them): a follower that has caught up is ~250 cycles behind the publication it waits for.
__device__ __forceinline__ void strips_follow(const CritLds& S, int kb, acc4& cur, acc4& dg, int wave, int lane, int* info) {
    const int i = lane & 15, g = lane >> 4;
    LD* own = S.own + 64 * wave;
    own[i * 4 + g] = cur[0];                                         // this wave's D[i][j0 + g], exchanged inside the wave
    __builtin_amdgcn_wave_barrier();
    d2v p01 = *(const LD2*)(own + i * 4), p23 = *(const LD2*)(own + i * 4 + 2);
    __builtin_amdgcn_sched_barrier(0);
    const LD* cb = S.cpub + kb * 4 * 80;
    d2v f0, f1, f2, f3, f4;
    double ng;
    // the flag word and what it guards in ONE LDS round trip: the reads are issued behind the flag read and LDS executes a wave's instructions in
    // order, so values read behind a flag that reads "set" are the published ones; otherwise the whole batch is repeated
    auto fetch = [&](int want) {
        int n = 0;
        for (;;) {
            const int fv = *(S.fl + FL_C + kb);
            asm volatile("" ::: "memory");
            f0 = *(const LD2*)(cb + 0); f1 = *(const LD2*)(cb + 2); f2 = *(const LD2*)(cb + 4); f3 = *(const LD2*)(cb + 6); f4 = *(const LD2*)(cb + 8);
            ng = cb[16 + lane];
            asm volatile("" ::: "memory");
            if (fv >= want || ++n > (1 << 22)) break;
#if POTRF_SPIN_SLEEP
            __builtin_amdgcn_s_sleep(POTRF_SPIN_SLEEP);
#endif
        }
    };
    fetch(1);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
#ifdef POTRF_DEBUG
        if (kb == 1 && wave == 2) { const int gidx0 = S.dbg_gidx0; STRIPS_STAMP(16 + jb); }      // (factors of block jb read)
#endif
        Piv4 v;
        v.r0 = f0.x; v.r1 = f0.y; v.r2 = f1.x; v.r3 = f1.y; v.l10 = f2.x; v.l20 = f2.y; v.l30 = f3.x; v.l21 = f3.y; v.l31 = f4.x; v.l32 = f4.y;
        const double mw = subst4(p01.x, p01.y, p23.x, p23.y, v, g);  // M_w[i][g]
        if (jb < 3) cur = __builtin_amdgcn_mfma_f64_16x16x4f64(-ng, mw, cur, 0, 0, 0);
        dg = __builtin_amdgcn_mfma_f64_16x16x4f64(-mw, mw, dg, 0, 0, 0);
        if (jb < 3) {
            cb += 80;
            fetch(jb + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur[jb] = mw;
        if (jb < 3) {
            own[i * 4 + g] = cur[jb + 1];
            __builtin_amdgcn_wave_barrier();
            p01 = *(const LD2*)(own + i * 4); p23 = *(const LD2*)(own + i * 4 + 2);
        }
#ifdef POTRF_DEBUG
        if (kb == 1 && wave == 2) { const int gidx0 = S.dbg_gidx0; STRIPS_STAMP(20 + jb); }      // (block jb done)
#endif
    }
}
// The inverse of the diagonal block of column kb, formed by an IDLE wave one 4-column block behind the diagonal wave (round 5 ran this
// forward elimination of the identity inside the chain; as a pass of its own behind the last chain it was 2.8k cycles of the tile's tail):
// Y' -= (M diag(1 / L_cc), strictly below the diagonal) Z per block, M and the pivot factors as the diagonal wave published them;
// returns x[t] = X[4 t + g][i], the MFMA result layout.
__device__ __forceinline__ acc4 strips_invfollow(const CritLds& S, int kb, int wave, int lane, int* info) {
    const int i = lane & 15, g = lane >> 4;
    LD* own = S.own + 64 * wave;
    acc4 yt, rrx;
#pragma unroll
    for (int t = 0; t < 4; ++t) yt[t] = (4 * t + g == i) ? 1.0 : 0.0;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const int j0 = 4 * jb;
        own[g * 16 + i] = yt[jb];                                    // Y'[j0 + g][i]
        __builtin_amdgcn_wave_barrier();
        const double y0 = own[0 * 16 + i], y1 = own[1 * 16 + i], y2 = own[2 * 16 + i], y3 = own[3 * 16 + i];
        __builtin_amdgcn_sched_barrier(0);
        const LD* cb = S.cpub + (kb * 4 + jb) * 80;
        fl_wait(S.fl + FL_C + kb, jb < 3 ? jb + 1 : 5, info);
        const d2v f0 = *(const LD2*)(cb + 0), f1 = *(const LD2*)(cb + 2), f2 = *(const LD2*)(cb + 4), f3 = *(const LD2*)(cb + 6), f4 = *(const LD2*)(cb + 8);
        const double mgv = cb[16 + lane];                            // M[i][g] = L[i][j0 + g] (zero above the diagonal)
        const double r0 = f0.x, r1 = f0.y, r2 = f1.x, r3 = f1.y;
        const double w10 = f2.x * r0, w20 = f2.y * r0, w30 = f3.x * r0, w21 = f3.y * r1, w31 = f4.x * r1, w32 = f4.y * r2;
        const double z1 = fma(-w10, y0, y1);
        const double z2 = fma(-w21, z1, fma(-w20, y0, y2));
        const double z3 = fma(-w32, z2, fma(-w31, z1, fma(-w30, y0, y3)));
        asm volatile("" :: "v"(z1), "v"(z2), "v"(z3));
        const double zg = (g == 0) ? y0 : ((g == 1) ? z1 : ((g == 2) ? z2 : z3));
        const double rg = (g == 0) ? r0 : ((g == 1) ? r1 : ((g == 2) ? r2 : r3));
        const double sg = (i > j0 + g) ? mgv * rg : 0.0;             // L[i][j0+g] / L[j0+g][j0+g], strictly below the diagonal
        yt = __builtin_amdgcn_mfma_f64_16x16x4f64(-sg, zg, yt, 0, 0, 0);
        rrx[jb] = rg;
    }
    acc4 x;
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = (i > 4 * t + g) ? 0.0 : yt[t] * rrx[t];
    return x;
}
// X_ww = L_ww^-1 from the register image of L_ww (a[t] = L[i][4 t + g], zero above the diagonal; rr[t] = 1 / L_cc, c = 4 t + g):
// forward elimination of the identity with unscaled rows, four rows per exchange (the Y' half of factor16_wave); returns
// x[t] = X[4 t + g][i], the MFMA result layout
__device__ __forceinline__ acc4 invert16_regs(LD* own, const acc4& a, const acc4& rr, int lane) {
    const int i = lane & 15, g = lane >> 4;
    acc4 yt;
#pragma unroll
    for (int t = 0; t < 4; ++t) yt[t] = (4 * t + g == i) ? 1.0 : 0.0;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const int j0 = 4 * jb;
        own[g * 16 + i] = yt[jb];                                    // Y'[j0 + g][i]
        __builtin_amdgcn_wave_barrier();
        const double y0 = own[0 * 16 + i], y1 = own[1 * 16 + i], y2 = own[2 * 16 + i], y3 = own[3 * 16 + i];
        __builtin_amdgcn_wave_barrier();
        // L[j0 + r][j0 + c] sits in a[jb] of lane (i = j0 + r, g = c); 1 / L_cc of column j0 + k in rr[jb] of lane (., g = k)
        const double l10 = readlane_f64(a[jb], j0 + 1), l20 = readlane_f64(a[jb], j0 + 2), l30 = readlane_f64(a[jb], j0 + 3);
        const double l21 = readlane_f64(a[jb], j0 + 2 + 16), l31 = readlane_f64(a[jb], j0 + 3 + 16), l32 = readlane_f64(a[jb], j0 + 3 + 32);
        const double r0 = readlane_f64(rr[jb], 0), r1 = readlane_f64(rr[jb], 16), r2 = readlane_f64(rr[jb], 32);
        const double w10 = l10 * r0, w20 = l20 * r0, w30 = l30 * r0, w21 = l21 * r1, w31 = l31 * r1, w32 = l32 * r2;
        const double z1 = fma(-w10, y0, y1);
        const double z2 = fma(-w21, z1, fma(-w20, y0, y2));
        const double z3 = fma(-w32, z2, fma(-w31, z1, fma(-w30, y0, y3)));
        const double zg = (g == 0) ? y0 : ((g == 1) ? z1 : ((g == 2) ? z2 : z3));
        const double sg = (i > j0 + g) ? a[jb] * rr[jb] : 0.0;       // L[i][j0+g] / L[j0+g][j0+g], strictly below the diagonal
        yt = __builtin_amdgcn_mfma_f64_16x16x4f64(-sg, zg, yt, 0, 0, 0);
    }
    acc4 x;
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = (i > 4 * t + g) ? 0.0 : yt[t] * rr[t];
    return x;
}
// block (r, c) of X from its result-layout registers: LDS image [row][XB] and the 64 x 64 row-major tile in global memory
__device__ __forceinline__ void put_x(const CritLds& S, double* __restrict__ Xg, int r, int c, const acc4& x, int lane) {
    const int i = lane & 15, g = lane >> 4;
    LD* b = S.xs + xblk(r, c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        b[(g + 4 * q) * XB + i] = x[q];
        Xg[(16 * r + g + 4 * q) * 64 + 16 * c + i] = x[q];
    }
    fl_set(S.fl + FL_X + 4 * r + c, 1, lane);
}
// one block column: the diagonal wave factors, the waves below follow, publish their panel block and update their trailing blocks
// (kb is a run-time value: one copy of the code; the strip's blocks are read and written as WHOLE vectors under wave-uniform branches --
// element writes into an array of vectors keep it in scratch)
__device__ __forceinline__ void strips_step(const CritLds& S, int kb, acc4 (&a)[4], acc4 (&rks)[4], int wave, int lane, int* info, int gidx0,
                                            int nvalid) {
    if (wave < kb) return;
    acc4 cur = a[0];
    if (kb == 1) cur = a[1];
    if (kb == 2) cur = a[2];
    if (kb == 3) cur = a[3];
    if (wave == kb) {
        STRIPS_STAMP(48 + kb);                                       // (this chain starts)
        strips_diag(S, kb, cur, rks, lane, info, gidx0, nvalid);
        STRIPS_STAMP(12 + kb);
    } else {
        acc4 dg = (wave == 1) ? a[1] : ((wave == 2) ? a[2] : a[3]);  // this wave's own diagonal block
        strips_follow(S, kb, cur, dg, wave, lane, info);
        if (wave == 1) a[1] = dg;
        else if (wave == 2) a[2] = dg;
        else a[3] = dg;
        if (wave == kb + 1) STRIPS_STAMP(52 + kb);                   // (the next diagonal wave has its panel block and its updated diagonal block)
        if (kb == 0) {                                               // lpub lives in the P tile: every wave must have left the update phase
            typedef int __attribute__((ext_vector_type(4))) i4v;
            typedef __attribute__((address_space(3))) volatile i4v LVI4;
            int n = 0;
            for (;;) {                                               // (the four FL_P words in one read)
                const i4v pd = *(LVI4*)(S.fl + FL_P);
                if ((pd.x & pd.y & pd.z & pd.w) != 0 || ++n > (1 << 22)) break;
            }
            asm volatile("" ::: "memory");
        }
        LD* lp = S.lpub + lblk(wave, kb);
#pragma unroll
        for (int t = 0; t < 4; ++t) lp[t * 64 + lane] = cur[t];
        fl_set(S.fl + FL_L + 4 * wave + kb, 1, lane);
        if (kb == 1 && wave == 2) STRIPS_STAMP(24);                  // (panel block published)
        // the blocks between: D_{w,cb} -= L_{w,kb} L_{cb,kb}^T for kb < cb < wave, in the a image: A operand = L_{cb,kb} (published image), B
        // operand = the own L_{w,kb}; two accumulators per block (dependent fp64 MFMAs wait out their latency)
#pragma unroll
        for (int cb = 1; cb < 3; ++cb) {
            if (cb <= kb || cb >= wave) continue;
            fl_wait(S.fl + FL_L + 4 * cb + kb, 1, info);
            const LD* lq = S.lpub + lblk(cb, kb);
            acc4 d0 = a[cb], d1 = acc4{0, 0, 0, 0};
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[0 * 64 + lane], -cur[0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[1 * 64 + lane], -cur[1], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[2 * 64 + lane], -cur[2], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[3 * 64 + lane], -cur[3], d1, 0, 0, 0);
            a[cb] = d0 + d1;
        }
    }
    if (kb == 0) a[0] = cur;
    else if (kb == 1) a[1] = cur;
    else if (kb == 2) a[2] = cur;
    else a[3] = cur;
}
// the W blocks (mb >= nb) a wave accumulates: rows of X at or below its own only (wave v's blocks have mb >= v)
__device__ __forceinline__ void w_block(int wave, int s, int& mb, int& nb) {
    //  wave 0: (0,0) (1,0) (3,0)    wave 1: (1,1) (2,0) (3,1)    wave 2: (2,1) (2,2)    wave 3: (3,2) (3,3)
    mb = -1; nb = 0;
    if (wave == 0) { mb = (s == 0) ? 0 : ((s == 1) ? 1 : 3); nb = 0; }
    else if (wave == 1) { mb = (s == 0) ? 1 : ((s == 1) ? 2 : 3); nb = (s == 1) ? 0 : 1; }
    else if (wave == 2) { if (s < 2) { mb = 2; nb = 1 + s; } }
    else { if (s < 2) { mb = 3; nb = 2 + s; } }
}
// The whole tile: a[cb] (cb <= wave) = this wave's strip of the updated diagonal tile (identity padding of a ragged tile), 256 threads.
// Ag: the tile's place in the matrix (lower triangle of its nvalid valid rows stored), Xg / Wg: X = L^-1 and W = X^T X, 64 x 64 each.
__device__ __forceinline__ void factor64_strips(const CritLds& S, acc4 (&a)[4], int tid, int* info, int gidx0, int nvalid,
                                                double* __restrict__ Ag, int64_t lda, double* __restrict__ Xg, double* __restrict__ Wg) {
    const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    acc4 rks[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rks[t] = acc4{1, 1, 1, 1};
    if (wave == 0) STRIPS_STAMP(8);
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) strips_step(S, kb, a, rks, wave, lane, info, gidx0, nvalid);
    // ---- the rest of the tile's work, in the order of urgency for the waves that wait on it.  Wave 0 inverts its own diagonal block (nobody
    // is idle beside its chain); wave w < 3 forms the inverse of the NEXT diagonal block one 4-column block behind that chain.
    if (wave == 0) {
        acc4 rr;                                                  // rr[t] = 1 / L_cc of column c = 4 t + g
#pragma unroll
        for (int t = 0; t < 4; ++t) rr[t] = (g == 0) ? rks[t][0] : ((g == 1) ? rks[t][1] : ((g == 2) ? rks[t][2] : rks[t][3]));
        const acc4 xw = invert16_regs(S.own, a[0], rr, lane);
        put_x(S, Xg, 0, 0, xw, lane);
    }
    // the zero blocks of X to the right of this wave's diagonal block, and COLUMN block `wave` of L: the diagonal block from the registers,
    // the blocks below it from the images their waves published (those waves are busy with the next chains; wave 3 has only its own block,
    // stored at the very end)
    auto store_l = [&](acc4 lv, int rb, int cbk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 16 * rb + i, c = 16 * cbk + 4 * t + g;
            if (r < nvalid && c <= r) Ag[(int64_t)r * lda + c] = lv[t];
        }
    };
    if (wave < 3) {
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            if (c <= wave) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) Xg[(16 * wave + g + 4 * q) * 64 + 16 * c + i] = 0.0;
        }
        // (the diagonal block picked under wave-uniform branches with compile-time block indices: a block picked at run time -- a select of
        //  references, or of the loaded vectors -- puts the whole array in memory)
        if (wave == 0) store_l(a[0], 0, 0);
        else if (wave == 1) store_l(a[1], 1, 1);
        else store_l(a[2], 2, 2);
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            if (r <= wave) continue;
            fl_wait(S.fl + FL_L + 4 * r + wave, 1, info);
            const LD* lq = S.lpub + lblk(r, wave);
            acc4 lv;
#pragma unroll
            for (int t = 0; t < 4; ++t) lv[t] = lq[t * 64 + lane];
            store_l(lv, r, wave);
        }
        // T_{w+1,w} = L_{w+1,w} X_ww BEFORE the inverse of the next diagonal block is followed (both operands exist soon after this wave's own
        // chain; for wave 2 this is T_{3,2}: X_{3,2} then leaves four MFMAs after X_33, not 2k cycles)
        fl_wait(S.fl + FL_X + 5 * wave, 1, info);                    // X_ww (wave w - 1 formed it; wave 0 its own)
        fl_wait(S.fl + FL_L + 4 * (wave + 1) + wave, 1, info);
        acc4 tn0 = acc4{0, 0, 0, 0}, tn1 = acc4{0, 0, 0, 0};
        {
            const LD* lq = S.lpub + lblk(wave + 1, wave);
            const LD* xq = S.xs + xblk(wave, wave);
            tn0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[0 * 64 + lane], xq[(0 + g) * XB + i], tn0, 0, 0, 0);
            tn1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[1 * 64 + lane], xq[(4 + g) * XB + i], tn1, 0, 0, 0);
            tn0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[2 * 64 + lane], xq[(8 + g) * XB + i], tn0, 0, 0, 0);
            tn1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[3 * 64 + lane], xq[(12 + g) * XB + i], tn1, 0, 0, 0);
        }
        const acc4 Tn = tn0 + tn1;
        if (wave == 2) STRIPS_STAMP(58);                             // (T_{3,2} formed)
        const acc4 xn = strips_invfollow(S, wave + 1, wave, lane, info);
        put_x(S, Xg, wave + 1, wave + 1, xn, lane);
        if (wave == 2) STRIPS_STAMP(9);
        {   // X_{w+1,w} = -X_{w+1,w+1} Tn
            const LD* xr = S.xs + xblk(wave + 1, wave + 1);
            acc4 x0 = acc4{0, 0, 0, 0}, x1 = acc4{0, 0, 0, 0};
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 0 + g], Tn[0], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 4 + g], Tn[1], x1, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 8 + g], Tn[2], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 12 + g], Tn[3], x1, 0, 0, 0);
            put_x(S, Xg, wave + 1, wave, x0 + x1, lane);
            if (wave == 2) STRIPS_STAMP(57);                         // (X_{3,2} published)
        }
    }
    // ---- X_{r,w} for the rows below (what the other waves wait for): T_{r,c} = sum_{j = c}^{r-1} L_{r,j} X_{j,c} (A operand lane (m, k) =
    // L_{r,j}[m][4 t + k], the published register image; B operand lane (n, k) = X_{j,c}[4 t + k][n]), then X_{r,c} = -X_rr T (A operand lane
    // (m, k) = X_rr[m][4 t + k], B operand = register t of T: its result layout).  Rows 1, 2 in full; of row 3 first T only -- X_33 is the last
    // thing the tile produces, the blocks of W that do not need it come in between.
    acc4 wacc[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) wacc[s] = acc4{0, 0, 0, 0};
    auto w_row = [&](int r) {       // W_{mb,nb} += X_{r,mb}^T X_{r,nb} for this wave's blocks with mb <= r: ONE poll for the row (xflag[r][0..3] in one
                                    // read: a poll is an LDS round trip, ~100 cycles), the blocks' MFMAs interleaved (independent accumulators)
        typedef int __attribute__((ext_vector_type(4))) i4v;
        typedef __attribute__((address_space(3))) volatile i4v LVI4;
        int n = 0;
        for (;;) {
            const i4v xf = *(LVI4*)(S.fl + FL_X + 4 * r);
            const int have = (xf.x != 0) + (r >= 1 ? (xf.y != 0) : 1) + (r >= 2 ? (xf.z != 0) : 1) + (r >= 3 ? (xf.w != 0) : 1);
            if (have == 4 || ++n > (1 << 22)) break;
#if POTRF_SPIN_SLEEP
            __builtin_amdgcn_s_sleep(POTRF_SPIN_SLEEP);
#endif
        }
        asm volatile("" ::: "memory");
        double wa[3][4], wb[3][4];
        bool on[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            int mb, nb;
            w_block(wave, s, mb, nb);
            on[s] = mb >= 0 && mb <= r;
            const LD* xa = S.xs + xblk(r, on[s] ? mb : 0);
            const LD* xb = S.xs + xblk(r, on[s] ? nb : 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wa[s][t] = xa[(4 * t + g) * XB + i];
                wb[s][t] = xb[(4 * t + g) * XB + i];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (on[s]) wacc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[s][t], wb[s][t], wacc[s], 0, 0, 0);      // (wave-uniform)
    };
#pragma unroll
    for (int r = 2; r < 4; ++r) {
        if (r <= wave + 1) continue;                                 // (row wave + 1: above)
        const int c = wave;
        acc4 t0 = acc4{0, 0, 0, 0}, t1 = acc4{0, 0, 0, 0};
        {   // ONE poll for the panel blocks L_{r,c..r-1} (lflag[r][0..3] in one read); X_{j,c}, j > c, are this wave's own earlier rows
            typedef int __attribute__((ext_vector_type(4))) i4v;
            typedef __attribute__((address_space(3))) volatile i4v LVI4;
            int n = 0;
            for (;;) {
                const i4v lf = *(LVI4*)(S.fl + FL_L + 4 * r);
                const int have = ((c > 0) | (lf.x != 0)) + ((c > 1 || r <= 1) | (lf.y != 0)) + ((r <= 2) | (lf.z != 0));
                if (have == 3 || ++n > (1 << 22)) break;
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < c || j >= r) continue;
            const LD* lq = S.lpub + lblk(r, j);
            const LD* xq = S.xs + xblk(j, c);
            t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[0 * 64 + lane], xq[(0 + g) * XB + i], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[1 * 64 + lane], xq[(4 + g) * XB + i], t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[2 * 64 + lane], xq[(8 + g) * XB + i], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lq[3 * 64 + lane], xq[(12 + g) * XB + i], t1, 0, 0, 0);
        }
        const acc4 T = t0 + t1;
        if (r == 3) {
            if (wave == 0) STRIPS_STAMP(60);                         // (T_{3,0} formed)
            if (wave == 1) STRIPS_STAMP(59);
#pragma unroll
            for (int rr_ = 0; rr_ < 3; ++rr_) if (rr_ >= wave) w_row(rr_);      // (waves 0, 1: the rows of W that do not need X_33, while chain 3 runs)
        }
        fl_wait(S.fl + FL_X + 4 * r + r, 1, info);
        if (r == 3 && wave == 0) STRIPS_STAMP(61);                   // (X_33 seen)
        const LD* xr = S.xs + xblk(r, r);
        acc4 x0 = acc4{0, 0, 0, 0}, x1 = acc4{0, 0, 0, 0};
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 0 + g], T[0], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 4 + g], T[1], x1, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 8 + g], T[2], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr[i * XB + 12 + g], T[3], x1, 0, 0, 0);
        put_x(S, Xg, r, c, x0 + x1, lane);
        if (r == 3 && wave == 0) STRIPS_STAMP(62);                   // (X_{3,0} published)
        if (r == 3 && wave == 1) STRIPS_STAMP(56);
    }
    if (wave == 2) w_row(2);
    if (wave == 3) STRIPS_STAMP(25);
    w_row(3);
    if (wave == 0) STRIPS_STAMP(63);                                 // (row 3 of W taken by wave 0)
    if (wave == 3) STRIPS_STAMP(26);                                 // (... by wave 3)
    if (wave == 3) STRIPS_STAMP(27);                                 // (wave 3: its W blocks accumulated)
    // W blocks -> the 64 x 64 tile, both triangles.  The mirrored block goes through LDS (the cpub words are free now: every reader of them is
    // behind the row-3 flags), so that its stores run along rows too -- stored from the registers, one instruction touched 64 cache lines and
    // the 12 of them were ~2k cycles at the very end of the critical workgroup
    LD* tw = S.cpub + 272 * wave;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int mb, nb;
        w_block(wave, s, mb, nb);
        if (mb < 0) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = 16 * mb + g + 4 * q, nn = 16 * nb + i;
            Wg[m * 64 + nn] = wacc[s][q];
            tw[(g + 4 * q) * XB + i] = wacc[s][q];
        }
        __builtin_amdgcn_wave_barrier();
        if (mb != nb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Wg[(16 * nb + g + 4 * q) * 64 + 16 * mb + i] = tw[i * XB + g + 4 * q];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (wave == 3) {
        const acc4 l33 = a[3];
        store_l(l33, 3, 3);
    }
    STRIPS_STAMP(28 + wave);                                         // this wave is done
#ifdef POTRF_DEBUG
    if (gidx0 == (POTRF_DEBUG_K + 1) * 64) {                         // the stamp words of this wave's slots -> chol_dbg (slots 8 .. 63; 0 .. 7 belong to CHOL_STAMP)
        __syncthreads();
        if (tid >= 8 && tid < 64) chol_dbg[tid] = S.dbg[tid];
    }
#endif
}

// The update phase for factor64_strips: as crit_tile_update (P = A X_k^T strip by strip, P back to LDS, D = C - P P^T), but EVERY wave's
// blocks come out in the a image: block (w, nb) as the transposed product  D^T = C^T - P_nb P_w^T  (A operand: P_nb out of LDS, B operand:
// the registers of U = P_w^T), C loaded in the a image.  k < 0: the first tile of the matrix, loaded as it is.
__device__ __forceinline__ void crit_strips_update(double (*S)[64][LDT], const double* __restrict__ A, int64_t lda, int n, int k,
                                                   const double* __restrict__ Xk, int tid, acc4 (&a)[4], LVI* fl
#ifdef POTRF_DEBUG
                                                   , unsigned long long* st_
#endif
                                                   ) {
    const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int i0 = (k + 1) * 64, nr = n - i0;
    if (tid < FL_N) fl[tid] = 0;
    // this strip's blocks of C in the a image (clamped addresses, masked below): requested BEHIND the operands of the first product
    // (loads return in order: in front of them they held up the LDS fill by 1.5k cycles)
    auto load_c = [&]() {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb > wave) { a[nb] = acc4{0, 0, 0, 0}; continue; }
            acc4 cv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gr = min(i0 + 16 * wave + i, n - 1), gc = min(i0 + 16 * nb + 4 * q + g, n - 1);
                cv[q] = A[(int64_t)gr * lda + gc];
            }
            a[nb] = cv;
        }
    };
    if (k >= 0) {
        const int k0 = k * 64;
        {
            double ra[16], rw[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {           // (rows past the matrix: clamped row index, zeroed on the way to LDS -- no branch per load)
                const int r = (tid >> 6) + 4 * u, c = tid & 63;
                rw[u] = Xk[r * 64 + c];
                ra[u] = A[(int64_t)min(i0 + r, n - 1) * lda + k0 + c];
            }
            load_c();
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = (tid >> 6) + 4 * u, c = tid & 63;
                S[0][r][c] = (i0 + r < n) ? ra[u] : 0.0;
                S[1][r][c] = rw[u];
            }
        }
        __syncthreads();
        CHOL_STAMP(1);
        // U = X_k A_w^T = P_w^T (X_k lower triangular: column block cb of P needs the k blocks 0 .. cb only)
        acc4 U[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) U[cb] = acc4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int kq = 4 * ks + g;
            const double bop = S[0][16 * wave + i][kq];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (cb < (ks >> 2)) continue;
                U[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1][16 * cb + i][kq], bop, U[cb], 0, 0, 0);
            }
        }
        CHOL_STAMP(4);
        __syncthreads();                 // X_k is dead: its tile becomes xs / cpub
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[0][16 * wave + i][16 * cb + 4 * q + g] = U[cb][q];
        __syncthreads();
        CHOL_STAMP(5);
        if (wave == 0) {                 // one block: two accumulators
            acc4 d1 = acc4{0, 0, 0, 0};
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double aop = S[0][i][16 * cb + 4 * q + g];
                    if (q & 1) d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, -U[cb][q], d1, 0, 0, 0);
                    else a[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, -U[cb][q], a[0], 0, 0, 0);
                }
            a[0] += d1;
        } else {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        if (nb > wave) continue;
                        a[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0][16 * nb + i][16 * cb + 4 * q + g], -U[cb][q], a[nb], 0, 0, 0);
                    }
        }
        CHOL_STAMP(7);
    } else {
        load_c();
        __syncthreads();                 // (the flag words are zero for every wave)
    }
    fl_set(fl + FL_P + wave, 1, lane);   // this wave is done with the P tile
    // identity padding of a ragged tile, zero above the diagonal
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb > wave) continue;
        acc4 mv = a[nb];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * wave + i, c = 16 * nb + 4 * q + g;
            const bool in = r < nr && c <= r;
            mv[q] = in ? mv[q] : ((r == c) ? 1.0 : 0.0);
        }
        a[nb] = mv;
    }
}
#endif      // POTRF_CRIT_STRIPS

#ifndef POTRF_MINW
#define POTRF_MINW (POTRF_NW / 2)
#endif
#ifndef POTRF_PIPE_MIN_TILES
#define POTRF_PIPE_MIN_TILES 300    // launches with fewer tiles are bound by the critical workgroup anyway: they keep the two-per-CU kernel
#endif
#ifndef POTRF_PIPE_WGS
#define POTRF_PIPE_WGS 250          // workgroups of a PIPE launch (one per CU; the critical workgroup is one of them)
#endif
#ifndef POTRF_PIPE_LATE0
#define POTRF_PIPE_LATE0 1          // 1: column 0 of a pipelined strip (B_0 by LDS-DMA, C_0) is requested behind the barrier in front of the T product
#endif
#ifndef POTRF_PIPE_FROM
#define POTRF_PIPE_FROM 0           // (A/B builds) first block column that may use the PIPE kernel, on top of the caller's pipe_from
#endif
#ifndef POTRF_PIPE_MINW
#define POTRF_PIPE_MINW 1           // the PIPE instantiation: 101 KB of LDS, one workgroup per CU whatever the registers (asking for 2 here to get a
                                    // 256-register budget does not take: the compiler sees the LDS size and budgets for one wave per SIMD anyway)
#endif
#ifndef POTRF_PIPE
#define POTRF_PIPE 1        // 1: launches from `pipe_from` on (launch_potrf_blocked) use the PIPE instantiation: software-pipelined strips (strip_pipe)
#endif
// PIPE: three LDS tiles and up to 512 registers -- ONE workgroup per CU -- with the strips of both tile roles software-pipelined
// (strip_pipe above); strip_e: strip length of the ragged last tile row of the update role, which keeps the two-barrier strip loop
// (its tiles need clamped addresses) and must not be the launch's longest chain.
template <int NW, bool PIPE = false>
__global__ __launch_bounds__(64 * NW, PIPE ? POTRF_PIPE_MINW : POTRF_MINW) void chol_step_kernel(double* __restrict__ A, int64_t lda, int n, int k,
                                                           double* __restrict__ Xws, double* __restrict__ Wws,
                                                           int* __restrict__ info, double* __restrict__ Rw, int64_t ldr,
                                                           double* __restrict__ Yinv, int64_t ldy, int nA,
                                                           double* __restrict__ YinvT, int strip, int strip_e) {
    // TWO 64 x 64 LDS tiles (70 KB with the factorisation scratch): two workgroups share a CU, which halves the rounds
    // the ~1100 update / inverse tiles of a mid-chain launch need.  The second right operand of every tile waits in
    // registers while the first product runs.
    __shared__ double S[PIPE ? 3 : 2][64][LDT];
    __shared__ double Xd[16][17];
    __shared__ double colbuf[64], rowbuf[64];      // column-by-column chain: [16..31] dummy slots of the non-owner lanes; 4-column blocks: [16][4] / [4][16]
    constexpr int WC = NW / 2, NJ = 8 / NW, NU = 64 / NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / WC, wc = wave % WC;
    const int b = blockIdx.x;
    TR_DECL;
    TR(0);
    if (b >= nA) {                                 // fused-inverse tiles (only launched with k >= 0 and Rw != nullptr)
#ifdef POTRF_TRACE
        chol_inverse_tile<NW, PIPE>(S, A, lda, n, k, b - nA, Xws, Wws, Rw, ldr, Yinv, ldy, (n + 63) / 64, YinvT, strip, trs_);
        TR_VAL(22, 2);
        TR_FLUSH;
#else
        chol_inverse_tile<NW, PIPE>(S, A, lda, n, k, b - nA, Xws, Wws, Rw, ldr, Yinv, ldy, (n + 63) / 64, YinvT, strip);
#endif
        return;
    }
#if POTRF_CRIT_STRIPS && POTRF_CRIT_PFORM
    if (NW == 4 && b == 0) {                       // the critical workgroup (k = -1: the matrix's first tile, a launch of its own)
        CHOL_STAMP_DECL;
        CHOL_STAMP(0);
        acc4 a[4];
        CritLds cl;
        cl.xs = (LD*)&S[1][0][0];                  // (the X_k tile: dead after the first product of the update)
        cl.cpub = cl.xs + 10 * 16 * XB;
        cl.lpub = (LD*)&S[0][0][0];                // (the P tile: written only behind the FL_P flags)
        cl.own = (LD*)&Xd[0][0];
        cl.fl = (LVI*)colbuf;
#ifdef POTRF_DEBUG
        cl.dbg = (__attribute__((address_space(3))) unsigned long long*)rowbuf;
        if (tid < 64) cl.dbg[tid] = 0;
        cl.dbg_gidx0 = (k + 1) * 64;
#endif
        static_assert(10 * 16 * XB + 16 * 80 <= 64 * LDT && 4 * 272 <= 16 * 80 && 6 * 256 <= 64 * LDT && 4 * 64 <= 16 * 17 && FL_N * sizeof(int) <= 64 * sizeof(double),
                      "LDS views of the critical workgroup");
#ifdef POTRF_DEBUG
        crit_strips_update(S, A, lda, n, k, Xws + (size_t)(k < 0 ? 0 : k) * 4096, tid, a, cl.fl, st_);
        st_[6] = st_[5];
#else
        crit_strips_update(S, A, lda, n, k, Xws + (size_t)(k < 0 ? 0 : k) * 4096, tid, a, cl.fl);
#endif
        CHOL_STAMP(2);
        TR(1);
        const int kk = k + 1, r0 = kk * 64, nr = (n - r0 < 64) ? (n - r0) : 64;
        factor64_strips(cl, a, tid, info, r0, nr, A + (int64_t)r0 * lda + r0, lda, Xws + (size_t)kk * 4096, Wws + (size_t)kk * 4096);
        CHOL_STAMP(3);
        CHOL_STAMP_FLUSH;
        TR(2);
        TR_VAL(22, 0);
        TR_FLUSH;
        return;
    }
#elif POTRF_CRIT_SLIVER
    if (NW == 4 && b == 0 && k >= 0) {             // the critical workgroup: update of the next diagonal tile ordered for the chain
        CHOL_STAMP_DECL;
        CHOL_STAMP(0);
        acc4 a0;
#ifdef POTRF_DEBUG
        crit_tile_update(S, A, lda, n, k, (POTRF_CRIT_PFORM ? Xws : Wws) + (size_t)k * 4096, tid, a0, st_);
        st_[6] = st_[5];
#else
        crit_tile_update(S, A, lda, n, k, (POTRF_CRIT_PFORM ? Xws : Wws) + (size_t)k * 4096, tid, a0);
#endif
        CHOL_STAMP(2);
        TR(1);
        const int kk = k + 1, r0 = kk * 64, nr = (n - r0 < 64) ? (n - r0) : 64;
        factor64_lds(S[1], S[0], Xd, colbuf, rowbuf, tid, info, r0, nr, A + (int64_t)r0 * lda + r0, lda, Xws + (size_t)kk * 4096,
                     Wws + (size_t)kk * 4096, true, a0);
        CHOL_STAMP(3);
        CHOL_STAMP_FLUSH;
        TR(2);
        TR_VAL(22, 0);
        TR_FLUSH;
        return;
    }
#endif
    // tile row ti holds ti + 1 update tiles, dealt to workgroups in strips of strip block columns (block 0: the diagonal tile
    // of row 0, alone: the critical workgroup)
    int ti = 0, first = 0;
    const int ntr = (n + 63) / 64 - (k + 1);        // tile rows of the trailing matrix; the last one may be ragged: strips of strip_e
    while (ti < ntr - 1 && first + (ti + strip) / strip <= b) { first += (ti + strip) / strip; ++ti; }
    if (ti == ntr - 1) strip = strip_e;
    int tj = (b - first) * strip;
    const int ncols = min(strip, ti + 1 - tj);
    const int i0 = (k + 1 + ti) * 64;
    int j0 = (k + 1 + tj) * 64;
    double (*F)[LDT] = S[0];
    if constexpr (PIPE && NW == 4) {
        if (k >= 0 && b != 0 && i0 + 64 <= n) {        // interior tile row (its strip's columns lie left of it: inside too)
            const int k0 = k * 64;
            const double* Wk = Wws + (size_t)k * 4096;
            const int ml0 = wr * 32 + (lane >> 4), nl0 = wc * 32 + (lane & 15);
            const int64_t astep = (int64_t)4 * lda;
            double cs[3][2][2][4];
            double* const Cg = A + (int64_t)(i0 + ml0) * lda + j0 + nl0;
            const double* const Bg = A + (int64_t)j0 * lda + k0;
            {   // the operands of T first; C_0 and B_0 (LDS-DMA, issued LAST: the compiler's count for A_ik / W_k then leaves them
                // in flight) are only waited for behind the T product (strip_pipe)
                double ra[16], rw[16];
                const double* pa = A + (int64_t)(i0 + (tid >> 6)) * lda + k0 + (tid & 63);
#pragma unroll
                for (int u = 0; u < 16; ++u) { ra[u] = pa[u * astep]; rw[u] = Wk[((tid >> 6) + 4 * u) * 64 + (tid & 63)]; }
#if !POTRF_PIPE_LATE0
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) cs[0][t >> 2][j][t & 3] = Cg[t * astep + j * 16];
                pipe_dma_tile(S[2], Bg, lda);              // B_0 = A_jk of the strip's first column
#endif
#pragma unroll
                for (int u = 0; u < 16; ++u) { S[0][(tid >> 6) + 4 * u][tid & 63] = ra[u]; S[1][(tid >> 6) + 4 * u][tid & 63] = rw[u]; }
            }
            __syncthreads();
#if POTRF_PIPE_LATE0
            pipe_dma_tile(S[2], Bg, lda);                  // B_0, C_0 requested behind the barrier: in flight under the T product
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) cs[0][t >> 2][j][t & 3] = Cg[t * astep + j * 16];
#endif
            acc4 acc[2][2];
            TR(1);
            tile_product<true, 4>(S[0], S[1], lane, wr, wc, acc);            // T = A_ik W_k (W symmetric: [n][k] == [k][n])
#ifdef POTRF_TRACE
            strip_pipe<true>(S, Bg, lda, (int64_t)64 * lda, Cg, lda, ncols, false, ti == tj + ncols - 1, acc, cs, k == POTRF_DEBUG_K ? trs_ : nullptr);
            TR_VAL(20, (ncols << 16) | ti);
            TR_VAL(22, 1);
            TR_FLUSH;
#else
            strip_pipe<true>(S, Bg, lda, (int64_t)64 * lda, Cg, lda, ncols, false, ti == tj + ncols - 1, acc, cs);
#endif
            return;
        }
    }
    CHOL_STAMP_DECL;
    CHOL_STAMP(0);
    if (k >= 0) {
        const int k0 = k * 64;
        const double* Wk = Wws + (size_t)k * 4096;
        double cv[2][NJ][4];    // C tile, requested behind the operand loads (clamped addresses: no predicated load -> wait -> store chains)
        double rb[NU];         // A_jk, parked until T = A_ik W_k is in LDS
        {   // all 48 + 16 loads of a thread in flight at once (a rolled loop pays the global latency 16 times)
            double ra[NU], rw[NU];
            const int64_t alast = (int64_t)(n - 1) * lda, astep = (int64_t)NW * lda;
            const int64_t arow0 = (int64_t)(i0 + (tid >> 6)) * lda, brow0 = (int64_t)(j0 + (tid >> 6)) * lda;
#pragma unroll
            for (int u = 0; u < NU; ++u) {       // (rows past the matrix: clamped to the last row, zeroed on the way to LDS)
                const int r = (tid >> 6) + NW * u, c = tid & 63;
                ra[u] = A[((i0 + r < n) ? arow0 + u * astep : alast) + k0 + c];
                rw[u] = Wk[r * 64 + c];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {       // (behind the operands of T = A_ik W_k: the first LDS fill waits for those only)
                const int r = (tid >> 6) + NW * u, c = tid & 63;
                rb[u] = A[((j0 + r < n) ? brow0 + u * astep : alast) + k0 + c];
            }
            const int m0 = i0 + wr * 32 + (lane >> 4);
            const int64_t crow0 = (int64_t)m0 * lda;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nn = min(j0 + wc * (16 * NJ) + j * 16 + (lane & 15), n - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int dm = i * 16 + 4 * q;
#ifdef POTRF_ABL_NOC
                        cv[i][j][q] = 1e-3 * dm;
#else
                        cv[i][j][q] = A[((m0 + dm < n) ? crow0 + (int64_t)dm * lda : alast) + nn];
#endif
                    }
                }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int r = (tid >> 6) + NW * u, c = tid & 63;
                S[0][r][c] = (i0 + r < n) ? ra[u] : 0.0;
                S[1][r][c] = rw[u];
            }
        }
        __syncthreads();
        CHOL_STAMP(1);
        TR(1);
        acc4 acc[2][NJ];
#if defined(POTRF_ABL_P1) || defined(POTRF_ABL_NOP)     // timing ablation (wrong numbers): every tile but the critical one skips its first product
        if (b == 0) tile_product<true, NW>(S[0], S[1], lane, wr, wc, acc);
        else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = acc4{1e-3, 1e-3, 1e-3, 1e-3};
        }
#else
#if POTRF_IL
        if constexpr (NW == 4) tile_product_il<true, false>(S[0], S[1], lane, wr, wc, acc, [](int) {});
        else
#endif
        tile_product<true, NW>(S[0], S[1], lane, wr, wc, acc);          // T = A_ik W_k   (W symmetric: [n][k] == [k][n])
#endif
        CHOL_STAMP(4);
        TR(2);
        __syncthreads();
        CHOL_STAMP(5);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {       // -T to LDS, the accumulators restart from the C tile: the second product leaves C - T A_jk^T
                    S[0][wr * 32 + i * 16 + (lane >> 4) + 4 * q][wc * (16 * NJ) + j * 16 + (lane & 15)] = -acc[i][j][q];
                    acc[i][j][q] = cv[i][j][q];
                }
        const int ml0 = wr * 32 + (lane >> 4), nl0 = wc * (16 * NJ) + (lane & 15);
#pragma unroll 1
        for (int cc = 0; cc < ncols; ++cc) {         // the strip: -T stays in S[0]; A_jk and the C tile change per block column
            if (cc > 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][j][q] = cv[i][j][q];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int r = (tid >> 6) + NW * u, c = tid & 63;
                S[1][r][c] = (j0 + r < n) ? rb[u] : 0.0;
            }
            __syncthreads();
            CHOL_STAMP(6);
            if (cc < 5) TR(3 + 3 * cc);
            const bool diag = ti == tj;
            const int jcur = j0;
#if POTRF_IL && !defined(POTRF_ABL_NOP) && !defined(POTRF_ABL_NOC)
            if constexpr (NW == 4) {
                if (b != 0) {
                    if (cc + 1 < ncols) {            // next block column of the strip: its loads go out BETWEEN the MFMAs of this product
                        ++tj; j0 += 64;
                        // two pointers walking down 4 rows per k-step (no per-load 64-bit multiplies in the MFMA stream); in the ragged
                        // last block row the walk stops at the matrix's last row (those values are never used), and a strip's columns
                        // lie left of its rows: inside the matrix
                        const int64_t step4 = (int64_t)4 * lda;
                        const double* pr = A + (int64_t)min(j0 + (tid >> 6), n - 1) * lda + k * 64 + (tid & 63);
                        const int cc0 = min(j0 + nl0, n - 1), dc1 = min(j0 + nl0 + 16, n - 1) - cc0;     // (columns clamped like the rows)
                        const double* pc = A + (int64_t)min(i0 + ml0, n - 1) * lda + cc0;
                        const int rrow = j0 + (tid >> 6), crow = i0 + ml0;
                        auto pf = [&](int ks) {
                            rb[ks] = *pr;
                            pr += (rrow + 4 * (ks + 1) < n) ? step4 : 0;
                            const int t = ks & 7;                // rows 4 t of this wave's 32: cv[t >> 2][.][t & 3]
                            cv[t >> 2][ks >> 3][t & 3] = pc[(ks >> 3) ? dc1 : 0];
                            if (t == 7) pc = A + (int64_t)min(crow, n - 1) * lda + cc0;
                            else pc += (crow + 4 * (t + 1) < n) ? step4 : 0;
                        };
                        tile_product_il<true, true>(S[0], S[1], lane, wr, wc, acc, pf);
                    } else {
                        tile_product_il<true, true>(S[0], S[1], lane, wr, wc, acc, [](int) {});
                    }
                    goto product_done;
                }
            }
#endif
            if (cc + 1 < ncols) {                    // next block column of the strip: requested now, consumed after this product
                ++tj; j0 += 64;
                const int64_t alast = (int64_t)(n - 1) * lda, astep = (int64_t)NW * lda;
                const int64_t brow0 = (int64_t)(j0 + (tid >> 6)) * lda;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int r = (tid >> 6) + NW * u, c = tid & 63;
                    rb[u] = A[((j0 + r < n) ? brow0 + u * astep : alast) + k * 64 + c];
                }
                const int m0 = i0 + ml0;
                const int64_t crow0 = (int64_t)m0 * lda;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int nn = min(j0 + nl0 + j * 16, n - 1);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int dm = i * 16 + 4 * q;
#ifdef POTRF_ABL_NOC
                            cv[i][j][q] = 1e-3 * dm;
#else
                            cv[i][j][q] = A[((m0 + dm < n) ? crow0 + (int64_t)dm * lda : alast) + nn];
#endif
                        }
                    }
            }
#ifdef POTRF_ABL_NOP
            if (b == 0)
#endif
#if POTRF_IL
            if constexpr (NW == 4) tile_product_il<true, true>(S[0], S[1], lane, wr, wc, acc, [](int) {});
            else
#endif
            tile_product<true, NW, true>(S[0], S[1], lane, wr, wc, acc);    // C - T A_jk^T
#if POTRF_IL && !defined(POTRF_ABL_NOP) && !defined(POTRF_ABL_NOC)
product_done:
#endif
            CHOL_STAMP(7);
            CHOL_STAMP(8);
            if (cc < 5) TR(4 + 3 * cc);
            if (b == 0) break;                                          // (the critical tile: a strip of one, kept in LDS below)
            double* const cdst = A + (int64_t)(i0 + ml0) * lda + jcur + nl0;
#ifndef POTRF_ABL_NOC
            if (POTRF_IL && NW == 4 && !diag) {      // off-diagonal tile (its columns are inside the matrix): one pointer walking down 4 rows per pair of stores
                double* pd = cdst;
                const int64_t step4 = (int64_t)4 * lda;
                const int crow = i0 + ml0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (crow + 4 * t < n) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) pd[j * 16] = acc[t >> 2][j][t & 3];
                    }
                    pd += step4;
                }
            } else
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ml = ml0 + i * 16 + 4 * q, nl = nl0 + j * 16;
#ifdef POTRF_ABL_NOC
                        if (acc[i][j][q] == 123.456)
#endif
                        if (i0 + ml < n && jcur + nl < n && !(diag && nl > ml))
                            cdst[(int64_t)(i * 16 + 4 * q) * lda + j * 16] = acc[i][j][q];
                    }
            if (cc < 5) TR(5 + 3 * cc);
            if (cc + 1 < ncols) __syncthreads();     // every wave is done reading A_jk out of S[1]
        }
        if (b != 0) {
            TR_VAL(20, (ncols << 16) | ti);
            TR_VAL(22, 1);
            TR_FLUSH;
            return;
        }
        __syncthreads();                                            // F aliases the T tile: every wave is done reading it
        CHOL_STAMP(10);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ml = ml0 + i * 16 + 4 * q, nl = nl0 + j * 16;
                    const bool in = i0 + ml < n && j0 + nl < n && nl <= ml;
                    F[ml][nl] = in ? acc[i][j][q] : ((ml == nl) ? 1.0 : 0.0);   // identity padding of a ragged last block
                }
        CHOL_STAMP(9);
    } else {
        double ra[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int r = (tid >> 6) + NW * u, c = tid & 63;
            ra[u] = A[(int64_t)min(r, n - 1) * lda + min(c, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int r = (tid >> 6) + NW * u, c = tid & 63;
            F[r][c] = (r < n && c <= r) ? ra[u] : ((r == c) ? 1.0 : 0.0);
        }
        __syncthreads();        // (row-wise fill: the leading 16 x 16 block comes from several waves)
    }
    // (update path: wave 0 owns rows / columns 0..31 of F, so the block its first chain reads is its own -- no barrier here)
    CHOL_STAMP(2);
    // ---- factor the diagonal tile kk = k + 1 ----
    const int kk = k + 1, r0 = kk * 64, nr = (n - r0 < 64) ? (n - r0) : 64;
    double (*Y)[LDT] = S[1];
    factor64_lds(F, Y, Xd, colbuf, rowbuf, tid, info, r0, nr, A + (int64_t)r0 * lda + r0, lda, Xws + (size_t)kk * 4096,
                 Wws + (size_t)kk * 4096);
    CHOL_STAMP(3);
    CHOL_STAMP_FLUSH;
}

// the last block row of L^-1 (its X is produced by the last step launch): Y_kj = X_k R_kj, j <= k = nblk - 1
__device__ __forceinline__ void chol_yrow_tile(double (*S)[64][LDT], int b, const double* __restrict__ A, int64_t lda, int n,
                                               const double* __restrict__ Xws, const double* __restrict__ Wws,
                                               double* __restrict__ Rw, int64_t ldr, double* __restrict__ Yinv, int64_t ldy,
                                               double* __restrict__ YinvT) {
    const int nblk = (n + 63) / 64, k = nblk - 1;
    chol_inverse_tile(S, A, lda, n, k, (nblk - (k + 1)) * (k + 1) + b, Xws, Wws, Rw, ldr, Yinv, ldy, nblk, YinvT);
}

// all solved panels in one launch:  L_ik = A_ik X_k^T  (i > k), in place, one 64 x 64 tile per workgroup
// (the same launch carries the last block row of L^-1 in its first ny workgroups: the two are independent, and at M' = 600 a
//  launch of its own costs as much as either of them)
__global__ __launch_bounds__(256) void chol_panels_kernel(double* __restrict__ A, int64_t lda, int n,
                                                          const double* __restrict__ Xws, const double* __restrict__ Wws,
                                                          double* __restrict__ Rw, int64_t ldr, double* __restrict__ Yinv,
                                                          int64_t ldy, double* __restrict__ YinvT, int ny) {
    __shared__ double S[2][64][LDT];
    if ((int)blockIdx.x < ny) {
        chol_yrow_tile(S, blockIdx.x, A, lda, n, Xws, Wws, Rw, ldr, Yinv, ldy, YinvT);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int b = blockIdx.x - ny;
    int ti = (int)((sqrtf(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while (ti * (ti + 1) / 2 > b) --ti;
    const int k = b - ti * (ti + 1) / 2, i0 = (ti + 1) * 64, k0 = k * 64;     // strict lower: block row ti + 1, column k
    const double* Xk = Xws + (size_t)k * 4096;
    {
        double ra[16], rx[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = (tid >> 6) + 4 * u, c = tid & 63;
            ra[u] = A[(int64_t)min(i0 + r, n - 1) * lda + k0 + c];
            rx[u] = Xk[r * 64 + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            S[0][(tid >> 6) + 4 * u][tid & 63] = (i0 + (tid >> 6) + 4 * u < n) ? ra[u] : 0.0;
            S[1][(tid >> 6) + 4 * u][tid & 63] = rx[u];
        }
    }
    __syncthreads();
    acc4 acc[2][2];
    tile_product<true>(S[0], S[1], lane, wr, wc, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = i0 + wr * 32 + i * 16 + (lane >> 4) + 4 * q, c = k0 + wc * 32 + j * 16 + (lane & 15);
                if (m < n) A[(int64_t)m * lda + c] = acc[i][j][q];
            }
}

}  // namespace

size_t potrf_blocked_workspace_bytes(int n) {
    const size_t nblk = (size_t)cdiv(n, NBC);
    // X_k, W_k blocks + the working matrix R of the fused inverse (padded to whole blocks)
    return sizeof(double) * (2 * nblk * NBC * NBC + nblk * NBC * nblk * NBC);
}

// one fused launch per block column + one batched panel launch (see chol_step_kernel).  Yinv != nullptr: also L^-1
// (lower triangle, leading dimension ldy) by the fused forward elimination, and (YinvT != nullptr) its transpose with the
// same leading dimension.
int launch_potrf_blocked(hipStream_t st, double* A, int n, int64_t lda, int* info, double* ws, double* Yinv, int64_t ldy,
                         double* YinvT, bool info_zeroed, const PotrfHook* hooks, int nhooks, int pipe_from) {
    if (!info_zeroed) {
        hipError_t e = hipMemsetAsync(info, 0, sizeof(int), st);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    const int nblk = cdiv(n, NBC);
    double* Xws = ws;
    double* Wws = ws + (size_t)nblk * NBC * NBC;
    double* Rw = Yinv ? Wws + (size_t)nblk * NBC * NBC : nullptr;
    const int64_t ldr = (int64_t)nblk * NBC;
    for (int k = -1; k < nblk - 1; ++k) {
        const int nt = nblk - (k + 1);
        const int tiles = (k < 0) ? 1 : nt * (nt + 1) / 2 + (Yinv ? nt * (k + 1) + (k + 1) : 0);
        const int strip = tiles > POTRF_STRIP_T6 ? POTRF_STRIP_V6 : (tiles > POTRF_STRIP_T4 ? POTRF_STRIP_V4 : (tiles > POTRF_STRIP_T2 ? POTRF_STRIP_V2 : 1));
        int nA = 1;                                  // update tiles, in strips of `strip` block columns per tile row
        if (k >= 0) {
            nA = 0;
            for (int ti = 0; ti < nt; ++ti) nA += (ti + strip) / strip;
        }
        const int nI = (k >= 0 && Yinv) ? nt * ((k + strip) / strip) + (k + 1) : 0;     // R strips + Y tiles
#if POTRF_PIPE && POTRF_NW == 4
        if (k >= 0 && k >= pipe_from && k >= POTRF_PIPE_FROM && tiles > POTRF_PIPE_MIN_TILES && lda % 2 == 0 && ((uintptr_t)A % 16) == 0 &&
            (!Rw || ((uintptr_t)Rw % 16) == 0)) {      // (the B tiles arrive by 16-byte LDS-DMA pieces: base and row stride 16-byte aligned)
            // one workgroup per CU: the shortest strips that keep the launch within one round of the 256 CUs (the pipelined strip
            // costs ~13k cycles + 4.7k per column; a second round would cost a whole strip)
            int sp = 2, nAp = 0, nIp = 0, se = 2;
            for (;; ++sp) {
                se = (n % 64 == 0) ? sp : (sp < 2 ? sp : 2);           // the ragged last tile row keeps the two-barrier loop: short strips
                nAp = 0;
                for (int ti = 0; ti < nt - 1; ++ti) nAp += (ti + sp) / sp;
                nAp += (nt - 1 + se) / se;
                nIp = Yinv ? nt * ((k + sp) / sp) + (k + 1) : 0;
                if (nAp + nIp <= POTRF_PIPE_WGS || sp >= 12) break;
            }
            hipLaunchKernelGGL((chol_step_kernel<4, true>), dim3(nAp + nIp), dim3(256), 0, st, A, lda, n, k, Xws, Wws, info, Rw, ldr, Yinv,
                               ldy, nAp, YinvT, sp, se);
        } else
#endif
        hipLaunchKernelGGL((chol_step_kernel<POTRF_NW, false>), dim3(nA + nI), dim3(64 * POTRF_NW), 0, st, A, lda, n, k, Xws, Wws, info, Rw,
                           ldr, Yinv, ldy, nA, YinvT, strip, strip);
        DSVGP_LAUNCH_CHECK();
        for (int h = 0; h < nhooks; ++h)
            if (hooks[h].after_k == k) {
                const hipError_t e = hipEventRecord(hooks[h].ev, st);
                if (e != hipSuccess) return 1000 + (int)e;
            }
    }
    const int ny = Yinv ? nblk : 0, npan = nblk * (nblk - 1) / 2;
    if (ny + npan > 0) {
        hipLaunchKernelGGL(chol_panels_kernel, dim3(ny + npan), dim3(256), 0, st, A, lda, n, (const double*)Xws, (const double*)Wws, Rw,
                           ldr, Yinv, ldy, YinvT, ny);
        DSVGP_LAUNCH_CHECK();
    }
    return 0;
}

#ifdef POTRF_TRACE
extern "C" int dsvgp_debug_potrf_trace(unsigned long long* out, int nwg) {
    if (nwg > TR_MAXWG) nwg = TR_MAXWG;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chol_trace), sizeof(unsigned long long) * (size_t)nwg * TR_SLOTS);
}
#endif
#ifdef POTRF_DEBUG
extern "C" int dsvgp_debug_potrf_clock(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(chol_dbg), sizeof(unsigned long long) * 64);
}

#endif
