// Block-kernel assembly for RBFKernelDirectionalGrad (reference
// directionalvi/RBFKernelDirectionalGrad.py:41-119) and its backward, gfx950.
//
// Formulation.  For every point pack the (p+1) rows  [(x-c)/ell ; v_1 ; ... ; v_p]  (unit directions; c = a
// common shift of both point sets, the column mean of side 1 like gpytorch's covar_dist adjustment, which
// keeps the quadratic expansion below well conditioned -- the kernel only sees differences)
// into P[n(p+1), DP].  Then T = P1 P2^T is ALREADY laid out like the interleaved kernel matrix and
// holds every inner product the four block types need:
//     T[i0,j0] = x1~.x2~      T[i0,jb] = x1~.v2_b      T[ia,j0] = v1_a.x2~      T[ia,jb] = v1_a.v2_b
// With the per-point self terms  nrm = |x~|^2,  alpha_a = x~.v_a :
//     |r|^2 = nrm1 + nrm2 - 2 T00,  u_a = r.v1_a = alpha_a - Ta0,  w_b = r.v2_b = T0b - beta_b
//     K00 = k, K0b = w k/ell, Ka0 = -u k/ell, Kab = (Tab - u w) k/ell^2,  k = s exp(-|r|^2/2)
// (same quadratic-expansion arithmetic as the reference's covar_dist / x@v.T products, :71-102).
// T is computed per 96x96 tile on v_mfma_f32_16x16x4_f32 (K = d) from LDS-staged point packs, the
// micro-block transform runs out of LDS and every output row is written with coalesced stores
// directly in the interleaved M(p+1)-stride layout (no permutation pass, :105-107).
// The self terms use the same k-ordered fma chain as the MFMA, so r == 0 exactly on the diagonal
// of K_ZZ.  The backward recomputes T, forms Tbar per micro-block and contracts Tbar . P2 on MFMA.
#include "common.h"
#include <type_traits>

#ifndef ASM_ABLATE
#define ASM_ABLATE 0      // tools only: 1 = no global stores, 2 = stores of a constant (no MFMA / epilogue math)
#endif

// One-wave workgroups (64 threads) exchange through LDS, which serves a wave's instructions in issue order: all they need between a write
// and another lane's read is that the COMPILER keeps the order.  __syncthreads() would add "s_waitcnt vmcnt(0)" (its fence covers global
// memory too), i.e. drain the tile's global stores and the next tile's prefetch at every exchange (profiles/r06_c_*).
// ASM_WAVE_SYNC = 0 restores the workgroup barrier (tools/assemble_variants.sh).
#ifndef ASM_WAVE_SYNC
#define ASM_WAVE_SYNC 1
#endif
#if ASM_WAVE_SYNC
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define WAVE_SYNC() __syncthreads()
#endif


namespace {

constexpr int TMAX = 96;   // tile rows/cols of the interleaved matrix handled per workgroup
constexpr int LDT = 100;   // LDS row stride of the T / G tiles (100 % 32 == 4 -> acc writes conflict-free per half)


using f4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fdiv_small(int e, float inv) { return (int)(((float)e + 0.5f) * inv); }

// ---- pack -------------------------------------------------------------------------------------
// column means of x[n, d] (one block per column, fixed summation order): the common shift of both point sets
__global__ void column_mean_kernel(const float* __restrict__ x, int n, int d, float* __restrict__ out) {
    __shared__ double part[256];
    const int k = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[(int64_t)i * d + k];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = (float)(part[0] / (double)n);
}

// one packed row (value row a = 0 / direction row a >= 1 of point i); center: global or LDS, null = no shift
__device__ __forceinline__ void pack_row(const float* __restrict__ x, const float* __restrict__ v, int row, int d, int p, float ell,
                                         const float* center, float* __restrict__ P, float* __restrict__ self,
                                         float* __restrict__ vnorm, int K4, int DP) {
    const int q = p + 1;
    const int i = row / q, a = row - i * q;
    float* Pr = P + (int64_t)row * DP;
    const float* xi = x + (int64_t)i * d;
    if (a == 0) {
        float acc = 0.f;
        for (int k = 0; k < d; ++k) {
            const float xt = (xi[k] - (center ? center[k] : 0.f)) / ell;   // x.div(lengthscale), :67-68
            Pr[k] = xt;
            acc = __builtin_fmaf(xt, xt, acc);
        }
        for (int k = d; k < DP; ++k) Pr[k] = 0.f;
        Pr[K4] = 1.f;                               // indicator column (row sums in the backward)
        self[row] = acc;
    } else {
        const float* vi = v + ((int64_t)i * p + (a - 1)) * d;
        float ss = 0.f;
        for (int k = 0; k < d; ++k) ss = __builtin_fmaf(vi[k], vi[k], ss);
        const float nrm = sqrtf(ss);               // :57-58
        float acc = 0.f;
        for (int k = 0; k < d; ++k) {
            const float vh = vi[k] / nrm;
            Pr[k] = vh;
            acc = __builtin_fmaf(vh, (xi[k] - (center ? center[k] : 0.f)) / ell, acc);
        }
        for (int k = d; k < DP; ++k) Pr[k] = 0.f;
        self[row] = acc;
        vnorm[(int64_t)i * p + (a - 1)] = nrm;
    }
}

__global__ void pack_points_kernel(const float* __restrict__ x, const float* __restrict__ v, int n, int d,
                                   int p, const float* __restrict__ hyp, const float* __restrict__ center,
                                   float* __restrict__ P,
                                   float* __restrict__ self, float* __restrict__ vnorm, int K4, int DP) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n * (p + 1)) return;
    pack_row(x, v, row, d, p, hyp[0], center, P, self, vnorm, K4, DP);
}

// The one-call step's first launch (round 6): column_mean_hyp_kernel + pack_points_kernel of the inducing set + pack_points_kernel of the
// minibatch in ONE.  Every workgroup forms the centre (column means of Z: one wave per column, lanes over the rows, double sums, the
// wave's lanes added in a fixed order -- the same in every workgroup) and the constrained hyper-parameters for itself, workgroup 0
// publishes them (centre, hyp: later launches read them), and each packs 256 rows of Z (workgroups < nbz) or of x.
__global__ __launch_bounds__(256) void pack_both_kernel(const float* __restrict__ Z, const float* __restrict__ V, int M,
                                                        const float* __restrict__ X, const float* __restrict__ Dm, int B, int d, int p,
                                                        const float* rl, const float* rs, const float* rn, float* __restrict__ hyp,
                                                        float* __restrict__ center, float* __restrict__ PZ, float* __restrict__ sZ,
                                                        float* __restrict__ vZ, float* __restrict__ PX, float* __restrict__ sX,
                                                        float* __restrict__ vX, int K4, int DP, int nbz) {
    extern __shared__ float cs[];           // [d] centre, then ell
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int k = wave; k < d; k += 4) {
        double acc = 0.0;
        for (int i = lane; i < M; i += 64) acc += Z[(int64_t)i * d + k];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) cs[k] = (float)(acc / (double)M);
    }
    if (t == 0) {       // softplus constraints of gpytorch Positive / GreaterThan(1e-4) (csrc/elbo.hip: hyp_forward_kernel)
        auto sp = [](float v) { return v > 20.f ? v : log1pf(expf(v)); };
        const float ell = sp(rl[0]);
        cs[d] = ell;
        if (blockIdx.x == 0) { hyp[0] = ell; hyp[1] = sp(rs[0]); hyp[2] = sp(rn[0]) + 1e-4f; hyp[3] = 0.f; }
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int k = t; k < d; k += 256) center[k] = cs[k];
    const int q = p + 1;
    if ((int)blockIdx.x < nbz) {
        const int row = blockIdx.x * 256 + t;
        if (row < M * q) pack_row(Z, V, row, d, p, cs[d], cs, PZ, sZ, vZ, K4, DP);
    } else {
        const int row = ((int)blockIdx.x - nbz) * 256 + t;
        if (row < B * q) pack_row(X, Dm, row, d, p, cs[d], cs, PX, sX, vX, K4, DP);
    }
}

// stage `rows` packed rows (zero filled past `nvalid`/`limit`) and their self terms into LDS
__device__ __forceinline__ void stage_pack(float* Ps, float* selfs, const float* __restrict__ P,
                                           const float* __restrict__ self, int row0, int nvalid, int limit,
                                           int rows_pad, int DP, int ncol, int LDP) {
    const float inv_ncol = 1.f / (float)ncol;
    for (int e = threadIdx.x; e < rows_pad * ncol; e += blockDim.x) {
        const int r = fdiv_small(e, inv_ncol), k = e - r * ncol;
        const int gr = row0 + r;
        Ps[r * LDP + k] = (r < nvalid && gr < limit && k < DP) ? P[(int64_t)gr * DP + k] : 0.f;
    }
    for (int r = threadIdx.x; r < rows_pad; r += blockDim.x) {
        const int gr = row0 + r;
        selfs[r] = (r < nvalid && gr < limit) ? self[gr] : 0.f;
    }
}

// T tile = P1s P2s^T on MFMA; wave w takes 16x16 tiles w, w+4, ...
__device__ __forceinline__ void mfma_T(float* Ts, const float* P1s, const float* P2s, int ntr, int ntc,
                                       int K4, int LDP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int id = wave; id < ntr * ntc; id += 4) {
        const int tr = id / ntc, tc = id - tr * ntc;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* pa = P1s + (tr * 16 + (lane & 15)) * LDP + (lane >> 4);
        const float* pb = P2s + (tc * 16 + (lane & 15)) * LDP + (lane >> 4);
        for (int kk = 0; kk < K4; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk], pb[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(tr * 16 + (lane >> 4) * 4 + r) * LDT + tc * 16 + (lane & 15)] = acc[r];
    }
}

// ---- forward ------------------------------------------------------------------------------------
// Workgroup tile: (Rr points x Rc points) = (Tr x Tc) outputs with Tr <= 48, Tc <= 96: ~37 KB of LDS, so four
// workgroups (16 waves) share a CU and hide each other's staging / LDS latency.
template <typename OutT>
__global__ __launch_bounds__(256) void kernel_fwd_kernel(const float* __restrict__ P1, const float* __restrict__ self1,
                                                         int n1q, const float* __restrict__ P2,
                                                         const float* __restrict__ self2, int n2q, int q, int Rr,
                                                         int Rc, int K4, int DP, const float* __restrict__ hyp,
                                                         float jitter, OutT* __restrict__ out, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = K4 + 1;
    const int Tr = Rr * q, Tc = Rc * q;         // valid tile extent
    const int Trp = (Tr + 15) & ~15, Tcp = (Tc + 15) & ~15;
    float* P1s = smem;
    float* P2s = P1s + Trp * LDP;
    float* Ts = P2s + Tcp * LDP;
    float* s1 = Ts + Trp * LDT;
    float* s2 = s1 + Trp;
    float* KK = s2 + Tcp;                       // Rr * Rc pair values
    const int row0 = blockIdx.y * Tr, col0 = blockIdx.x * Tc;
    const int rows = min(Tr, n1q - row0), cols = min(Tc, n2q - col0);
    if (ASM_ABLATE == 2) {
        for (int e = threadIdx.x; e < Tr * Tc; e += 256) {
            const int r = e / Tc, c = e - r * Tc;
            if (r < rows && c < cols) out[(int64_t)(row0 + r) * ld + col0 + c] = (OutT)1.f;
        }
        return;
    }
    stage_pack(P1s, s1, P1, self1, row0, Tr, n1q, Trp, DP, K4, LDP);
    stage_pack(P2s, s2, P2, self2, col0, Tc, n2q, Tcp, DP, K4, LDP);
    __syncthreads();
    if (ASM_ABLATE == 3) { if (P1s[threadIdx.x] == 123.456f) out[0] = (OutT)1.f; return; }
    mfma_T(Ts, P1s, P2s, Trp / 16, Tcp / 16, K4, LDP);
    __syncthreads();
    if (ASM_ABLATE == 4) { if (Ts[threadIdx.x] == 123.456f) out[0] = (OutT)1.f; return; }

    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    const float invq = 1.f / (float)q, invRc = 1.f / (float)Rc;
    // pass A: one exp per PAIR of points (not per output): KK[ri][rj] = s exp(-|r|^2 / 2)
    if (q > 1) {
        for (int pid = threadIdx.x; pid < Rr * Rc; pid += 256) {
            const int pi = fdiv_small(pid, invRc), pj = pid - pi * Rc;
            const float nn = fmaxf(s1[pi * q] + s2[pj * q] - 2.f * Ts[pi * q * LDT + pj * q], 0.f);   // covar_dist clamps at 0
            KK[pid] = s * expf(-0.5f * nn);                                                          // postprocess_rbf, ScaleKernel
        }
        __syncthreads();
    }
    // pass B: thread <-> fixed column c (consecutive lanes -> consecutive columns: coalesced stores), rows strided
    // by the number of row groups; everything that depends only on the column (point j, direction b, s2, the
    // addresses of T[.,c0] / T[.,c], the global column) is loop invariant, the row decomposition (point i,
    // direction a) advances by counters: ~20 instructions and 5 LDS reads per output.
    const int ngrp = 256 / Tc;                       // row groups that fit the workgroup (>= 2 for Tc <= 96)
    const int c = threadIdx.x % Tc, rg = threadIdx.x / Tc;
    if (rg < ngrp && c < cols) {
        const int rj = fdiv_small(c, invq);
        const int c0 = rj * q, b = c - c0;
        const float s2c = s2[c];
        OutT* optr = out + (int64_t)(row0 + rg) * ld + col0 + c;
        const int64_t ostep = (int64_t)ngrp * ld;
        const int64_t gc = col0 + c;
        if (q > 1) {
            int ri = fdiv_small(rg, invq);
            int a = rg - ri * q;
            const int da = ngrp % q, di = ngrp / q;
#pragma unroll 4
            for (int r = rg; r < rows; r += ngrp) {
                const int r0 = r - a;
                const float k = KK[ri * Rc + rj];
                const float t = Ts[r * LDT + c];
                const float u = s1[r] - Ts[r * LDT + c0];           // r.v1_a   (a > 0)
                const float w = Ts[r0 * LDT + c] - s2c;             // r.v2_b   (b > 0)
                const float f0 = b ? (w * il) : 1.f;
                const float f1 = b ? ((t - u * w) * il2) : (-u * il);
                float val = (a ? f1 : f0) * k;
                if (row0 + r == gc) val += jitter;
                if (ASM_ABLATE == 1) { if (val == 123.456f) *optr = (OutT)val; }
                else *optr = (OutT)val;
                optr += ostep;
                a += da; ri += di;
                if (a >= q) { a -= q; ++ri; }
            }
        } else {
            for (int r = rg; r < rows; r += ngrp) {
                float val = s * expf(-0.5f * fmaxf(s1[r] + s2c - 2.f * Ts[r * LDT + c], 0.f));
                if (row0 + r == gc) val += jitter;
                *optr = (OutT)val;
                optr += ostep;
            }
        }
    }
}

// ---- forward, register-resident variant for small compile-time q ---------------------------------------------
// Same organisation as kernel_bwd_pair_kernel below: ONE WAVE per workgroup (no barriers), 48 x 48 tiles, the wave owns
// R = 48/Q points of side 1 and sweeps column tiles.  T' = P1' P2'^T (self terms folded in as two extra packed columns:
// T'[r0, cb] = w_b, T'[ra, c0] = -u_a) goes MFMA -> LDS -> registers, every lane turns the Q x Q values of its point
// pair(s) into the kernel micro-block with one exp and writes it straight to HBM in the interleaved layout.
// ~15 KB of LDS per wave: 8 waves per CU overlap each other's staging, MFMA and store phases.
constexpr int FWD_PAIR_LDT = 52;
#ifndef FWDP_ABL
#define FWDP_ABL 0        // tools only (results wrong): 1 = no global stores, 2 = no T' MFMA product, 4 = no P2 staging loads
#endif
#ifndef FWDP_ST16
#define FWDP_ST16 1       // q = 6, float output: 16-byte stores shared by lane pairs (two store instructions per row instead of three)
#endif
#ifndef FWDP_AREG
#define FWDP_AREG 1       // A fragments of the wave's 48 rows in registers for the whole column sweep (LDS: 10 KB per wave, 16 waves per CU)
#endif
#ifndef FWDP_OVERLAY
#define FWDP_OVERLAY 1    // T' overlays the P2 image (10 instead of 8 waves per CU) and the next tile's P2 rows are prefetched
#endif
#ifndef FWD_PAIR_WGS_
#define FWD_PAIR_WGS_ (256 * 16)      // probed 4 / 6 / 8 / 12 / 16 / 32 per CU: 207 / 170 / 137 / 133 / 121 / 135 us for K_ZX at C4
#endif

#ifndef FWDP_MINW
#define FWDP_MINW 1
#endif
#ifndef FWDP_LDS_EXTRA
#define FWDP_LDS_EXTRA 0  // tools only: unused LDS bytes per wave (30720 -> 4 waves per CU = 1 per SIMD: the occupancy ablation)
#endif
#ifndef FWDP_CHUNK
#define FWDP_CHUNK 1      // consecutive column tiles per wave (87.7 -> 85.7 us)
#endif
#ifndef FWDP_LEAN
#define FWDP_LEAN 1       // tiles inside the matrix are fetched and staged without predicates (needs FWDP_OVERLAY and a 49th LDS row)
#endif
#ifndef FWDP_FAST
#define FWDP_FAST 1       // interior tiles leave through the T' tile as fully coalesced 16-byte stores
#endif
// KSM: k-steps of the T' product the A fragments are held for: 8 (packed width DP <= 32: d <= 28, the BASELINE configs 2 and 4) or 16
// (DP <= 64: d <= 60, BASELINE config 5 at d = 50; round 5 -- more registers: one wave per SIMD instead of two)
template <typename OutT, int Q, int KSM = 8>
__global__ __launch_bounds__(64, FWDP_MINW) void kernel_fwd_pair_kernel(const float* __restrict__ P1, const float* __restrict__ self1,
                                                             int n1q, const float* __restrict__ P2,
                                                             const float* __restrict__ self2, int n2q, int K4, int DP,
                                                             int ovec, const float* __restrict__ hyp, float jitter,
                                                             OutT* __restrict__ out, int64_t ld) {
    constexpr int R = 48 / Q, T = R * Q;
    constexpr int NPAIR = R * R, PPL = (NPAIR + 63) / 64;
    constexpr int LDT2 = FWD_PAIR_LDT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = K4 + 5;
    float* P1s = smem;                  // [48][LDP]   (FWDP_AREG: not used, the A fragments of the wave's 48 rows stay in registers)
    float* P2s = FWDP_AREG ? smem : P1s + 48 * LDP;        // [48][LDP]
    float* TT = FWDP_OVERLAY ? P2s : P2s + 48 * LDP;         // [48][LDT2] (overlay: T' is written once every P2 fragment is in registers)
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * T;
    const int ncoltiles = (n2q + T - 1) / T;
    const int KS = K4 / 4 + 1;
    const int pch = DP / 4;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;

    float areg[3][KSM];                 // A fragments: areg[i][ks] = P1'[row0 + 16 i + m16][4 ks + kg]  (KS <= KSM since DP <= 4 KSM)
    if (FWDP_AREG) {
        for (int e = lane; e < 48 * LDP; e += 64) P2s[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int gr = row0 + i * 16 + m16;
            const bool ok = gr < n1q;
            const int a = (i * 16 + m16) % Q;
            const float sf = ok ? self1[gr] : 0.f;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                float v = 0.f;
                if (ks < K4 / 4) v = ok ? P1[(int64_t)gr * DP + ks * 4 + kg] : 0.f;
                else if (ks == K4 / 4 && ok) v = (kg == 1) ? (a == 0 ? 1.f : 0.f) : ((kg == 2) ? (a == 0 ? 0.f : -sf) : 0.f);
                areg[i][ks] = v;
            }
        }
    } else {
    for (int e = lane; e < 48 * LDP; e += 64) { P1s[e] = 0.f; P2s[e] = 0.f; }
    WAVE_SYNC();
    for (int e = lane; e < T * K4; e += 64) {
        const int r = e / K4, k = e - r * K4;
        if (row0 + r < n1q) P1s[r * LDP + k] = P1[(int64_t)(row0 + r) * DP + k];
    }
    if (lane < T && row0 + lane < n1q) {
        const int a = lane % Q;
        P1s[lane * LDP + K4 + 1] = a == 0 ? 1.f : 0.f;
        P1s[lane * LDP + K4 + 2] = a == 0 ? 0.f : -self1[row0 + lane];
    }
    }
    int pr0[PPL], pc0[PPL];
    float s1r0[PPL];
    bool prow[PPL];
#pragma unroll
    for (int pp = 0; pp < PPL; ++pp) {
        const int pid = lane + 64 * pp;
        const int pi = pid / R, pj = pid - pi * R;
        pr0[pp] = pi * Q; pc0[pp] = pj * Q;
        prow[pp] = pid < NPAIR && row0 + pr0[pp] < n1q;
        s1r0[pp] = prow[pp] ? self1[row0 + pr0[pp]] : 0.f;
    }

    constexpr int NPFP = (48 * KSM + 63) / 64;     // float4 per lane of one prefetched P2 tile (DP <= 4 KSM)
    f4 pf[NPFP];
    float pselfv = 0.f;
    // The packed rows of a column tile are ONE contiguous piece of P2 (T rows of DP floats): float4 number e = lane + 64 u of it
    // goes to row e / pch of the LDS image.  Per-lane constants: the element index (clamped to the tile for the lanes past its
    // end) and the LDS offset (those lanes write to a scratch row behind the image) -- a tile that lies inside the matrix is then
    // fetched and staged without a single predicate (the guarded form below spends ~15 instructions per float4 on masks, zero
    // fills and 64-bit address arithmetic: PMC, profiles/r03_c_pmc_assemble_fwd_*.txt).
    int pf_e[NPFP], pf_lds[NPFP];
#pragma unroll
    for (int u = 0; u < NPFP; ++u) {
        const int e = lane + 64 * u;
        const int ec = min(e, T * pch - 1);
        const int r = ec / pch, k = (ec - r * pch) * 4;
        pf_e[u] = ec;
        pf_lds[u] = (e < T * pch ? r : 48) * LDP + k;
    }
    const bool rows_full = row0 + T <= n1q;
    auto prefetch = [&](int ct_) {
        const int c0_ = ct_ * T;
#if FWDP_LEAN
        if (c0_ + T <= n2q) {
            const f4* src = reinterpret_cast<const f4*>(P2 + (int64_t)c0_ * DP);
#pragma unroll
            for (int u = 0; u < NPFP; ++u) pf[u] = src[pf_e[u]];
            pselfv = -self2[c0_ + min(lane, T - 1)];
            return;
        }
#endif
#pragma unroll
        for (int u = 0; u < NPFP; ++u) {
            const int e = lane + 64 * u;
            const int r = e / pch, k = (e - r * pch) * 4;
            pf[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (e < T * pch && c0_ + r < n2q) pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)(c0_ + r) * DP + k);
        }
        pselfv = (lane < T && c0_ + lane < n2q) ? -self2[c0_ + lane] : 0.f;
    };
    // ovec bit 3 (the one-call steps' K_ZZ: ctx->fwd_lower_only): only the column tiles that reach into the 64 x 64 blocks on or below the block
    // diagonal of this row tile's rows -- what the blocked Cholesky factorisation reads
    const int ct_end = (ovec & 8) ? min(ncoltiles, (64 * ((row0 + T - 1) / 64) + 63) / T + 1) : ncoltiles;
#if FWDP_CHUNK
    // consecutive column tiles per wave: the 192-byte row pieces of neighbouring tiles complete each other's 128-byte lines in ONE L2
    const int cper = (ct_end + (int)gridDim.x - 1) / (int)gridDim.x;
    const int ct_lo = blockIdx.x * cper, ct_hi = min(ct_lo + cper, ct_end), ct_step = 1;
#else
    const int ct_lo = blockIdx.x, ct_hi = ct_end, ct_step = gridDim.x;
#endif
    if (FWDP_OVERLAY && ct_lo < ct_hi) prefetch(ct_lo);
    for (int ct = ct_lo; ct < ct_hi; ct += ct_step) {
        const int col0 = ct * T;
        float s2c0[PPL];
        const bool full = FWDP_LEAN && rows_full && col0 + T <= n2q;       // (wave-uniform)
        if (full) {
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) s2c0[pp] = self2[col0 + min(pc0[pp], T - 1)];
        } else {
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) s2c0[pp] = (prow[pp] && col0 + pc0[pp] < n2q) ? self2[col0 + pc0[pp]] : 0.f;
        }
        WAVE_SYNC();   // single wave: orders the previous tile's LDS reads before the new stores
        if (FWDP_OVERLAY) {
#if FWDP_LEAN
#pragma unroll
            for (int u = 0; u < NPFP; ++u) {           // (lanes past the end of the tile: the scratch row behind the image)
#pragma unroll
                for (int t = 0; t < 4; ++t) P2s[pf_lds[u] + t] = pf[u][t];
            }
#else
#pragma unroll
            for (int u = 0; u < NPFP; ++u) {
                const int e = lane + 64 * u;
                if (e < T * pch) {
                    const int r = e / pch, k = (e - r * pch) * 4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) P2s[r * LDP + k + t] = pf[u][t];
                }
            }
#endif
            WAVE_SYNC();
            if (lane < T) {
                P2s[lane * LDP + K4 + 1] = pselfv;
                P2s[lane * LDP + K4 + 2] = (col0 + lane < n2q && lane % Q == 0) ? 1.f : 0.f;
            }
            if (ct + ct_step < ct_hi) prefetch(ct + ct_step);      // in flight under the rest of this tile
        } else {
        if (!(FWDP_ABL & 4))
        for (int e = lane; e < T * pch; e += 64) {
            const int r = e / pch, k = (e - r * pch) * 4;
            const int gr = col0 + r;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < n2q) v = *reinterpret_cast<const f4*>(P2 + (int64_t)gr * DP + k);
#pragma unroll
            for (int t = 0; t < 4; ++t) P2s[r * LDP + k + t] = v[t];
        }
        WAVE_SYNC();
        if (lane < T) {
            const int gr = col0 + lane;
            const bool ok = gr < n2q;
            P2s[lane * LDP + K4 + 1] = ok ? -self2[gr] : 0.f;
            P2s[lane * LDP + K4 + 2] = (ok && lane % Q == 0) ? 1.f : 0.f;
        }
        }
        WAVE_SYNC();
        {
            f4 t[3][3];
            if (!FWDP_AREG) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[i][j] = f4{0.f, 0.f, 0.f, 0.f};
            }
            const float* pa = P1s + m16 * LDP + kg;
            const float* pb = P2s + m16 * LDP + kg;
            if (FWDP_AREG) {
                // one straight-line copy per K depth: no per-step branches, the first product starts from the inline zero
                auto product = [&](auto ksc) {
                    constexpr int KS_ = decltype(ksc)::value;
#pragma unroll
                    for (int ks = 0; ks < KS_; ++ks) {
                        float bv[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) bv[j] = pb[j * 16 * LDP + ks * 4];
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][ks], bv[j], ks == 0 ? f4{0.f, 0.f, 0.f, 0.f} : t[i][j], 0, 0, 0);
                    }
                };
                switch ((FWDP_ABL & 2) ? 1 : KS) {
                    case 1: product(std::integral_constant<int, 1>{}); break;
                    case 2: product(std::integral_constant<int, 2>{}); break;
                    case 3: product(std::integral_constant<int, 3>{}); break;
                    case 4: product(std::integral_constant<int, 4>{}); break;
                    case 5: product(std::integral_constant<int, 5>{}); break;
                    case 6: product(std::integral_constant<int, 6>{}); break;
                    case 7: product(std::integral_constant<int, 7>{}); break;
                    case 8: product(std::integral_constant<int, 8>{}); break;
                    default:
                        if constexpr (KSM > 8) {
                            switch (KS) {
                                case 9: product(std::integral_constant<int, 9>{}); break;
                                case 10: product(std::integral_constant<int, 10>{}); break;
                                case 11: product(std::integral_constant<int, 11>{}); break;
                                case 12: product(std::integral_constant<int, 12>{}); break;
                                case 13: product(std::integral_constant<int, 13>{}); break;
                                case 14: product(std::integral_constant<int, 14>{}); break;
                                case 15: product(std::integral_constant<int, 15>{}); break;
                                default: product(std::integral_constant<int, 16>{}); break;
                            }
                        } else product(std::integral_constant<int, 8>{});
                        break;
                }
            } else
            for (int ks = 0; ks < ((FWDP_ABL & 2) ? 1 : KS); ++ks) {
                float av[3], bv[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { av[i] = pa[i * 16 * LDP + ks * 4]; bv[i] = pb[i * 16 * LDP + ks * 4]; }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], t[i][j], 0, 0, 0);
            }
            if (FWDP_OVERLAY) WAVE_SYNC();          // every P2 fragment has been read: T' may overlay the image
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) TT[(i * 16 + kg * 4 + r) * LDT2 + j * 16 + m16] = t[i][j][r];
        }
        WAVE_SYNC();
#if FWDP_FAST
        // Interior tile, float output, 16-byte aligned rows: the micro-blocks go back to the T' tile IN PLACE (every lane rewrites
        // only what it read) and the tile leaves as nine fully coalesced store instructions -- every row 192 contiguous bytes,
        // 12 lanes x 16 B.  The per-lane micro-block stores of the general path below write 16-byte pieces 16 bytes apart: each
        // instruction touches 16 cache lines partially and the next one touches them again; the CU's store path, not HBM, was the
        // bound (stores redirected to an L2-resident region: 86 -> 81 us; no stores: 60 us).
        if constexpr (sizeof(OutT) == 4 && T == 48) {
            if ((ovec & 2) && row0 + T <= n1q && col0 + T <= n2q && !(jitter != 0.f && row0 == col0)) {
#pragma unroll
                for (int pp = 0; pp < PPL; ++pp) {
                    if (lane + 64 * pp < NPAIR) {
                        float* blk = TT + pr0[pp] * LDT2 + pc0[pp];
                        float tq[Q][Q];
#pragma unroll
                        for (int a = 0; a < Q; ++a) {
                            if constexpr (Q % 2 == 0) {
                                using F2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
                                for (int b = 0; b < Q; b += 2) {
                                    const F2 v = *reinterpret_cast<const F2*>(blk + a * LDT2 + b);
                                    tq[a][b] = v[0]; tq[a][b + 1] = v[1];
                                }
                            } else {
#pragma unroll
                                for (int b = 0; b < Q; ++b) tq[a][b] = blk[a * LDT2 + b];
                            }
                        }
                        const float nn = fmaxf(s1r0[pp] - s2c0[pp] - 2.f * tq[0][0], 0.f);
                        const float k = s * expf(-0.5f * nn);
                        const float kil = k * il, kil2 = k * il2;
#pragma unroll
                        for (int a = 0; a < Q; ++a) {
                            float v[Q];
                            if (a == 0) {
                                v[0] = k;
#pragma unroll
                                for (int b = 1; b < Q; ++b) v[b] = tq[0][b] * kil;
                            } else {
                                v[0] = tq[a][0] * kil;
#pragma unroll
                                for (int b = 1; b < Q; ++b) v[b] = (tq[a][b] + tq[a][0] * tq[0][b]) * kil2;
                            }
                            if constexpr (Q % 2 == 0) {
                                using F2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
                                for (int b = 0; b < Q; b += 2) *reinterpret_cast<F2*>(blk + a * LDT2 + b) = F2{v[b], v[b + 1]};
                            } else {
#pragma unroll
                                for (int b = 0; b < Q; ++b) blk[a * LDT2 + b] = v[b];
                            }
                        }
                    }
                }
                WAVE_SYNC();
                float* orow = (float*)out + (int64_t)row0 * ld + col0;
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int id = lane + 64 * u, r = id / 12, c4 = (id - 12 * r) * 4;
                    const f4 x = *reinterpret_cast<const f4*>(TT + r * LDT2 + c4);
                    if (!(FWDP_ABL & 1)) *reinterpret_cast<f4*>(orow + (int64_t)r * ld + c4) = x;
                }
                continue;
            }
        }
#endif
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            const bool mine = prow[pp] && col0 + pc0[pp] < n2q;
            // (computed by every lane, stored by the valid ones: the 16-byte store path below exchanges values between
            //  neighbouring lanes, which must not sit inside divergent control flow)
            const float* blk = TT + pr0[pp] * LDT2 + pc0[pp];
            float tq[Q][Q];
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                if constexpr (Q % 2 == 0) {
                    using F2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int b = 0; b < Q; b += 2) {
                        const F2 v = *reinterpret_cast<const F2*>(blk + a * LDT2 + b);
                        tq[a][b] = v[0]; tq[a][b + 1] = v[1];
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < Q; ++b) tq[a][b] = blk[a * LDT2 + b];
                }
            }
            const float nn = fmaxf(s1r0[pp] - s2c0[pp] - 2.f * tq[0][0], 0.f);     // covar_dist clamps at 0
            const float k = s * expf(-0.5f * nn);                                  // postprocess_rbf, ScaleKernel
            const float kil = k * il, kil2 = k * il2;
            const int64_t gr0 = (int64_t)row0 + pr0[pp], gc0 = (int64_t)col0 + pc0[pp];
            OutT* o = out + ((FWDP_ABL & 8) ? (gr0 & 63) : gr0) * ld + gc0;      // (ablation 8: every store lands in the first 64 rows -- L2 hits)
            // Q = 6, float: lanes 2j / 2j + 1 hold horizontally adjacent micro-blocks = 12 consecutive floats per row, 48-byte
            // aligned.  The even lane stores floats 0..3 and 4..7 (its last two + the neighbour's first two, fetched with a
            // DPP quad permute), the odd lane floats 8..11: two 16-byte store instructions per row instead of three 8-byte ones
            // (the kernel is store-issue bound)
            bool both = false;
            if constexpr (Q == 6 && sizeof(OutT) == 4) {
                const int nb = __builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
                both = mine && nb && (ovec & 2);
            }
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                float v[Q];
                if (a == 0) {
                    v[0] = k;
#pragma unroll
                    for (int b = 1; b < Q; ++b) v[b] = tq[0][b] * kil;                          // w_b k / ell
                } else {
                    v[0] = tq[a][0] * kil;                                                      // -u_a k / ell
#pragma unroll
                    for (int b = 1; b < Q; ++b) v[b] = (tq[a][b] + tq[a][0] * tq[0][b]) * kil2;  // (G_ab - u_a w_b) k / ell^2
                }
                if (jitter != 0.f && gr0 == gc0) v[a] += jitter;      // diagonal micro-block: global row == global column
                if ((FWDP_ABL & 1) && v[0] != 123.456f) continue;
                if constexpr (Q == 6 && sizeof(OutT) == 4) {
                    const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[0]), 0xB1, 0xF, 0xF, false));
                    const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[1]), 0xB1, 0xF, 0xF, false));
                    if (both) {
                        const bool odd = lane & 1;
                        const f4 x = odd ? f4{v[2], v[3], v[4], v[5]} : f4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f4*>((float*)o + a * ld + (odd ? 2 : 0)) = x;
                        if (!odd) *reinterpret_cast<f4*>((float*)o + a * ld + 4) = f4{v[4], v[5], n0, n1};
                        continue;
                    }
                }
                if (!mine) continue;
                if constexpr (Q % 2 == 0) {
                    if (ovec & 1) {
                        using O2 = OutT __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int b = 0; b < Q; b += 2) *reinterpret_cast<O2*>(o + a * ld + b) = O2{(OutT)v[b], (OutT)v[b + 1]};
                        continue;
                    }
                }
#pragma unroll
                for (int b = 0; b < Q; ++b) o[a * ld + b] = (OutT)v[b];
            }
        }
    }
}

// ---- forward, "split" variant for wide micro-blocks (round 5) ------------------------------------------------------------
// q = 11 (full-gradient SVGP at d = 10: BASELINE config 3, GradVariationalStrategy.py:89-99) leaves the pair kernel's mapping
// (one lane per point pair, the Q x Q micro-block in registers) with R^2 = 16 pairs per 44 x 44 tile and 121 values per lane.
// Here SPL = 64 / R^2 lanes share a pair: lane (pair, sub) takes the micro-block rows a = sub, sub + SPL, ..., reads row 0 of the
// T' micro-block (w_b; every lane of the pair needs it) and its own rows from LDS, and writes its rows back IN PLACE; the tile
// then leaves as fully coalesced 16-byte stores (T = 44: every row 176 contiguous bytes) like the pair kernel's interior path.
// One wave per workgroup, the A fragments of its T side-1 rows in registers for the whole column sweep, T' overlays the packed
// side-2 rows it was computed from.  Same arithmetic as kernel_fwd_pair_kernel (self terms folded into two extra packed columns).
template <typename OutT, int Q, int KSM>
__global__ __launch_bounds__(64) void kernel_fwd_split_kernel(const float* __restrict__ P1, const float* __restrict__ self1, int n1q,
                                                              const float* __restrict__ P2, const float* __restrict__ self2, int n2q,
                                                              int K4, int DP, int ovec, const float* __restrict__ hyp, float jitter,
                                                              OutT* __restrict__ out, int64_t ld) {
    constexpr int R = 48 / Q, T = R * Q, NPAIR = R * R, SPL = 64 / NPAIR, RPL = (Q + SPL - 1) / SPL;
    constexpr int LDT2 = FWD_PAIR_LDT;
    static_assert(NPAIR * SPL <= 64 && SPL >= 2 && T % 4 == 0, "split mapping: R^2 pairs x SPL lanes, 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = K4 + 5;
    float* P2s = smem;                  // [49][LDP]  packed side-2 rows of the tile (+ a scratch row), overlaid by ...
    float* TT = smem;                   // [48][LDT2] ... the T' tile once every B fragment is in registers
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * T;
    const int ncoltiles = (n2q + T - 1) / T;
    const int KS = K4 / 4 + 1, pch = DP / 4;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;

    float areg[3][KSM];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int lr = i * 16 + m16, gr = row0 + lr;
        const bool ok = lr < T && gr < n1q;
        const int a = lr % Q;
        const float sf = ok ? self1[gr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            float v = 0.f;
            if (ks < K4 / 4) v = ok ? P1[(int64_t)gr * DP + ks * 4 + kg] : 0.f;
            else if (ks == K4 / 4 && ok) v = (kg == 1) ? (a == 0 ? 1.f : 0.f) : ((kg == 2) ? (a == 0 ? 0.f : -sf) : 0.f);
            areg[i][ks] = v;
        }
    }
    for (int e = lane; e < 49 * LDP; e += 64) P2s[e] = 0.f;
    // this lane's share of the tile: pair (pi, pj), rows a = sub + SPL i of its micro-block
    const int pid = lane / SPL, sub = lane - pid * SPL;
    const int pi = pid / R, pj = pid - pi * R;
    const int pr0 = pi * Q, pc0 = pj * Q;
    const bool pvalid = pid < NPAIR;
    const bool prow = pvalid && row0 + pr0 < n1q;
    const float s1r0 = prow ? self1[row0 + pr0] : 0.f;
    const bool rows_full = row0 + T <= n1q;
    // consecutive column tiles per wave (as in the pair kernel: neighbouring tiles complete each other's cache lines)
    const int cper = (ncoltiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int ct_lo = blockIdx.x * cper, ct_hi = min(ct_lo + cper, ncoltiles);
    // the packed rows of a column tile are one contiguous piece of P2 (T rows of DP floats); the NEXT tile's travel through
    // registers under the current tile's transform and stores
    constexpr int NPF = (T * KSM + 63) / 64;       // float4 per lane (pch <= KSM)
    f4 pf[NPF];
    float pselfv = 0.f;
    auto prefetch = [&](int ct_) {
        const int c0_ = ct_ * T;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = lane + 64 * u, r = e / pch, k = (e - r * pch) * 4;
            pf[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (e < T * pch && c0_ + r < n2q) pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)(c0_ + r) * DP + k);
        }
        pselfv = (lane < T && c0_ + lane < n2q) ? -self2[c0_ + lane] : 0.f;
    };
    if (ct_lo < ct_hi) prefetch(ct_lo);
    for (int ct = ct_lo; ct < ct_hi; ++ct) {
        const int col0 = ct * T;
        const bool colok = prow && col0 + pc0 < n2q;
        const float s2c0 = colok ? self2[col0 + pc0] : 0.f;
        WAVE_SYNC();                 // (single wave: the previous tile's LDS reads are done)
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = lane + 64 * u, r = e / pch, k = (e - r * pch) * 4;
            if (e < T * pch) {
#pragma unroll
                for (int t = 0; t < 4; ++t) P2s[r * LDP + k + t] = pf[u][t];
            }
        }
        WAVE_SYNC();
        if (lane < T) {
            P2s[lane * LDP + K4 + 1] = pselfv;
            P2s[lane * LDP + K4 + 2] = (col0 + lane < n2q && lane % Q == 0) ? 1.f : 0.f;
        }
        if (ct + 1 < ct_hi) prefetch(ct + 1);
        WAVE_SYNC();
        {
            f4 t[3][3];
            const float* pb = P2s + m16 * LDP + kg;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                if (ks < KS) {
                    float bv[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) bv[j] = pb[j * 16 * LDP + ks * 4];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][ks], bv[j], ks == 0 ? f4{0.f, 0.f, 0.f, 0.f} : t[i][j], 0, 0, 0);
                }
            }
            WAVE_SYNC();             // every B fragment has been read: T' may overlay the packed rows
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) TT[(i * 16 + kg * 4 + r) * LDT2 + j * 16 + m16] = t[i][j][r];
        }
        WAVE_SYNC();
        // ---- micro-block transform, rows split over the SPL lanes of a pair
        float* blk = TT + pr0 * LDT2 + pc0;
        float t0[Q], ta[RPL][Q];
        if (pvalid) {
#pragma unroll
            for (int b = 0; b < Q; ++b) t0[b] = blk[b];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int a = sub + SPL * i;
#pragma unroll
                for (int b = 0; b < Q; ++b) ta[i][b] = (a < Q) ? blk[a * LDT2 + b] : 0.f;
            }
        }
        WAVE_SYNC();                 // row 0 is rewritten by the lane with sub = 0: every lane of the pair has read it
        const float nn = fmaxf(s1r0 - s2c0 - 2.f * t0[0], 0.f);      // covar_dist clamps at 0
        const float k = s * expf(-0.5f * nn);                          // postprocess_rbf, ScaleKernel
        const float kil = k * il, kil2 = k * il2;
        const int64_t gr0 = (int64_t)row0 + pr0, gc0 = (int64_t)col0 + pc0;
        const bool fast = sizeof(OutT) == 4 && (ovec & 2) && rows_full && col0 + T <= n2q;     // (wave-uniform)
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int a = sub + SPL * i;
            float v[Q];
            if (a == 0) {
                v[0] = k;
#pragma unroll
                for (int b = 1; b < Q; ++b) v[b] = t0[b] * kil;                                  // w_b k / ell
            } else {
                v[0] = ta[i][0] * kil;                                                           // -u_a k / ell
#pragma unroll
                for (int b = 1; b < Q; ++b) v[b] = (ta[i][b] + ta[i][0] * t0[b]) * kil2;         // (G_ab - u_a w_b) k / ell^2
            }
            if (jitter != 0.f && gr0 == gc0) {
#pragma unroll
                for (int b = 0; b < Q; ++b) if (b == a) v[b] += jitter;                          // diagonal of the diagonal micro-block
            }
            if (!pvalid || a >= Q) continue;
            if (fast) {
#pragma unroll
                for (int b = 0; b < Q; ++b) blk[a * LDT2 + b] = v[b];
            } else if (colok) {          // edge tiles / double output: the lane stores its own rows
                OutT* o = out + (gr0 + a) * ld + gc0;
#pragma unroll
                for (int b = 0; b < Q; ++b) o[b] = (OutT)v[b];
            }
        }
        if (fast) {
            WAVE_SYNC();
            if constexpr (sizeof(OutT) == 4) {
                float* orow = (float*)out + (int64_t)row0 * ld + col0;
                constexpr int C4 = T / 4;
                for (int id = lane; id < T * C4; id += 64) {
                    const int r = id / C4, c4 = (id - C4 * r) * 4;
                    *reinterpret_cast<f4*>(orow + (int64_t)r * ld + c4) = *reinterpret_cast<const f4*>(TT + r * LDT2 + c4);
                }
            }
        }
    }
}

// ---- forward, "run" variant: 16-byte stores ------------------------------------------------------------------------
// The pair kernel above is store-ISSUE bound (ablation, K_ZX at C4: 121 us, 89 without the stores, 87 without the MFMA
// product, 94 without the P2 loads, 49 without stores and MFMA): every lane writes its 6 x 6 micro-block as 18 8-byte pieces.
// Here a lane owns a RUN of 12 consecutive output columns (two micro-blocks at q = 6, four at q = 3) of its Q rows: 48 bytes
// per row, 16-byte aligned, three dwordx4 stores -- half the store instructions per byte.  The tile is 48 rows x TC columns
// (TC = 96 at q = 6, 48 at q = 3: 64 lanes = 48/Q point rows x TC/12 runs), the packed rows of the NEXT column tile travel
// through registers while the current one is transformed (the global-load latency no longer sits on the per-tile chain), and
// the T' tile overlays the P2 image it was computed from (the wave is alone in its workgroup: barriers order the phases).
#ifndef FWD_RUN_WGS_
#define FWD_RUN_WGS_ (256 * 12)
#endif
#ifndef FWD_RUN
#define FWD_RUN 0           // measured slower than the pair kernel (128-135 vs 87-123 us for K_ZX at C4): kept for the record, not dispatched
#endif
template <typename OutT, int Q>
__global__ __launch_bounds__(64) void kernel_fwd_run_kernel(const float* __restrict__ P1, const float* __restrict__ self1,
                                                            int n1q, const float* __restrict__ P2,
                                                            const float* __restrict__ self2, int n2q, int K4, int DP,
                                                            const float* __restrict__ hyp, float jitter,
                                                            OutT* __restrict__ out, int64_t ld) {
    constexpr int R = 48 / Q, NRUN = 64 / R, PPR = 12 / Q, TC = NRUN * 12, NCT = TC / 16;
    constexpr int LDTT = TC + 4;                        // T' row stride (floats): rows 16-byte aligned
    constexpr int NPF = (TC * 32 / 4 + 63) / 64;        // float4 per lane that hold a prefetched P2 tile (DP <= 32)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = K4 + 5;
    float* P1s = smem;                                  // [48][LDP]
    float* P2s = P1s + 48 * LDP;                        // [TC][LDP], later overlaid by TT [48][LDTT]
    float* TT = P2s;
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * 48;
    const int ncoltiles = (n2q + TC - 1) / TC;
    const int KS = K4 / 4 + 1;
    const int pch = DP / 4;                             // float4 chunks per packed row
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;

    for (int e = lane; e < 48 * LDP; e += 64) P1s[e] = 0.f;
    WAVE_SYNC();
    for (int e = lane; e < 48 * K4; e += 64) {
        const int r = e / K4, k = e - r * K4;
        if (row0 + r < n1q) P1s[r * LDP + k] = P1[(int64_t)(row0 + r) * DP + k];
    }
    if (lane < 48 && row0 + lane < n1q) {
        const int a = lane % Q;
        P1s[lane * LDP + K4 + 1] = a == 0 ? 1.f : 0.f;
        P1s[lane * LDP + K4 + 2] = a == 0 ? 0.f : -self1[row0 + lane];
    }
    // this lane's run: point row pi (rows pr0 .. pr0 + Q - 1), columns pc0 .. pc0 + 11 of the tile
    const int pi = lane / NRUN, run = lane - pi * NRUN;
    const int pr0 = pi * Q, pc0 = run * 12;
    const bool rowok = row0 + pr0 < n1q;
    const float s1r0 = rowok ? self1[row0 + pr0] : 0.f;
    const int nchunk = TC * pch;                        // float4 chunks of one P2 tile

    f4 pf[NPF];
    float pself = 0.f;                                  // -self2 of tile row `lane` (+ lane + 64 at TC = 96)
    float pself2 = 0.f;
    auto prefetch = [&](int ct) {
        const int col0 = ct * TC;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = lane + 64 * u;
            const int r = e / pch, k = (e - r * pch) * 4;
            pf[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (e < nchunk && col0 + r < n2q) pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)(col0 + r) * DP + k);
        }
        pself = (lane < TC && col0 + lane < n2q) ? -self2[col0 + lane] : 0.f;
        if (TC > 64) pself2 = (lane + 64 < TC && col0 + lane + 64 < n2q) ? -self2[col0 + lane + 64] : 0.f;
    };
    int ct = blockIdx.x;
    if (ct < ncoltiles) prefetch(ct);
    for (; ct < ncoltiles; ct += gridDim.x) {
        const int col0 = ct * TC;
        WAVE_SYNC();                                // the previous tile's T' reads are done: the overlay may be rewritten
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = lane + 64 * u;
            if (e < nchunk) {
                const int r = e / pch, k = (e - r * pch) * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t) P2s[r * LDP + k + t] = pf[u][t];
            }
        }
        WAVE_SYNC();
        if (lane < TC) {
            P2s[lane * LDP + K4 + 1] = pself;
            P2s[lane * LDP + K4 + 2] = (col0 + lane < n2q && lane % Q == 0) ? 1.f : 0.f;
        }
        if (TC > 64 && lane + 64 < TC) {
            P2s[(lane + 64) * LDP + K4 + 1] = pself2;
            P2s[(lane + 64) * LDP + K4 + 2] = (col0 + lane + 64 < n2q && (lane + 64) % Q == 0) ? 1.f : 0.f;
        }
        float s2v[PPR];
#pragma unroll
        for (int pp = 0; pp < PPR; ++pp) s2v[pp] = (col0 + pc0 + pp * Q < n2q) ? self2[col0 + pc0 + pp * Q] : 0.f;
        if (ct + (int)gridDim.x < ncoltiles) prefetch(ct + gridDim.x);       // in flight under everything below
        WAVE_SYNC();
        f4 t[3][NCT];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) t[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        {
            const float* pa = P1s + m16 * LDP + kg;
            const float* pb = P2s + m16 * LDP + kg;
            for (int ks = 0; ks < KS; ++ks) {
                float av[3], bv[NCT];
#pragma unroll
                for (int i = 0; i < 3; ++i) av[i] = pa[i * 16 * LDP + ks * 4];
#pragma unroll
                for (int j = 0; j < NCT; ++j) bv[j] = pb[j * 16 * LDP + ks * 4];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < NCT; ++j) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], t[i][j], 0, 0, 0);
            }
        }
        WAVE_SYNC();                                // every P2 fragment has been read: T' may overlay the image
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) TT[(i * 16 + kg * 4 + r) * LDTT + j * 16 + m16] = t[i][j][r];
        WAVE_SYNC();
        if (rowok && col0 + pc0 < n2q) {
            float tq[Q][12];
#pragma unroll
            for (int a = 0; a < Q; ++a)
#pragma unroll
                for (int c4 = 0; c4 < 3; ++c4) {
                    const f4 v = *reinterpret_cast<const f4*>(TT + (pr0 + a) * LDTT + pc0 + 4 * c4);
                    tq[a][4 * c4] = v[0]; tq[a][4 * c4 + 1] = v[1]; tq[a][4 * c4 + 2] = v[2]; tq[a][4 * c4 + 3] = v[3];
                }
            float val[Q][12];
#pragma unroll
            for (int pp = 0; pp < PPR; ++pp) {
                const int o = pp * Q;
                const float nn = fmaxf(s1r0 - s2v[pp] - 2.f * tq[0][o], 0.f);       // covar_dist clamps at 0
                const float k = s * expf(-0.5f * nn);                                // postprocess_rbf, ScaleKernel
                const float kil = k * il, kil2 = k * il2;
                const bool diag = jitter != 0.f && (int64_t)row0 + pr0 == (int64_t)col0 + pc0 + o;
#pragma unroll
                for (int a = 0; a < Q; ++a) {
                    if (a == 0) {
                        val[0][o] = k;
#pragma unroll
                        for (int b = 1; b < Q; ++b) val[0][o + b] = tq[0][o + b] * kil;                              // w_b k / ell
                    } else {
                        val[a][o] = tq[a][o] * kil;                                                                  // -u_a k / ell
#pragma unroll
                        for (int b = 1; b < Q; ++b) val[a][o + b] = (tq[a][o + b] + tq[a][o] * tq[0][o + b]) * kil2;  // (G_ab - u_a w_b) k / ell^2
                    }
                    if (diag) val[a][o + a] += jitter;
                }
            }
            OutT* op = out + ((int64_t)row0 + pr0) * ld + col0 + pc0;
            const int vc = min(12, n2q - (col0 + pc0));                  // valid columns of the run (a multiple of Q)
            if (vc == 12) {
                using O4 = OutT __attribute__((ext_vector_type(16 / sizeof(OutT))));
                constexpr int EPV = 16 / sizeof(OutT);
#pragma unroll
                for (int a = 0; a < Q; ++a)
#pragma unroll
                    for (int c = 0; c < 12; c += EPV) {
                        O4 v;
#pragma unroll
                        for (int e = 0; e < EPV; ++e) v[e] = (OutT)val[a][c + e];
                        *reinterpret_cast<O4*>(op + a * ld + c) = v;
                    }
            } else {
#pragma unroll
                for (int a = 0; a < Q; ++a)
#pragma unroll
                    for (int c = 0; c < 12; ++c)
                        if (c < vc) op[a * ld + c] = (OutT)val[a][c];
            }
        }
    }
}

__global__ void kernel_diag_kernel(int n, int p, const float* __restrict__ hyp, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * (p + 1)) return;
    const float ell = hyp[0], s = hyp[1];
    out[idx] = (idx % (p + 1) == 0) ? s : s / (ell * ell);
}

// ---- backward -------------------------------------------------------------------------------------
// grid = (nsplit, row tiles).  Workgroup (sx, by) owns Rr points (Tr = Rr q <= 48 rows for q <= 48) of side 1 and
// sweeps column tiles sx, sx+nsplit, ... (Rc points, Tc = Rc q <= 96 columns).  Per tile:
//   1. the upstream tile Gbar and the P2 packs arrive through registers (loads for tile t+1 are in flight while
//      tile t is transformed), T = P1 P2^T is recomputed on MFMA;
//   2. ROW pass, one thread per (tile row, point column): reads its 1 x q strip of Gbar / T, writes Tbar_a0 and
//      Tbar_ab in place, the strip's share of <Gbar, dK/dk> and u_a to small side arrays;
//   3. COLUMN pass, one thread per (point row, tile column): Tbar_0b (column sums over a) and Tbar_00;
//   4. dP1[Tr, NP] += Tbar[Tr, Tc] . P2ext[Tc, NP] on MFMA, accumulators stay in registers over the sweep.
// Both passes are q-fold parallel within a micro-block with in-thread reductions only (the earlier
// one-thread-per-point-pair transform was LDS-latency bound at 4 waves per CU).
#ifndef BWD_ABLATE
#define BWD_ABLATE 0      // tools only: bit 0 = no row/column passes, bit 1 = no dP1 MFMA, bit 2 = no T MFMA, bit 3 = no Gbar loads
#endif
#ifndef BWD_NT_
#define BWD_NT_ 384
#endif
#ifndef BWD_WGS_
#define BWD_WGS_ 512
#endif
constexpr int BWD_NT = BWD_NT_;                 // 6 waves: 2 row + 2 column tasks per thread on a 48 x 96 tile
constexpr int BWD_NW = BWD_NT / 64;
constexpr int BWD_MAXACC = (36 + BWD_NW - 1) / BWD_NW;   // 16x16 dP1 tiles per wave, worst case 6 x 6
constexpr int BWD_GCH = (48 * (TMAX / 4) + BWD_NT - 1) / BWD_NT;       // 4-wide Gbar chunks per thread (Tr <= 48)
constexpr int BWD_GCH_MAX = (TMAX * (TMAX / 4) + BWD_NT - 1) / BWD_NT; // ... when one point fills the tile (q > 48)
constexpr int BWD_TARGET_WGS = BWD_WGS_;         // 2 resident workgroups per CU

template <typename GT> struct GVec;
template <> struct GVec<float> { using type = float __attribute__((ext_vector_type(4))); };
template <> struct GVec<double> { using type = double __attribute__((ext_vector_type(4))); };

template <typename GT>
__device__ __forceinline__ f4 load_g4(const GT* __restrict__ G, int64_t ldg, int64_t gr, int64_t gc, int r, int c,
                                      int Tr, int Tc, int n1q, int n2q, bool vec) {
    f4 o = {0.f, 0.f, 0.f, 0.f};
    if (r >= Tr || gr >= n1q || c >= Tc) return o;
    const GT* src = G + gr * ldg + gc;
    if (vec && c + 3 < Tc && gc + 3 < n2q) {
        const typename GVec<GT>::type v = *reinterpret_cast<const typename GVec<GT>::type*>(src);
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (c + t < Tc && gc + t < n2q) o[t] = (float)src[t];
    }
    return o;
}

template <typename GT, int Q, int NCH>
__global__ __launch_bounds__(BWD_NT) void kernel_bwd_kernel(const GT* __restrict__ G, int64_t ldg,
                                                            const float* __restrict__ P1, const float* __restrict__ self1,
                                                            int n1q, const float* __restrict__ P2,
                                                            const float* __restrict__ self2, int n2q, int q_rt, int Rr,
                                                            int Rc, int K4, int DP, int NP, int gvec,
                                                            const float* __restrict__ hyp, float* __restrict__ slab,
                                                            float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = BWD_NT, NW = BWD_NW;
    const int q = Q > 0 ? Q : q_rt, p = q - 1;
    const int LDP1 = K4 + 1, LDP = NP + 1;
    const int Tr = Rr * q, Tc = Rc * q;
    const int Trp = (Tr + 15) & ~15, Tcp = (Tc + 15) & ~15;
    float* P1s = smem;                      // [Trp][LDP1]
    float* P2s = P1s + Trp * LDP1;          // [Tcp][LDP]   B operand of both MFMA products
    float* Ts = P2s + Tcp * LDP;            // [Trp][LDT]
    float* Gs = Ts + Trp * LDT;             // [Trp][LDT]   Gbar, then Tbar in place
    float* s1 = Gs + Trp * LDT;             // [Trp]
    float* s2 = s1 + Trp;                   // [Tcp]
    float* KK = s2 + Tcp;                   // [Rr][Rc]     k per point pair
    float* Us = KK + Rr * Rc;               // [Tr][Rc]     u_a = r . v1_a
    float* Ps = Us + Tr * Rc;               // [Tr][Rc]     per-strip share of kbar
    float* red = Ps + Tr * Rc;              // [2 NW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntr = Trp / 16, ntc = Tcp / 16, nnp = NP / 16;
    const int row0 = blockIdx.y * Tr;
    const int ncoltiles = (n2q + Tc - 1) / Tc;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    const float invq = 1.f / (float)q, invRc = 1.f / (float)Rc, invTc = 1.f / (float)Tc;
    const int gch_row = Tcp / 4;            // 4-wide Gbar chunks per tile row
    const int pch_row = DP / 4;             // 4-wide chunks per packed row
    const bool vec = gvec != 0;
    const bool ppre = Tcp * pch_row <= 2 * NT;      // P2 packs ride in registers too when two chunks per thread cover them

    stage_pack(P1s, s1, P1, self1, row0, Tr, n1q, Trp, DP, K4, LDP1);
    for (int e = tid; e < Tcp * LDP; e += NT) P2s[e] = 0.f;       // columns [DP, NP) and pad rows stay zero
    for (int e = tid; e < Trp * LDT; e += NT) Gs[e] = 0.f;

    f4 greg[NCH], preg[2];
    float s2reg = 0.f;
    auto prefetch = [&](int ct) {
        const int col0 = ct * Tc;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * NT;
            const int r = e / gch_row, c = (e - r * gch_row) * 4;
            if (BWD_ABLATE & 8) greg[i] = f4{1.f, 1.f, 1.f, 1.f};
            else greg[i] = load_g4(G, ldg, (int64_t)row0 + r, (int64_t)col0 + c, r, c, Tr, Tc, n1q, n2q, vec);
        }
        if (ppre) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * NT;
                const int r = e / pch_row, k = (e - r * pch_row) * 4;
                const int gr = col0 + r;
                preg[i] = (r < Tc && gr < n2q) ? *reinterpret_cast<const f4*>(P2 + (int64_t)gr * DP + k) : f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        s2reg = (tid < Tc && col0 + tid < n2q) ? self2[col0 + tid] : 0.f;
    };

    f4 acc[BWD_MAXACC];
#pragma unroll
    for (int i = 0; i < BWD_MAXACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float sK_sum = 0.f, l_acc = 0.f;

    int ct = blockIdx.x;
    if (ct < ncoltiles) prefetch(ct);
    for (; ct < ncoltiles; ct += gridDim.x) {
        __syncthreads();  // previous iteration's MFMA reads of P2s / Gs are done (first: P1s / zero fill visible)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * NT;
            const int r = e / gch_row, c = (e - r * gch_row) * 4;
            if (r < Trp) *reinterpret_cast<f4*>(Gs + r * LDT + c) = greg[i];
        }
        if (ppre) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * NT;
                const int r = e / pch_row, k = (e - r * pch_row) * 4;
                if (r < Tcp) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) P2s[r * LDP + k + t] = preg[i][t];
                }
            }
        } else {
            const int col0 = ct * Tc;
            for (int e = tid; e < Tcp * DP; e += NT) {
                const int r = e / DP, k = e - r * DP;
                P2s[r * LDP + k] = (r < Tc && col0 + r < n2q) ? P2[(int64_t)(col0 + r) * DP + k] : 0.f;
            }
        }
        if (tid < Tcp) s2[tid] = s2reg;
        if (ct + (int)gridDim.x < ncoltiles) prefetch(ct + gridDim.x);
        __syncthreads();

        // T = P1s P2s^T
        for (int id = wave; id < ntr * ntc && !(BWD_ABLATE & 4); id += NW) {
            const int tr = id / ntc, tc = id - tr * ntc;
            f4 t4 = {0.f, 0.f, 0.f, 0.f};
            const float* pa = P1s + (tr * 16 + (lane & 15)) * LDP1 + (lane >> 4);
            const float* pb = P2s + (tc * 16 + (lane & 15)) * LDP + (lane >> 4);
            for (int kk = 0; kk < K4; kk += 4) t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk], pb[kk], t4, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Ts[(tr * 16 + (lane >> 4) * 4 + r) * LDT + tc * 16 + (lane & 15)] = t4[r];
        }
        __syncthreads();

        // ROW pass: task = (tile row r = (point pi, slot a), point column pj)
        for (int task = tid; task < Tr * Rc && !(BWD_ABLATE & 1); task += NT) {
            const int r = fdiv_small(task, invRc), pj = task - r * Rc;
            const int pi = fdiv_small(r, invq), a = r - pi * q;
            const int r0 = pi * q, c0 = pj * q;
            float* gr_ = Gs + r * LDT + c0;
            const float* tr_ = Ts + r * LDT + c0;
            const float* t0_ = Ts + r0 * LDT + c0;
            const float* s2_ = s2 + c0;
            const float nn = fmaxf(s1[r0] + s2_[0] - 2.f * t0_[0], 0.f);
            const float k = s * expf(-0.5f * nn);
            if constexpr (Q > 0) {
                // all LDS reads first (Gs / Ts may alias for the compiler), then the in-place stores
                float gv[Q], tv[Q], t0v[Q], s2v[Q];
#pragma unroll
                for (int b = 0; b < Q; ++b) { gv[b] = gr_[b]; tv[b] = tr_[b]; t0v[b] = t0_[b]; s2v[b] = s2_[b]; }
                const float g0 = gv[0];
                if (a == 0) {
                    float first = 0.f;
#pragma unroll
                    for (int b = 1; b < Q; ++b) first = __builtin_fmaf(gv[b], t0v[b] - s2v[b], first);
                    first *= il;
                    Ps[task] = g0 + first;
                    KK[pi * Rc + pj] = k;
                    l_acc = __builtin_fmaf(k, first, l_acc);                  // first-order entries of <Gbar, K>
                } else {
                    const float u = s1[r] - tv[0];
                    float hs = 0.f, gw = 0.f;
                    const float kil2 = k * il2;
#pragma unroll
                    for (int b = 1; b < Q; ++b) {
                        const float w = t0v[b] - s2v[b];
                        hs = __builtin_fmaf(gv[b], tv[b] - u * w, hs);
                        gw = __builtin_fmaf(gv[b], w, gw);
                        gr_[b] = kil2 * gv[b];                                // Tbar_ab
                    }
                    hs *= il2;
                    const float ubar = k * (-g0 * il - gw * il2);
                    gr_[0] = -ubar;                                           // Tbar_a0
                    Us[task] = u;
                    Ps[task] = hs - g0 * u * il;
                    l_acc += k * (2.f * hs - g0 * u * il) + ubar * u;
                }
            } else {
                const float g0 = gr_[0];
                if (a == 0) {
                    float first = 0.f;
                    for (int b = 1; b <= p; ++b) first = __builtin_fmaf(gr_[b], t0_[b] - s2_[b], first);
                    first *= il;
                    Ps[task] = g0 + first;
                    KK[pi * Rc + pj] = k;
                    l_acc = __builtin_fmaf(k, first, l_acc);
                } else {
                    const float u = s1[r] - tr_[0];
                    float hs = 0.f, gw = 0.f;
                    const float kil2 = k * il2;
                    for (int b = 1; b <= p; ++b) {
                        const float w = t0_[b] - s2_[b];
                        const float g = gr_[b];
                        hs = __builtin_fmaf(g, tr_[b] - u * w, hs);
                        gw = __builtin_fmaf(g, w, gw);
                        gr_[b] = kil2 * g;
                    }
                    hs *= il2;
                    const float ubar = k * (-g0 * il - gw * il2);
                    gr_[0] = -ubar;
                    Us[task] = u;
                    Ps[task] = hs - g0 * u * il;
                    l_acc += k * (2.f * hs - g0 * u * il) + ubar * u;
                }
            }
        }
        __syncthreads();

        // COLUMN pass: task = (point row pi, tile column c = (point pj, slot b))
        for (int task = tid; task < Rr * Tc && !(BWD_ABLATE & 1); task += NT) {
            const int pi = fdiv_small(task, invTc), c = task - pi * Tc;
            const int pj = fdiv_small(c, invq), b = c - pj * q;
            const int r0 = pi * q;
            const float k = KK[pi * Rc + pj];
            float* g0c = Gs + r0 * LDT + c;
            if (b == 0) {
                float kbar = 0.f;
#pragma unroll
                for (int a = 0; a <= p; ++a) kbar += Ps[(r0 + a) * Rc + pj];
                const float t00 = k * kbar;                                   // Tbar_00 = <Gbar, K> of the micro-block
                const float nn = fmaxf(s1[r0] + s2[c] - 2.f * Ts[r0 * LDT + c], 0.f);
                *g0c = t00;
                sK_sum += t00;
                l_acc = __builtin_fmaf(-t00, nn, l_acc);
            } else {
                const float w = Ts[r0 * LDT + c] - s2[c];
                float wbar = k * il * *g0c;
#pragma unroll
                for (int a = 1; a <= p; ++a) wbar = __builtin_fmaf(-g0c[a * LDT], Us[(r0 + a) * Rc + pj], wbar);
                *g0c = wbar;                                                  // Tbar_0b
                l_acc = __builtin_fmaf(wbar, w, l_acc);
            }
        }
        __syncthreads();

        // dP1[Trp, NP] += Tbar[Trp, Tcp] . P2ext[Tcp, NP]   (A from Gs, B from P2s)
#pragma unroll
        for (int si = 0; si < BWD_MAXACC; ++si) {
            const int id = wave + NW * si;
            if (id < ntr * nnp && !(BWD_ABLATE & 2)) {
                const int tr = id / nnp, tn = id - tr * nnp;
                const float* pa = Gs + (tr * 16 + (lane & 15)) * LDT + (lane >> 4);
                const float* pb = P2s + (lane >> 4) * LDP + tn * 16 + (lane & 15);
                f4 c4 = acc[si];
                for (int kk = 0; kk < Tcp; kk += 4) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk], pb[kk * LDP], c4, 0, 0, 0);
                acc[si] = c4;
            }
        }
    }

    // partial slab: slab[split][row][NP]
    float* myslab = slab + ((int64_t)blockIdx.x * n1q) * NP;
#pragma unroll
    for (int si = 0; si < BWD_MAXACC; ++si) {
        const int id = wave + NW * si;
        if (id < ntr * nnp) {
            const int tr = id / nnp, tn = id - tr * nnp;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = tr * 16 + (lane >> 4) * 4 + r;
                const int64_t gr = row0 + rr;
                if (rr < Tr && gr < n1q) myslab[gr * NP + tn * 16 + (lane & 15)] = acc[si][r];
            }
        }
    }
    // block reduce the two scalars
    float l_sum = -il * l_acc;
    for (int off = 32; off > 0; off >>= 1) {
        sK_sum += __shfl_down(sK_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    __syncthreads();
    if (lane == 0) { red[wave * 2] = sK_sum; red[wave * 2 + 1] = l_sum; }
    __syncthreads();
    if (tid == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int w = 0; w < NW; ++w) { a0 += red[2 * w]; a1 += red[2 * w + 1]; }
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        partials[bid * 2] = a0;
        partials[bid * 2 + 1] = a1;
    }
}

// ---- backward, register-resident variant for small compile-time q ------------------------------------------
// ONE WAVE per workgroup (no barriers): the wave owns R = 48/Q points of side 1 (T = R Q <= 48 rows) and sweeps
// 48-column tiles.  Lane <-> point pair(s): the Q x Q micro-block of Gbar comes straight from HBM into registers,
// T' = P1' P2'^T comes from MFMA through LDS, the whole micro-block transform runs in registers and Tbar goes back
// to LDS in place as the A operand of dP1 += Tbar P2ext.  Two extra packed columns fold the self terms into T':
//   P1'[r, K4+1] = [a == 0],  P2'[c, K4+1] = -self2[c]      =>  T'[r0, cb] = x1~.v2_b - beta_b = w_b
//   P1'[r, K4+2] = -self1[r] [a > 0],  P2'[c, K4+2] = [b == 0]  =>  T'[ra, c0] = v1_a.x2~ - alpha_a = -u_a
// so the transform reads nothing but its own 36 T' values.  The A fragments of the wave's 48 side-1 rows stay in registers for
// the whole sweep (no LDS image of P1'), the T' product is one straight-line copy per K depth: ~16 KB of LDS per wave, 8 waves
// per CU hide each other's LDS / HBM latency (K_ZX-bar at C4, whole dsvgp_kernel_bwd: 194 -> 157 us on one box).
constexpr int PAIR_LDT = 52;     // LDS row stride of the 48 x 48 T' / Tbar tile (even: 8-byte strips)
#ifndef PAIR_WGS_
#define PAIR_WGS_ (256 * 8)       // 8 waves per CU (registers: 2 per SIMD; LDS: 16 KB per wave); 7 / 8 / 9 / 10 / 12: 169 / 157 / 262 / 246 / 225 us
#endif
#ifndef PAIR_ABLATE
#define PAIR_ABLATE 0      // tools only: bit 0 = no transform, bit 1 = no dP1 MFMA, bit 2 = no T' MFMA, bit 3 = no Gbar loads
#endif
constexpr int PAIR_WGS = PAIR_WGS_;

#ifndef BWDP_MINW
#define BWDP_MINW 2        // waves per SIMD the register allocation leaves room for: 2 (<= 256 VGPRs; the micro-block transform holds
#endif                     // 36 upstream + 36 T' + 24 A-fragment + 24 accumulator registers); 3 (<= 168) spills: 270 us instead of 157
#ifndef BWDP_LEAN
#define BWDP_LEAN 1        // tiles inside the matrix: packed side-2 rows fetched and staged without predicates (142 -> 138 us; the same for
#endif                     // the upstream micro-block loads measured SLOWER: 167 us)
#ifndef BWDP_PREFETCH
#define BWDP_PREFETCH 0    // 1: the next tile's packed side-2 rows travel through registers under the current tile (36 more registers,
#endif                     //    same time at 8 waves per CU: 157 us either way)
// KSM: as for kernel_fwd_pair_kernel (8: DP <= 32, 16: DP <= 64); the dP1 accumulators cover NP = 4 KSM packed columns
template <typename GT, int Q, int KSM = 8>
__global__ __launch_bounds__(64, KSM > 8 ? 1 : BWDP_MINW) void kernel_bwd_pair_kernel(const GT* __restrict__ G, int64_t ldg,
                                                             const float* __restrict__ P1, const float* __restrict__ self1,
                                                             int n1q, const float* __restrict__ P2,
                                                             const float* __restrict__ self2, int n2q, int K4, int DP,
                                                             int NP, int gvec, const float* __restrict__ hyp,
                                                             float* __restrict__ slab, float* __restrict__ partials) {
    constexpr int R = 48 / Q, T = R * Q;
    constexpr int NPAIR = R * R, PPL = (NPAIR + 63) / 64;      // point pairs per lane
    constexpr int LDT2 = PAIR_LDT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = NP + 1;
    // (the A fragments of the wave's 48 side-1 rows stay in registers for the whole sweep: no LDS image of P1')
    float* P2s = smem;                  // [48][LDP]   extended side-2 packs of the current tile
    float* TT = P2s + ((49 * LDP + 3) & ~3);      // [48][LDT2]  T', then Tbar in place (16-byte aligned: its 8-byte strips); row 48 of P2s: scratch of the predicate-free staging
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * T;
    const int ncoltiles = (n2q + T - 1) / T;
    const int nnp = NP / 16;            // 1 .. KSM / 4
    constexpr int NNP = KSM / 4;
    const int KS = K4 / 4 + 1;
    const int pch = DP / 4;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    const bool vec = gvec != 0;

    for (int e = lane; e < 48 * LDP; e += 64) P2s[e] = 0.f;
    float areg[3][KSM];                 // areg[i][ks] = P1'[row0 + 16 i + m16][4 ks + kg]  (KS <= KSM since DP <= 4 KSM)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int gr = row0 + i * 16 + m16;
        const bool ok = i * 16 + m16 < T && gr < n1q;
        const int a = (i * 16 + m16) % Q;
        const float sf = ok ? self1[gr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            float v = 0.f;
            if (ks < K4 / 4) v = ok ? P1[(int64_t)gr * DP + ks * 4 + kg] : 0.f;
            else if (ks == K4 / 4 && ok) v = (kg == 1) ? (a == 0 ? 1.f : 0.f) : ((kg == 2) ? (a == 0 ? 0.f : -sf) : 0.f);
            areg[i][ks] = v;
        }
    }
    // the packed rows of the NEXT column tile travel through registers under the current tile's work
    constexpr int NPFP = (48 * KSM + 63) / 64;     // float4 per lane of one P2 tile (DP <= 4 KSM)
    // (the packed rows of a column tile are one contiguous piece of P2: see kernel_fwd_pair_kernel)
    int pf_r[NPFP], pf_k[NPFP], pf_e[NPFP], pf_lds[NPFP];
#pragma unroll
    for (int u = 0; u < NPFP; ++u) {
        const int e = lane + 64 * u;
        const int r = e / pch;
        pf_r[u] = (e < T * pch) ? r : -1;
        pf_k[u] = (e - r * pch) * 4;
        const int ec = min(e, T * pch - 1), rc = ec / pch;
        pf_e[u] = ec;
        pf_lds[u] = (e < T * pch ? rc : 48) * LDP + (ec - rc * pch) * 4;
    }
    f4 pf[NPFP];
    float pselfv = 0.f;
    auto prefetch = [&](int ct_) {
        const int c0_ = ct_ * T;
        if (BWDP_LEAN && c0_ + T <= n2q) {               // a tile inside the matrix: no predicates
            const f4* src = reinterpret_cast<const f4*>(P2 + (int64_t)c0_ * DP);
#pragma unroll
            for (int u = 0; u < NPFP; ++u) pf[u] = src[pf_e[u]];
            pselfv = -self2[c0_ + min(lane, T - 1)];
            return;
        }
#pragma unroll
        for (int u = 0; u < NPFP; ++u) {
            pf[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (pf_r[u] >= 0 && c0_ + pf_r[u] < n2q) pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)(c0_ + pf_r[u]) * DP + pf_k[u]);
        }
        pselfv = (lane < T && c0_ + lane < n2q) ? -self2[c0_ + lane] : 0.f;
    };

    // lane-invariant pair geometry
    int pr0[PPL], pc0[PPL];
    float s1r0[PPL];
    bool prow[PPL];
#pragma unroll
    for (int pp = 0; pp < PPL; ++pp) {
        const int pid = lane + 64 * pp;
        const int pi = pid / R, pj = pid - pi * R;
        pr0[pp] = pi * Q; pc0[pp] = pj * Q;
        prow[pp] = pid < NPAIR && row0 + pr0[pp] < n1q;
        s1r0[pp] = prow[pp] ? self1[row0 + pr0[pp]] : 0.f;
    }

    f4 acc[3][NNP];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NNP; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float sK_sum = 0.f, l_acc = 0.f;

    if (BWDP_PREFETCH && (int)blockIdx.x < ncoltiles) prefetch(blockIdx.x);
    for (int ct = blockIdx.x; ct < ncoltiles; ct += gridDim.x) {
        const int col0 = ct * T;
        // upstream micro-blocks straight into registers (consumed after the T' product)
        float g[PPL][Q][Q];
        float s2c0[PPL];
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            const bool ok = prow[pp] && col0 + pc0[pp] < n2q && !(PAIR_ABLATE & 8);
            const GT* src = G + (int64_t)(row0 + pr0[pp]) * ldg + col0 + pc0[pp];
            s2c0[pp] = ok ? self2[col0 + pc0[pp]] : 0.f;
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                if constexpr (Q % 2 == 0) {
                    if (ok && vec) {
                        using V2 = GT __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int b = 0; b < Q; b += 2) {
                            const V2 v = *reinterpret_cast<const V2*>(src + a * ldg + b);
                            g[pp][a][b] = (float)v[0]; g[pp][a][b + 1] = (float)v[1];
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int b = 0; b < Q; ++b) g[pp][a][b] = ok ? (float)src[a * ldg + b] : 0.f;
            }
        }
        WAVE_SYNC();   // (single wave: orders the previous tile's MFMA reads of P2s / TT before the new stores)
        if (!BWDP_PREFETCH) prefetch(ct);
#pragma unroll
        for (int u = 0; u < NPFP; ++u) {               // (lanes past the end of the tile: the scratch row)
#pragma unroll
            for (int t = 0; t < 4; ++t) P2s[pf_lds[u] + t] = pf[u][t];
        }
        WAVE_SYNC();   // the extension columns go on top of the packed zeros
        if (lane < T) {
            P2s[lane * LDP + K4 + 1] = pselfv;
            P2s[lane * LDP + K4 + 2] = (col0 + lane < n2q && lane % Q == 0) ? 1.f : 0.f;
        }
        if (BWDP_PREFETCH && ct + (int)gridDim.x < ncoltiles) prefetch(ct + gridDim.x);      // in flight under the rest of this tile
        WAVE_SYNC();

        // T' = P1' P2'^T : 3 x 3 tiles of 16 x 16, K = K4 + 4  (one straight-line copy per K depth: no per-step branches, the
        // first product starts from the inline zero)
        if (!(PAIR_ABLATE & 4)) {
            f4 t[3][3];
            const float* pb = P2s + m16 * LDP + kg;
            auto product = [&](auto ksc) {
                constexpr int KS_ = decltype(ksc)::value;
#pragma unroll
                for (int ks = 0; ks < KS_; ++ks) {
                    float bv[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) bv[j] = pb[j * 16 * LDP + ks * 4];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][ks], bv[j], ks == 0 ? f4{0.f, 0.f, 0.f, 0.f} : t[i][j], 0, 0, 0);
                }
            };
            switch (KS) {
                case 1: product(std::integral_constant<int, 1>{}); break;
                case 2: product(std::integral_constant<int, 2>{}); break;
                case 3: product(std::integral_constant<int, 3>{}); break;
                case 4: product(std::integral_constant<int, 4>{}); break;
                case 5: product(std::integral_constant<int, 5>{}); break;
                case 6: product(std::integral_constant<int, 6>{}); break;
                case 7: product(std::integral_constant<int, 7>{}); break;
                case 8: product(std::integral_constant<int, 8>{}); break;
                default:
                    if constexpr (KSM > 8) {
                        switch (KS) {
                            case 9: product(std::integral_constant<int, 9>{}); break;
                            case 10: product(std::integral_constant<int, 10>{}); break;
                            case 11: product(std::integral_constant<int, 11>{}); break;
                            case 12: product(std::integral_constant<int, 12>{}); break;
                            case 13: product(std::integral_constant<int, 13>{}); break;
                            case 14: product(std::integral_constant<int, 14>{}); break;
                            case 15: product(std::integral_constant<int, 15>{}); break;
                            default: product(std::integral_constant<int, 16>{}); break;
                        }
                    } else product(std::integral_constant<int, 8>{});
                    break;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) TT[(i * 16 + kg * 4 + r) * LDT2 + j * 16 + m16] = t[i][j][r];
        }
        WAVE_SYNC();

        // micro-block transform in registers
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            if (lane + 64 * pp < NPAIR && !(PAIR_ABLATE & 1)) {
                float* blk = TT + pr0[pp] * LDT2 + pc0[pp];
                float tq[Q][Q];
#pragma unroll
                for (int a = 0; a < Q; ++a) {
                    if constexpr (Q % 2 == 0) {
                        using F2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int b = 0; b < Q; b += 2) {
                            const F2 v = *reinterpret_cast<const F2*>(blk + a * LDT2 + b);
                            tq[a][b] = v[0]; tq[a][b + 1] = v[1];
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < Q; ++b) tq[a][b] = blk[a * LDT2 + b];
                    }
                }
                const float nn = fmaxf(s1r0[pp] - s2c0[pp] - 2.f * tq[0][0], 0.f);
                const float k = s * expf(-0.5f * nn);
                const float kil = k * il, kil2 = k * il2;
                float first = 0.f, second = 0.f, hsum = 0.f, dots = 0.f;
                float gu[Q];
#pragma unroll
                for (int b = 0; b < Q; ++b) gu[b] = 0.f;
                float out[Q][Q];
#pragma unroll
                for (int b = 1; b < Q; ++b) first = __builtin_fmaf(g[pp][0][b], tq[0][b], first);        // sum g0b w_b
#pragma unroll
                for (int a = 1; a < Q; ++a) {
                    const float u = -tq[a][0];
                    const float ga0 = g[pp][a][0];
                    float gw = 0.f, gt = 0.f;
#pragma unroll
                    for (int b = 1; b < Q; ++b) {
                        const float gab = g[pp][a][b];
                        gw = __builtin_fmaf(gab, tq[0][b], gw);
                        gt = __builtin_fmaf(gab, tq[a][b], gt);
                        gu[b] = __builtin_fmaf(gab, u, gu[b]);
                        out[a][b] = kil2 * gab;                                                        // Tbar_ab
                    }
                    second = __builtin_fmaf(ga0, u, second);
                    hsum += gt - u * gw;
                    const float ubar = -(kil * ga0 + kil2 * gw);
                    out[a][0] = -ubar;                                                                 // Tbar_a0
                    dots = __builtin_fmaf(ubar, u, dots);
                }
                out[0][0] = 0.f;
#pragma unroll
                for (int b = 1; b < Q; ++b) {
                    const float wbar = kil * g[pp][0][b] - kil2 * gu[b];
                    out[0][b] = wbar;                                                                  // Tbar_0b
                    dots = __builtin_fmaf(wbar, tq[0][b], dots);
                }
                const float e1 = il * (first - second), e2 = il2 * hsum;
                const float t00 = k * (g[pp][0][0] + e1 + e2);                                         // Tbar_00
                out[0][0] = t00;
                sK_sum += t00;
                l_acc += k * (e1 + 2.f * e2) - t00 * nn + dots;
#pragma unroll
                for (int a = 0; a < Q; ++a) {
                    if constexpr (Q % 2 == 0) {
                        using F2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int b = 0; b < Q; b += 2) *reinterpret_cast<F2*>(blk + a * LDT2 + b) = F2{out[a][b], out[a][b + 1]};
                    } else {
#pragma unroll
                        for (int b = 0; b < Q; ++b) blk[a * LDT2 + b] = out[a][b];
                    }
                }
            }
        }
        WAVE_SYNC();

        // dP1[48, NP] += Tbar[48, 48] . P2ext[48, NP]
        if (!(PAIR_ABLATE & 2)) {
            const float* pa = TT + m16 * LDT2 + kg;
            const float* pb = P2s + kg * LDP + m16;
#pragma unroll
            for (int kk = 0; kk < 48; kk += 4) {
                float av[3], bv[NNP];
#pragma unroll
                for (int i = 0; i < 3; ++i) av[i] = pa[i * 16 * LDT2 + kk];
#pragma unroll
                for (int j = 0; j < NNP; ++j) bv[j] = (j < nnp) ? pb[kk * LDP + 16 * j] : 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < NNP; ++j)
                        if (j < nnp) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

    float* myslab = slab + ((int64_t)blockIdx.x * n1q) * NP;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NNP; ++j)
            if (j < nnp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = i * 16 + kg * 4 + r;
                    const int64_t gr = row0 + rr;
                    if (rr < T && gr < n1q) myslab[gr * NP + j * 16 + m16] = acc[i][j][r];
                }
            }
    float l_sum = -il * l_acc;
    for (int off = 32; off > 0; off >>= 1) {
        sK_sum += __shfl_down(sK_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    if (lane == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        partials[bid * 2] = sK_sum;
        partials[bid * 2 + 1] = l_sum;
    }
}

// ---- backward, "split" variant for wide micro-blocks (round 5; q = 11: BASELINE config 3) -----------------------------------------
// The mapping of kernel_fwd_split_kernel applied to kernel_bwd_pair_kernel's data flow: ONE WAVE per workgroup owns R = 48 / Q points
// of side 1 (T = R Q rows) and sweeps T-column tiles; SPL = 64 / R^2 consecutive lanes share a point pair, lane (pair, sub) takes the
// micro-block rows a = sub, sub + SPL, ...  The upstream tile arrives through LDS (coalesced 16-byte loads instead of Q x Q scalars
// per lane), T' = P1' P2'^T from MFMA through LDS; a lane reads row 0 of both micro-blocks and its own rows, the sums that run over
// ALL rows of the micro-block (gu_b = sum_a g_ab u_a and three scalars) are added across the SPL lanes of the pair by two quad
// shuffles, and the lane with sub = 0 finishes row 0.  Tbar goes back to the T' tile in place: the A operand of dP1 += Tbar P2ext.
// Measured at the C3 geometry (M = 300, B = 512, d = 10, q = 11; K_ZX-bar 74 MB, whole dsvgp_kernel_bwd; profiles/r05_c_assemble_c3_c5.txt):
// generic kernel with runtime q 132 us, its compile-time q = 11 instance 111, this kernel 101-104 at 2048 waves (with or without the
// register prefetch of the next tile), 93 at 1024 waves; phase ablations at 1024 waves: no transform 75, no dP1 product 61, no T'
// product 83, no upstream loads 79, none of them 23 -- the phases ADD UP (one wave per SIMD: nothing overlaps them), which is what
// keeps it at 0.10 of the HBM roof.  Kept because it is the fastest of the three; the backward of wide micro-blocks is an open item.
#ifndef BWDS_PREFETCH
#define BWDS_PREFETCH 0
#endif
#ifndef BWDS_WGS
#define BWDS_WGS (256 * 4)
#endif
#ifndef BWDS_ABL
#define BWDS_ABL 0         // tools only (wrong results): bit 0 = no transform, 1 = no dP1 MFMA, 2 = no T' MFMA, 3 = no upstream loads
#endif
template <typename GT, int Q, int KSM>
__global__ __launch_bounds__(64) void kernel_bwd_split_kernel(const GT* __restrict__ G, int64_t ldg, const float* __restrict__ P1,
                                                              const float* __restrict__ self1, int n1q, const float* __restrict__ P2,
                                                              const float* __restrict__ self2, int n2q, int K4, int DP, int NP, int gvec,
                                                              const float* __restrict__ hyp, float* __restrict__ slab,
                                                              float* __restrict__ partials) {
    constexpr int R = 48 / Q, T = R * Q, NPAIR = R * R, SPL = 64 / NPAIR, RPL = (Q + SPL - 1) / SPL;
    constexpr int LDT2 = PAIR_LDT, LDG = PAIR_LDT, NNP = KSM / 4;      // (row stride 52: 16-byte rows, 20 banks apart)
    static_assert(NPAIR * SPL == 64 && SPL == 4 && T % 4 == 0, "split mapping: R^2 pairs x 4 lanes (quad shuffles), 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = NP + 1;
    float* P2s = smem;                                   // [49][LDP]  extended side-2 packs of the tile
    float* TT = P2s + ((49 * LDP + 3) & ~3);             // [48][LDT2] T', then Tbar in place
    float* GG = TT + 48 * LDT2;                          // [T][LDG]   upstream tile
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * T;
    const int ncoltiles = (n2q + T - 1) / T;
    const int nnp = NP / 16, KS = K4 / 4 + 1, pch = DP / 4;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;

    for (int e = lane; e < 49 * LDP; e += 64) P2s[e] = 0.f;
    for (int e = lane; e < 48 * LDT2; e += 64) TT[e] = 0.f;
    float areg[3][KSM];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int lr = i * 16 + m16, gr = row0 + lr;
        const bool ok = lr < T && gr < n1q;
        const int a = lr % Q;
        const float sf = ok ? self1[gr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            float v = 0.f;
            if (ks < K4 / 4) v = ok ? P1[(int64_t)gr * DP + ks * 4 + kg] : 0.f;
            else if (ks == K4 / 4 && ok) v = (kg == 1) ? (a == 0 ? 1.f : 0.f) : ((kg == 2) ? (a == 0 ? 0.f : -sf) : 0.f);
            areg[i][ks] = v;
        }
    }
    const int pid = lane / SPL, sub = lane - pid * SPL;
    const int pi = pid / R, pj = pid - pi * R;
    const int pr0 = pi * Q, pc0 = pj * Q;
    const bool prow = row0 + pr0 < n1q;
    const float s1r0 = prow ? self1[row0 + pr0] : 0.f;
    f4 acc[3][NNP];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NNP; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float sK_sum = 0.f, l_acc = 0.f;
    constexpr int C4 = T / 4;                              // 16-byte pieces per upstream tile row

    // the NEXT tile's upstream rows and packed side-2 rows travel through registers under the current tile's work
    constexpr int NGF = (T * C4 + 63) / 64, NPF = (T * KSM + 63) / 64;
    f4 gf[NGF], pf[NPF];
    auto prefetch = [&](int ct_) {
        const int c0_ = ct_ * T;
        const bool interior = row0 + T <= n1q && c0_ + T <= n2q;
#pragma unroll
        for (int u = 0; u < NGF; ++u) {
            const int id = lane + 64 * u, r = id / C4, c = (id - C4 * r) * 4;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (id < T * C4 && !(BWDS_ABL & 8)) {
                const GT* src = G + (int64_t)(row0 + r) * ldg + c0_ + c;
                if (interior && gvec) {
                    if constexpr (sizeof(GT) == 4) {
                        v = *reinterpret_cast<const f4*>(src);
                    } else {
                        using D2 = double __attribute__((ext_vector_type(2)));
                        const D2 v0 = *reinterpret_cast<const D2*>(src), v1 = *reinterpret_cast<const D2*>(src + 2);
                        v = f4{(float)v0[0], (float)v0[1], (float)v1[0], (float)v1[1]};
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (row0 + r < n1q && c0_ + c + t < n2q) v[t] = (float)src[t];
                }
            }
            gf[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = lane + 64 * u, r = e / pch, k = (e - r * pch) * 4;
            pf[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (e < T * pch && c0_ + r < n2q) pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)(c0_ + r) * DP + k);
        }
    };
    if (BWDS_PREFETCH && (int)blockIdx.x < ncoltiles) prefetch(blockIdx.x);
    for (int ct = blockIdx.x; ct < ncoltiles; ct += gridDim.x) {
        const int col0 = ct * T;
        const bool colok = prow && col0 + pc0 < n2q;
        const float s2c0 = colok ? self2[col0 + pc0] : 0.f;
        WAVE_SYNC();                                   // (single wave: the previous tile's MFMA reads of P2s / TT are done)
        if (!BWDS_PREFETCH) prefetch(ct);
#pragma unroll
        for (int u = 0; u < NGF; ++u) {
            const int id = lane + 64 * u, r = id / C4, c = (id - C4 * r) * 4;
            if (id < T * C4) *reinterpret_cast<f4*>(GG + r * LDG + c) = gf[u];
        }
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = lane + 64 * u, r = e / pch, k = (e - r * pch) * 4;
            if (e < T * pch) {
#pragma unroll
                for (int t = 0; t < 4; ++t) P2s[r * LDP + k + t] = pf[u][t];
            }
        }
        if (BWDS_PREFETCH && ct + (int)gridDim.x < ncoltiles) prefetch(ct + gridDim.x);
        WAVE_SYNC();
        if (lane < T) {
            const bool ok = col0 + lane < n2q;
            P2s[lane * LDP + K4 + 1] = ok ? -self2[col0 + lane] : 0.f;
            P2s[lane * LDP + K4 + 2] = (ok && lane % Q == 0) ? 1.f : 0.f;
        }
        WAVE_SYNC();
        if (!(BWDS_ABL & 4)) {   // T' = P1' P2'^T
            f4 t[3][3];
            const float* pb = P2s + m16 * LDP + kg;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                if (ks < KS) {
                    float bv[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) bv[j] = pb[j * 16 * LDP + ks * 4];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][ks], bv[j], ks == 0 ? f4{0.f, 0.f, 0.f, 0.f} : t[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) TT[(i * 16 + kg * 4 + r) * LDT2 + j * 16 + m16] = t[i][j][r];
        }
        WAVE_SYNC();
        // ---- micro-block transform (kernel_bwd_pair_kernel's arithmetic), rows split over the four lanes of a pair
        float* blk = TT + pr0 * LDT2 + pc0;
        const float* gb = GG + pr0 * LDG + pc0;
        if (!(BWDS_ABL & 1)) {
        float t0[Q], g0[Q], ta[RPL][Q], ga[RPL][Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) { t0[b] = blk[b]; g0[b] = colok ? gb[b] : 0.f; }
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int a = sub + SPL * i;
#pragma unroll
            for (int b = 0; b < Q; ++b) {
                ta[i][b] = (a < Q) ? blk[a * LDT2 + b] : 0.f;
                ga[i][b] = (a < Q && colok) ? gb[a * LDG + b] : 0.f;
            }
        }
        WAVE_SYNC();                                   // row 0 of the T' micro-block is rewritten below: every lane has read it
        const float nn = fmaxf(s1r0 - s2c0 - 2.f * t0[0], 0.f);
        const float k = s * expf(-0.5f * nn);
        const float kil = k * il, kil2 = k * il2;
        float gu[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) gu[b] = 0.f;
        float second = 0.f, hsum = 0.f, dots = 0.f;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int a = sub + SPL * i;
            if (a >= 1 && a < Q) {
                const float u = -ta[i][0];
                const float ga0 = ga[i][0];
                float gw = 0.f, gt = 0.f;
#pragma unroll
                for (int b = 1; b < Q; ++b) {
                    const float gab = ga[i][b];
                    gw = __builtin_fmaf(gab, t0[b], gw);
                    gt = __builtin_fmaf(gab, ta[i][b], gt);
                    gu[b] = __builtin_fmaf(gab, u, gu[b]);
                    blk[a * LDT2 + b] = kil2 * gab;                                            // Tbar_ab
                }
                second = __builtin_fmaf(ga0, u, second);
                hsum += gt - u * gw;
                const float ubar = -(kil * ga0 + kil2 * gw);
                blk[a * LDT2] = -ubar;                                                         // Tbar_a0
                dots = __builtin_fmaf(ubar, u, dots);
            }
        }
        // sums over all rows of the micro-block: across the four lanes of the pair (consecutive lanes: quad shuffles)
#pragma unroll
        for (int b = 1; b < Q; ++b) { gu[b] += __shfl_xor(gu[b], 1); gu[b] += __shfl_xor(gu[b], 2); }
        second += __shfl_xor(second, 1); second += __shfl_xor(second, 2);
        hsum += __shfl_xor(hsum, 1); hsum += __shfl_xor(hsum, 2);
        dots += __shfl_xor(dots, 1); dots += __shfl_xor(dots, 2);
        if (sub == 0) {
            float first = 0.f;
#pragma unroll
            for (int b = 1; b < Q; ++b) first = __builtin_fmaf(g0[b], t0[b], first);            // sum g0b w_b
#pragma unroll
            for (int b = 1; b < Q; ++b) {
                const float wbar = kil * g0[b] - kil2 * gu[b];
                blk[b] = wbar;                                                                 // Tbar_0b
                dots = __builtin_fmaf(wbar, t0[b], dots);
            }
            const float e1 = il * (first - second), e2 = il2 * hsum;
            const float t00 = k * (g0[0] + e1 + e2);                                           // Tbar_00
            blk[0] = t00;
            sK_sum += t00;
            l_acc += k * (e1 + 2.f * e2) - t00 * nn + dots;
        }
        }
        WAVE_SYNC();
        if (!(BWDS_ABL & 2)) {   // dP1[48, NP] += Tbar[48, 48] . P2ext[48, NP]
            const float* pa = TT + m16 * LDT2 + kg;
            const float* pb = P2s + kg * LDP + m16;
#pragma unroll
            for (int kk = 0; kk < 48; kk += 4) {
                float av[3], bv[NNP];
#pragma unroll
                for (int i = 0; i < 3; ++i) av[i] = pa[i * 16 * LDT2 + kk];
#pragma unroll
                for (int j = 0; j < NNP; ++j) bv[j] = (j < nnp) ? pb[kk * LDP + 16 * j] : 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < NNP; ++j)
                        if (j < nnp) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* myslab = slab + ((int64_t)blockIdx.x * n1q) * NP;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NNP; ++j)
            if (j < nnp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = i * 16 + kg * 4 + r;
                    const int64_t gr = row0 + rr;
                    if (rr < T && gr < n1q) myslab[gr * NP + j * 16 + m16] = acc[i][j][r];
                }
            }
    float l_sum = -il * l_acc;
    for (int off = 32; off > 0; off >>= 1) {
        sK_sum += __shfl_down(sK_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    if (lane == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        partials[bid * 2] = sK_sum;
        partials[bid * 2 + 1] = l_sum;
    }
}

// one 256-thread block per point: slabs -> d_x1, d_v1 (through the x/ell scaling and the direction normalisation).  Each of the
// four waves sums every fourth slab (the slabs are 384 KB apart at C4: 32 dependent-latency loads per element on one wave was
// 21 us for 500 points), the partial sums meet in LDS.  Block 0 also folds the per-workgroup scalar partials into d_hyp (was a
// launch of its own).
constexpr int PTS_NT = 256;
// (round 6) a SECOND slab set (slab2 / nsplit2 / sym2 / partials2; null: none) is added on the way -- everything below is linear in
// the slab, so K_ZX-bar's and K_ZZ-bar's backwards share one launch -- and `tail` (scal != null) folds in the one-call step's scalar
// tail (scale_epilogue_kernel of elbo.hip: 2 vbar = 1 / (noise rows) on the point / direction gradients, the chain rule of the
// constrained hyper-parameters, the loss): d_x1 / d_v1 / d_hyp must then hold no other contribution that wants scaling later.
struct PointsTail {
    const float* scal; const float* kl0; float inv_rows, inv_num_data;
    const float *rl, *rs, *rn; float *drl, *drs, *drn, *dconst, *loss;
};
__global__ __launch_bounds__(PTS_NT) void kernel_bwd_points_kernel(const float* __restrict__ slab, int nsplit,
                                                                   const float* __restrict__ P1,
                                                                   const float* __restrict__ vnorm1, int n1, int d, int p,
                                                                   int K4, int DP, int NP, const float* __restrict__ hyp,
                                                                   float sym, float* __restrict__ d_x1,
                                                                   float* __restrict__ d_v1, const float* __restrict__ partials,
                                                                   int nblocks, float* __restrict__ d_hyp,
                                                                   const float* __restrict__ slab2 = nullptr, int nsplit2 = 0, float sym2 = 0.f,
                                                                   const float* __restrict__ partials2 = nullptr, int nblocks2 = 0,
                                                                   PointsTail tail = PointsTail{}) {
    extern __shared__ float dPs[];          // [q][DP] summed over the split slabs, then [q + 1] dots, then [waves - 1][q][DP] wave partials
    const int i = blockIdx.x, t = threadIdx.x, w = t >> 6, l = t & 63;
    const int nth = blockDim.x, nw = nth >> 6;          // 4 waves; 1 when the partials would not fit into LDS (q DP > 3072)
    const int q = p + 1, qd = q * DP;
    const int64_t n1q = (int64_t)n1 * q;
    const float ell = hyp[0];
    float* part = dPs + qd + q + 1;
    for (int e = l; e < qd; e += 64) {
        const int a = e / DP, col = e - a * DP;
        const float* src = slab + ((int64_t)i * q + a) * NP + col;
        float sum = 0.f;
        for (int sp = w; sp < nsplit; sp += nw) sum += src[(int64_t)sp * n1q * NP];
        if (slab2) {        // (both sets weighted here; `sym` below is then 1)
            const float* src2 = slab2 + ((int64_t)i * q + a) * NP + col;
            float sum2 = 0.f;
            for (int sp = w; sp < nsplit2; sp += nw) sum2 += src2[(int64_t)sp * n1q * NP];
            sum = sum * sym + sum2 * sym2;
        }
        if (w == 0) dPs[e] = sum; else part[(w - 1) * qd + e] = sum;
    }
    __syncthreads();
    if (nw > 1) {
        for (int e = t; e < qd; e += nth) {
            float sum = dPs[e];
            for (int ww = 1; ww < nw; ++ww) sum += part[(ww - 1) * qd + e];
            dPs[e] = sum;
        }
        __syncthreads();
    }
    const float* xt = P1 + (int64_t)i * q * DP;
    float* dots = dPs + qd;
    // vhat-bar_a = dP[a,:] + alphabar_a x~ ; dots[a] = vhat_a . vhat-bar_a ; alphabar_a = -dP[a,K4]
    for (int a = 1 + t; a <= p; a += nth) {
        const float* vh = P1 + ((int64_t)i * q + a) * DP;
        const float ab = -dPs[a * DP + K4];
        float dot = 0.f;
        for (int k = 0; k < d; ++k) dot += vh[k] * (dPs[a * DP + k] + ab * xt[k]);
        dots[a] = dot;
    }
    __syncthreads();
    if (slab2) sym = 1.f;
    if (tail.scal) sym *= tail.inv_rows / hyp[2];
    const float nbar = -0.5f * dPs[K4];
    for (int k = t; k < d; k += nth) {
        // x~bar = dP[0,:] + 2 nbar x~ + sum_a alphabar_a vhat_a
        float xb = dPs[k] + 2.f * nbar * xt[k];
        for (int a = 1; a <= p; ++a) xb += -dPs[a * DP + K4] * P1[((int64_t)i * q + a) * DP + k];
        d_x1[(int64_t)i * d + k] += sym * xb / ell;
    }
    for (int e = t; e < p * d; e += nth) {
        const int a = 1 + e / d, k = e - (a - 1) * d;
        const float* vh = P1 + ((int64_t)i * q + a) * DP;
        const float vb = dPs[a * DP + k] - dPs[a * DP + K4] * xt[k];
        const float inv = 1.f / vnorm1[(int64_t)i * p + (a - 1)];
        d_v1[((int64_t)i * p + (a - 1)) * d + k] += sym * (vb - vh[k] * dots[a]) * inv;   // normalisation Jacobian
    }
    if (i == 0) {       // sum(G o K) and the lengthscale partials of the tile workgroups -> d_hyp (fixed order: deterministic)
        __shared__ double r0[PTS_NT], r1[PTS_NT];
        double a = 0, b = 0;
        for (int j = t; j < nblocks; j += nth) { a += partials[2 * j]; b += partials[2 * j + 1]; }
        for (int j = t; j < nblocks2; j += nth) { a += partials2[2 * j]; b += partials2[2 * j + 1]; }
        r0[t] = a; r1[t] = b;
        for (int j = nth + t; j < PTS_NT; j += nth) { r0[j] = 0; r1[j] = 0; }
        __syncthreads();
        for (int off = PTS_NT / 2; off > 0; off >>= 1) {        // (threads past nth hold zeros)
            if (t < off) { r0[t] += r0[t + off]; r1[t] += r1[t + off]; }
            __syncthreads();
        }
        if (t == 0) {
            d_hyp[1] += (float)(r0[0] / (double)hyp[1]);   // d outputscale = sum(G o K)/s
            d_hyp[0] += (float)r1[0];                       // d lengthscale
            if (tail.scal) {                                // scale_epilogue_kernel's first thread (elbo.hip)
                const float sc = tail.inv_rows / hyp[2];
                const float* scal = tail.scal;
                const float d0 = d_hyp[0] * sc + scal[4], d1 = d_hyp[1] * sc + scal[3], d2 = d_hyp[2] + scal[1];
                d_hyp[0] = d0; d_hyp[1] = d1; d_hyp[2] = d2;
                auto sigm = [](float v) { return 1.f / (1.f + expf(-v)); };
                tail.drl[0] += d0 * sigm(tail.rl[0]);
                tail.drs[0] += d1 * sigm(tail.rs[0]);
                tail.drn[0] += d2 * sigm(tail.rn[0]);
                tail.dconst[0] += scal[2];
                tail.loss[0] = -scal[0] * tail.inv_rows + tail.kl0[0] * tail.inv_num_data;
            }
        }
    }
}

struct Geom { int q, R, T, K4, DP, NP, Rr, Tr; };
inline int make_geom(int d, int p, Geom& g) {
    g.q = p + 1;
    if (d < 1 || p < 0 || g.q > TMAX) return DSVGP_EINVAL;
    g.R = TMAX / g.q;
    g.T = g.R * g.q;
    g.Rr = g.R >= 2 ? g.R / 2 : g.R;       // row tiles hold half as many points as column tiles
    g.Tr = g.Rr * g.q;
    g.K4 = (d + 3) & ~3;
    g.DP = g.K4 + 4;
    g.NP = (g.DP + 15) & ~15;
    if (g.NP > 96) return DSVGP_EINVAL;   // d <= 88
    return 0;
}

inline int launch_points(hipStream_t st, const Geom& g, const float* slab, int ns, const float* P1, const float* vnorm1, int n1, int d, int p,
                         const float* hyp, float sym, float* d_x1, float* d_v1, const float* partials, int nparts, float* d_hyp,
                         const float* slab2 = nullptr, int ns2 = 0, float sym2 = 0.f, const float* partials2 = nullptr, int nparts2 = 0,
                         PointsTail tail = PointsTail{}) {
    const int pts_waves = (g.q * g.DP > 3072) ? 1 : PTS_NT / 64;       // (wave partials: <= 48 KB of LDS)
    hipLaunchKernelGGL(kernel_bwd_points_kernel, dim3(n1), dim3(64 * pts_waves), sizeof(float) * (pts_waves * g.q * g.DP + g.q + 1), st, slab, ns,
                       P1, vnorm1, n1, d, p, g.K4, g.DP, g.NP, hyp, sym, d_x1, d_v1, partials, nparts, d_hyp, slab2, ns2, sym2, partials2, nparts2,
                       tail);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
// the last launch of every kernel backward: queued now, or noted in the context for dsvgp_kernel_bwd_points_flush (common.h: defer_points)
inline int finish_points(dsvgp_ctx* ctx, const Geom& g, const float* slab, int ns, const float* P1, const float* vnorm1, int n1, int d, int p,
                         const float* hyp, float sym, float* d_x1, float* d_v1, const float* partials, int nparts, float* d_hyp) {
    if (ctx->defer_points && ctx->n_deferred < 2) {
        ctx->deferred[ctx->n_deferred++] = dsvgp_ctx::PointsJob{slab, ns, partials, nparts, sym};
        return 0;
    }
    return launch_points(ctx->stream, g, slab, ns, P1, vnorm1, n1, d, p, hyp, sym, d_x1, d_v1, partials, nparts, d_hyp);
}

inline bool bwd_use_pair(const Geom& g) { return (g.q == 6 || g.q == 3) && g.NP <= 64; }     // (NP <= 32: KSM = 8; <= 64: KSM = 16)
#ifdef BWD_NO_SPLIT
inline bool bwd_use_split(const Geom&) { return false; }
#else
inline bool bwd_use_split(const Geom& g) { return g.q == 11 && g.NP <= 16; }                  // (full-gradient SVGP at d <= 12: BASELINE config 3)
#endif
// row-tile height / column-tile width / workgroup budget of the backward variant that will run
inline void bwd_tiles(const Geom& g, int& tr, int& tc, int& wgs) {
    if (bwd_use_split(g)) { tr = tc = (48 / g.q) * g.q; wgs = BWDS_WGS; }
    else if (bwd_use_pair(g)) { tr = tc = (48 / g.q) * g.q; wgs = PAIR_WGS; }
    else { tr = g.Tr; tc = g.T; wgs = BWD_TARGET_WGS; }
}
inline int bwd_nsplit(int n1, int n2, const Geom& g) {
    int tr, tc, wgs;
    bwd_tiles(g, tr, tc, wgs);
    const int rt = cdiv((int64_t)n1 * g.q, tr), ctiles = cdiv((int64_t)n2 * g.q, tc);
    int ns = wgs / rt;
    if (ns < 1) ns = 1;
    if (ns > ctiles) ns = ctiles;
    return ns;
}

template <typename GT, int Q, int NCH>
inline void launch_bwd(hipStream_t st, dim3 grid, size_t lds, const GT* G, int64_t ldg, const float* P1,
                       const float* self1, int n1q, const float* P2, const float* self2, int n2q, const Geom& g,
                       int gvec, const float* hyp, float* slab, float* partials) {
    (void)hipFuncSetAttribute((const void*)kernel_bwd_kernel<GT, Q, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kernel_bwd_kernel<GT, Q, NCH>), grid, dim3(BWD_NT), lds, st, G, ldg, P1, self1, n1q, P2, self2,
                       n2q, g.q, g.Rr, g.R, g.K4, g.DP, g.NP, gvec, hyp, slab, partials);
}
template <typename GT>
inline void dispatch_bwd(hipStream_t st, dim3 grid, size_t lds, const GT* G, int64_t ldg, const float* P1,
                         const float* self1, int n1q, const float* P2, const float* self2, int n2q, const Geom& g,
                         int gvec, const float* hyp, float* slab, float* partials) {
    const int Trp = (g.Tr + 15) & ~15;
    if (Trp > 48) launch_bwd<GT, 0, BWD_GCH_MAX>(st, grid, lds, G, ldg, P1, self1, n1q, P2, self2, n2q, g, gvec, hyp, slab, partials);
    else if (g.q == 4) launch_bwd<GT, 4, BWD_GCH>(st, grid, lds, G, ldg, P1, self1, n1q, P2, self2, n2q, g, gvec, hyp, slab, partials);
#ifndef BWD_NO_Q11
    else if (g.q == 11) launch_bwd<GT, 11, BWD_GCH>(st, grid, lds, G, ldg, P1, self1, n1q, P2, self2, n2q, g, gvec, hyp, slab, partials);   // (full-gradient SVGP at d = 10: BASELINE config 3)
#endif
    else launch_bwd<GT, 0, BWD_GCH>(st, grid, lds, G, ldg, P1, self1, n1q, P2, self2, n2q, g, gvec, hyp, slab, partials);
}

// =================================================================================================
// Canonical (one-hot) directions on side 2, the same for every point of side 2 (round 6).
// The reference's callers guarantee this structure for K_ZX: at training time the data directions are the canonical vectors of the
// sampled derivative columns, tiled over the minibatch (directional_vi.py:81-88, 238), at evaluation time eye(d)[:p] (:292-294).
// With v2_b = e_{c_b} the inner products of the general formulation collapse:
//     T0b = x1~ . v2_b = x1~[c_b],   beta_b = x2~ . v2_b = x2~[c_b]      =>  w_b = x1~[c_b] - x2~[c_b]
//     Tab = v1_a . v2_b = v1_a[c_b]                                        (the same for every point of side 2)
// so only T00 = x1~ . x2~ and Ta0 = v1_a . x2~ are products: U = P1' X2'^T with ONE column per point of side 2 (not q) -- 1/q of
// the MFMA work and of the fragment traffic of T' = P1' P2'^T, and no T' round trip through LDS for (q - 1) / q of the columns.
// Backward: Tbar's direction columns contract with one-hot rows, i.e. they are COLUMN SUMS into packed column c_b (accumulated per
// lane over the sweep, reduced across the lanes of a point once per workgroup); only its value columns go through the MFMA
// (dP1 += Tbar[:, c0] X2', K = points of the tile instead of 48).
// One wave per workgroup, 48-row tiles of side 1 x 48-column tiles of the output, lane <-> point pair(s), as the pair kernels.
// q in {3, 6}, packed width <= 32 (d <= 28: BASELINE configs 2 and 4), float output / float or double upstream.
// dir_idx[p] (device memory): coordinate of direction b is dir_idx[b] - idx_base.
// =================================================================================================
constexpr int CAN_LDT = 52;            // row stride of the 48 x 48 output / upstream tile in LDS (even: 8-byte strips; 16-byte aligned rows)
#ifndef CAN_FWD_WGS
#define CAN_FWD_WGS (256 * 32)          // (probed with the result NOT in the memory-side cache: 12 / 24 / 32 / 48 / 64 per CU: 88 / 75 / 75 / 83 / 90 us at C4)
#endif
#ifndef CAN_BWD_WGS
#define CAN_BWD_WGS (256 * 8)
#endif
#ifndef CAN_BWD_DMA
#define CAN_BWD_DMA 1                  // 0: the backward kernel reads the upstream micro-blocks straight into registers (tools: the A/B)
#endif
#ifndef CAN_FWD_NT
#define CAN_FWD_NT 1                   // a result larger than the memory-side cache leaves as NON-TEMPORAL stores (ovec bit 2, set by the launcher from 192 MB up): written
                                       // once, read by the forward solve a millisecond later, whatever of it is cached by then (cold result buffer: 77 -> 69 us at C4)
#endif
#ifndef CAN_BWD_NT
#define CAN_BWD_NT 0                   // 1: the upstream tile's LDS-DMA loads with the nt bit (probe)
#endif
#ifndef CAN_FWD_STRIDED
#define CAN_FWD_STRIDED 1
#endif
#ifndef CAN_FWD_MINW
#define CAN_FWD_MINW 2                 // waves per SIMD the forward kernel's registers are budgeted for
#endif
#ifndef CAN_ABL
#define CAN_ABL 0                      // tools only (results wrong): 1 = no global stores (forward), 2 = no U product, 4 = whole cache lines only (2/3 of the bytes)
#endif

// what both kernels share: the A fragments of the wave's 48 side-1 rows (self terms folded as in the pair kernels: column K4 + 2 of P1' =
// -alpha_a for a > 0 against a 1 in the value rows of side 2, so that U[ra, j] = v1_a . x2~_j - alpha_a = -u_a), and U for one column tile
template <int Q>
struct CanTile {
    static constexpr int R = 48 / Q, PPL = (R * R + 63) / 64;
};

template <int Q>
__global__ __launch_bounds__(64, CAN_FWD_MINW) void kernel_fwd_canon_kernel(const float* __restrict__ P1, const float* __restrict__ self1, int n1q,
                                                              const float* __restrict__ P2, const float* __restrict__ self2, int n2,
                                                              int K4, int DP, const int* __restrict__ dir_idx, int idx_base, int ovec,
                                                              const float* __restrict__ hyp, float* __restrict__ out, int64_t ld) {
    constexpr int R = 48 / Q, T = 48, PPL = CanTile<Q>::PPL, NPAIR = R * R, KSM = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDX = K4 + 5;                 // row stride of the staged side-2 value rows [16][LDX]: columns K4 + 2 = 1, K4 + 3 = |x2~|^2
    float* Xs = smem;                       // [16][LDX]
    float* TT = smem + ((16 * LDX + 3) & ~3);     // [48][CAN_LDT]: U^T ([point][row], rows 0 .. R-1) first, then the output tile
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * T;
    const int n2q = n2 * Q;
    const int ncoltiles = (n2 + R - 1) / R;
    const int KS = K4 / 4 + 1;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    int cidx[Q];                            // coordinate of direction b (b >= 1)
#pragma unroll
    for (int b = 1; b < Q; ++b) cidx[b] = dir_idx[b - 1] - idx_base;
    cidx[0] = 0;

    float areg[3][KSM];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int gr = row0 + i * 16 + m16;
        const bool ok = gr < n1q;
        const int a = (i * 16 + m16) % Q;
        const float sf = ok ? self1[gr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            float v = 0.f;
            if (ks < K4 / 4) v = ok ? P1[(int64_t)gr * DP + ks * 4 + kg] : 0.f;
            else if (ks == K4 / 4 && ok) v = (kg == 2) ? (a == 0 ? 0.f : -sf) : 0.f;
            areg[i][ks] = v;
        }
    }
    // per-pair constants of side 1: |x1~|^2, x1~[c_b], v1_a[c_b]
    int pi_[PPL], pj_[PPL];
    bool prow[PPL];
    float nrm1[PPL], zc[PPL][Q], gc[PPL][Q][Q];
#pragma unroll
    for (int pp = 0; pp < PPL; ++pp) {
        const int pid = lane + 64 * pp;
        pi_[pp] = pid / R; pj_[pp] = pid - pi_[pp] * R;
        const int r0 = row0 + pi_[pp] * Q;
        prow[pp] = pid < NPAIR && r0 < n1q;
        nrm1[pp] = prow[pp] ? self1[r0] : 0.f;
#pragma unroll
        for (int b = 1; b < Q; ++b) {
            zc[pp][b] = prow[pp] ? P1[(int64_t)r0 * DP + cidx[b]] : 0.f;
#pragma unroll
            for (int a = 1; a < Q; ++a) gc[pp][a][b] = prow[pp] ? P1[(int64_t)(r0 + a) * DP + cidx[b]] : 0.f;
        }
    }
    for (int e = lane; e < 16 * LDX; e += 64) Xs[e] = 0.f;
    const int pch = DP / 4;                 // float4 per packed row
    const bool rows_full = row0 + T <= n1q;

    // Column tiles of a workgroup: blockIdx.x, + gridDim.x, ... (CAN_FWD_STRIDED = 1: at any moment the workgroups of a row tile write
    // ADJACENT 192-byte pieces of the same 48 rows -- whole cache lines and DRAM pages fill up together; in contiguous chunks per workgroup
    // the pieces written at one time lie 2 KB apart, which costs nothing while the result is still in the memory-side cache from the launch
    // before (what a probe that rewrites one buffer sees: 62 us) and a third of the rate when it is not (the step: 87 us) --
    // profiles/r06_c_canon_assembly.txt)
    const int cper = (ncoltiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int ct_step = CAN_FWD_STRIDED ? (int)gridDim.x : 1;
    const int ct_lo = CAN_FWD_STRIDED ? (int)blockIdx.x : (int)blockIdx.x * cper;
    const int ct_hi = CAN_FWD_STRIDED ? ncoltiles : min(ct_lo + cper, ncoltiles);
    // the tile's R value rows of P2: float4 number e = lane (< R pch <= 128: two per lane at most) of [R][DP]
    f4 pf[2];
    float pnrm = 0.f;
    auto prefetch = [&](int ct_) {
        // (loads without branches, from clamped addresses: with straight-line code the compiler counts the tile's stores issued behind
        //  them and waits with vmcnt(stores), not vmcnt(0) -- the stores of a tile drain under the next tile's work)
        const int j0 = ct_ * R;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = min(lane + 64 * u, R * pch - 1), r = e / pch, k = (e - r * pch) * 4;
            pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)min(j0 + r, n2 - 1) * Q * DP + k);      // (zeroed past n2 where it is used)
        }
        pnrm = self2[(int64_t)min(j0 + min(lane, R - 1), n2 - 1) * Q];
    };
    if (ct_lo >= ct_hi) return;
    prefetch(ct_lo);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the prologue's loads are in -- said once, so that no wait inside the loop is charged to them
    // the sweep in two instances: FULL = whole 48 x 48 tiles leaving through LDS as nine 16-byte stores per lane -- straight-line code, so the
    // wait for the next tile's operands is vmcnt(9 + ..) and the stores drain under the next tile -- and the general one (edge rows, the last
    // partial column tile, outputs that are not 16-byte aligned)
    auto sweep = [&](auto fullc, int lo, int hi) {
    constexpr bool FULL = decltype(fullc)::value;
    for (int ct = lo; ct < hi; ct += ct_step) {
        const int j0 = ct * R, col0 = j0 * Q;
        WAVE_SYNC();                    // (single wave: the previous tile's LDS reads are behind us)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u, r = e / pch, k = (e - r * pch) * 4;
            if (e < R * pch) {
                const bool in = j0 + r < n2;
#pragma unroll
                for (int t = 0; t < 4; ++t) Xs[r * LDX + k + t] = in ? pf[u][t] : 0.f;
            }
        }
        WAVE_SYNC();                    // (the extension columns go on top of the packed row's own columns K4 ..)
        if (lane < R) {
            const bool in = j0 + lane < n2;
            Xs[lane * LDX + K4 + 1] = 0.f;
            Xs[lane * LDX + K4 + 2] = in ? 1.f : 0.f;
            Xs[lane * LDX + K4 + 3] = in ? pnrm : 0.f;
        }
        prefetch(min(ct + ct_step, ncoltiles - 1)); // (unconditional: one harmless reload at the end of the range)
        WAVE_SYNC();
        // U = P1' X2'^T: 3 row tiles x (one 16-column tile of which R columns are points), K = K4 + 4
        f4 t[3];
        {
            const float* pb = Xs + m16 * LDX + kg;
            auto product = [&](auto ksc) {
                constexpr int KS_ = decltype(ksc)::value;
#pragma unroll
                for (int ks = 0; ks < KS_; ++ks) {
                    const float bv = pb[ks * 4];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][ks], bv, ks == 0 ? f4{0.f, 0.f, 0.f, 0.f} : t[i], 0, 0, 0);
                }
            };
            switch ((CAN_ABL & 2) ? 1 : KS) {
                case 1: product(std::integral_constant<int, 1>{}); break;
                case 2: product(std::integral_constant<int, 2>{}); break;
                case 3: product(std::integral_constant<int, 3>{}); break;
                case 4: product(std::integral_constant<int, 4>{}); break;
                case 5: product(std::integral_constant<int, 5>{}); break;
                case 6: product(std::integral_constant<int, 6>{}); break;
                case 7: product(std::integral_constant<int, 7>{}); break;
                default: product(std::integral_constant<int, 8>{}); break;
            }
        }
        // U^T -> LDS: lane (point m16, kg) holds rows 16 i + 4 kg .. + 3 of column m16: four consecutive floats of row m16 of [point][row]
        if (m16 < R) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f4*>(TT + m16 * CAN_LDT + i * 16 + kg * 4) = t[i];
        }
        WAVE_SYNC();
        float v[PPL][Q][Q];
        bool mine[PPL];
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            float uq[Q], xc[Q];
            const float* up = TT + pj_[pp] * CAN_LDT + pi_[pp] * Q;
            const float* xp = Xs + pj_[pp] * LDX;
#pragma unroll
            for (int a = 0; a < Q; ++a) uq[a] = up[a];
#pragma unroll
            for (int b = 1; b < Q; ++b) xc[b] = xp[cidx[b]];
            const float nrm2 = xp[K4 + 3];
            mine[pp] = prow[pp] && j0 + pj_[pp] < n2;
            const float nn = fmaxf(nrm1[pp] + nrm2 - 2.f * uq[0], 0.f);     // covar_dist clamps at 0
            const float k = s * expf(-0.5f * nn);                             // postprocess_rbf, ScaleKernel
            const float kil = k * il, kil2 = k * il2;
            v[pp][0][0] = k;
#pragma unroll
            for (int b = 1; b < Q; ++b) v[pp][0][b] = (zc[pp][b] - xc[b]) * kil;                                  // w_b k / ell
#pragma unroll
            for (int a = 1; a < Q; ++a) {
                v[pp][a][0] = uq[a] * kil;                                                                          // -u_a k / ell
#pragma unroll
                for (int b = 1; b < Q; ++b) v[pp][a][b] = (gc[pp][a][b] + uq[a] * (zc[pp][b] - xc[b])) * kil2;      // (G_ab - u_a w_b) k / ell^2
            }
        }
        WAVE_SYNC();                    // every lane has read its U values: the output tile may overlay them
        if constexpr (FULL) {
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) {
                if (lane + 64 * pp < NPAIR) {
                    float* blk = TT + pi_[pp] * Q * CAN_LDT + pj_[pp] * Q;
#pragma unroll
                    for (int a = 0; a < Q; ++a) {
                        if constexpr (Q % 2 == 0) {
                            using F2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int b = 0; b < Q; b += 2) *reinterpret_cast<F2*>(blk + a * CAN_LDT + b) = F2{v[pp][a][b], v[pp][a][b + 1]};
                        } else {
#pragma unroll
                            for (int b = 0; b < Q; ++b) blk[a * CAN_LDT + b] = v[pp][a][b];
                        }
                    }
                }
            }
            WAVE_SYNC();
            float* orow = out + (int64_t)row0 * ld + col0;
#pragma unroll
            for (int u = 0; u < 9; ++u) {   // the tile leaves as nine fully coalesced 16-byte store instructions (rows of 192 bytes)
                const int id = lane + 64 * u, r = id / 12, c4 = (id - 12 * r) * 4;
                const f4 x = *reinterpret_cast<const f4*>(TT + r * CAN_LDT + c4);
                if ((CAN_ABL & 4) && ((ct & 1) ? c4 < 16 : c4 >= 32)) continue;   // (tools: only the whole 128-byte line of each row piece)
                if (CAN_ABL & 1) continue;
                if (CAN_FWD_NT && (ovec & 4)) __builtin_nontemporal_store(x, reinterpret_cast<f4*>(orow + (int64_t)r * ld + c4));
                else *reinterpret_cast<f4*>(orow + (int64_t)r * ld + c4) = x;
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) {
                if (!mine[pp]) continue;
                float* o = out + (int64_t)(row0 + pi_[pp] * Q) * ld + col0 + pj_[pp] * Q;
#pragma unroll
                for (int a = 0; a < Q; ++a)
#pragma unroll
                    for (int b = 0; b < Q; ++b) o[a * ld + b] = v[pp][a][b];
            }
        }
    }
    };
    const int nfull = (rows_full && (ovec & 2)) ? min(ct_hi, n2q / T) : ct_lo;    // this workgroup's tiles below nfull are whole
    sweep(std::true_type{}, ct_lo, nfull);
    const int g0 = nfull > ct_lo ? ct_lo + (nfull - ct_lo + ct_step - 1) / ct_step * ct_step : ct_lo;     // its first tile that is not
    sweep(std::false_type{}, g0, ct_hi);
}

// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4), as gemm32.hip's: an asm statement, so that WE count it (s_waitcnt vmcnt) and the
// compiler does not park a vmcnt(0) in front of every later LDS read; M0 (the LDS destination base; lane l lands at base + 16 l) is saved,
// set and restored inside the statement.
__device__ __forceinline__ void can_dma16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
#if CAN_BWD_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
#endif
}

// backward: upstream micro-blocks HBM -> registers, the pair kernel's transform with (w_b, -u_a, G_ab) from the canonical sources;
// Tbar's value columns -> LDS as the A operand of dP1 += Tbar[:, c0] X2' (K = the R points of the tile), its direction columns ->
// per-lane column sums S[a][b]; at the end of the sweep S is added across the lanes of a point and lands in packed column c_b.
// DMA (float upstream, 16-byte aligned rows): the upstream tile of the NEXT column tile comes in by LDS-DMA as nine fully coalesced 1 KB
// pieces ([48][48] floats, dense) while this tile is worked on -- no registers held for it, no 8-byte strided loads; a last, partial
// column tile goes through the register path.
template <typename GT, int Q, bool DMA>
__global__ __launch_bounds__(64, 2) void kernel_bwd_canon_kernel(const GT* __restrict__ G, int64_t ldg, const float* __restrict__ P1,
                                                                 const float* __restrict__ self1, int n1q, const float* __restrict__ P2,
                                                                 const float* __restrict__ self2, int n2, int K4, int DP, int NP,
                                                                 const int* __restrict__ dir_idx, int idx_base, int gvec,
                                                                 const float* __restrict__ hyp, float* __restrict__ slab,
                                                                 float* __restrict__ partials) {
    constexpr int R = 48 / Q, T = 48, PPL = CanTile<Q>::PPL, NPAIR = R * R, KSM = 8, KP = (R + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDX = NP + 1;                 // row stride of the staged side-2 value rows [16][LDX] (B operand of the dP1 product: NP columns)
    float* Xs = smem;                       // [16][LDX]
    float* TT = smem + ((16 * LDX + 3) & ~3);     // [16][CAN_LDT]: U^T ([point][row]), then Tbar's value columns as [point][row] (the A operand, read transposed)
    float* CS = TT + 16 * CAN_LDT;          // [48][8]: column sums at the end of the sweep
    float* Gs = CS + 48 * 8;                // DMA: [48][48] upstream tile
    const int lane = threadIdx.x, m16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.y * T;
    const int n2q = n2 * Q;
    const int ncoltiles = (n2 + R - 1) / R;
    const int nnp = NP / 16;                // 1 or 2
    const int KS = K4 / 4 + 1;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    const bool vec = gvec != 0;
    int cidx[Q];
#pragma unroll
    for (int b = 1; b < Q; ++b) cidx[b] = dir_idx[b - 1] - idx_base;
    cidx[0] = 0;

    float areg[3][KSM];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int gr = row0 + i * 16 + m16;
        const bool ok = gr < n1q;
        const int a = (i * 16 + m16) % Q;
        const float sf = ok ? self1[gr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            float v = 0.f;
            if (ks < K4 / 4) v = ok ? P1[(int64_t)gr * DP + ks * 4 + kg] : 0.f;
            else if (ks == K4 / 4 && ok) v = (kg == 2) ? (a == 0 ? 0.f : -sf) : 0.f;
            areg[i][ks] = v;
        }
    }
    int pi_[PPL], pj_[PPL];
    bool prow[PPL];
    float nrm1[PPL], zc[PPL][Q], gc[PPL][Q][Q], S[PPL][Q][Q];
#pragma unroll
    for (int pp = 0; pp < PPL; ++pp) {
        const int pid = lane + 64 * pp;
        pi_[pp] = pid / R; pj_[pp] = pid - pi_[pp] * R;
        const int r0 = row0 + pi_[pp] * Q;
        prow[pp] = pid < NPAIR && r0 < n1q;
        nrm1[pp] = prow[pp] ? self1[r0] : 0.f;
#pragma unroll
        for (int a = 0; a < Q; ++a)
#pragma unroll
            for (int b = 0; b < Q; ++b) S[pp][a][b] = 0.f;
#pragma unroll
        for (int b = 1; b < Q; ++b) {
            zc[pp][b] = prow[pp] ? P1[(int64_t)r0 * DP + cidx[b]] : 0.f;
#pragma unroll
            for (int a = 1; a < Q; ++a) gc[pp][a][b] = prow[pp] ? P1[(int64_t)(r0 + a) * DP + cidx[b]] : 0.f;
        }
    }
    for (int e = lane; e < 16 * LDX; e += 64) Xs[e] = 0.f;
    for (int e = lane; e < 16 * CAN_LDT; e += 64) TT[e] = 0.f;
    const int pch = DP / 4;
    f4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float sK_sum = 0.f, l_acc = 0.f;
    // the tile's R value rows of P2, one tile ahead, in registers (clamped addresses, no branches; zeroed past n2 where they are used)
    f4 pf[2];
    float pnrm = 0.f;
    auto prefetch = [&](int ct_) {
        const int j0 = ct_ * R;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = min(lane + 64 * u, R * pch - 1), r = e / pch, k = (e - r * pch) * 4;
            pf[u] = *reinterpret_cast<const f4*>(P2 + (int64_t)min(j0 + r, n2 - 1) * Q * DP + k);
        }
        pnrm = self2[(int64_t)min(j0 + min(lane, R - 1), n2 - 1) * Q];
    };
    const unsigned gs_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Gs;
    auto dma_tile = [&](int ct_) {           // (whole column tiles only)
        if constexpr (DMA) {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int id = lane + 64 * u, r = id / 12, c4 = (id - 12 * r) * 4;
                can_dma16(reinterpret_cast<const float*>(G) + (int64_t)min(row0 + r, n1q - 1) * ldg + ct_ * T + c4, gs_addr + u * 1024);
            }
        }
    };
    if ((int)blockIdx.x >= ncoltiles) return;       // (never: the launcher's grid is at most the number of column tiles)
    prefetch(blockIdx.x);
    if (DMA && (int)blockIdx.x * T + T <= n2q) dma_tile(blockIdx.x);

    for (int ct = blockIdx.x; ct < ncoltiles; ct += gridDim.x) {
        const int j0 = ct * R, col0 = j0 * Q;
        float g[PPL][Q][Q];
        bool mine[PPL];
        const bool from_lds = DMA && col0 + T <= n2q;
        if (from_lds) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this tile's DMA (and P2 rows) are in
            asm volatile("" ::: "memory");
            using V2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) {
                mine[pp] = prow[pp] && j0 + pj_[pp] < n2;
                const float* src = Gs + (pi_[pp] * Q) * T + pj_[pp] * Q;
#pragma unroll
                for (int a = 0; a < Q; ++a) {
                    if constexpr (Q % 2 == 0) {
#pragma unroll
                        for (int b = 0; b < Q; b += 2) {
                            const V2 x = *reinterpret_cast<const V2*>(src + a * T + b);
                            g[pp][a][b] = mine[pp] ? x[0] : 0.f; g[pp][a][b + 1] = mine[pp] ? x[1] : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < Q; ++b) g[pp][a][b] = (mine[pp] && lane + 64 * pp < NPAIR) ? src[a * T + b] : 0.f;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done: the buffer may take the next tile
        } else {
        // upstream micro-blocks straight into registers (consumed after the U product)
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            mine[pp] = prow[pp] && j0 + pj_[pp] < n2;
            const GT* src = G + (int64_t)(row0 + pi_[pp] * Q) * ldg + col0 + pj_[pp] * Q;
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                if constexpr (Q % 2 == 0) {
                    if (mine[pp] && vec) {
                        using V2 = GT __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int b = 0; b < Q; b += 2) {
                            const V2 x = *reinterpret_cast<const V2*>(src + a * ldg + b);
                            g[pp][a][b] = (float)x[0]; g[pp][a][b + 1] = (float)x[1];
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int b = 0; b < Q; ++b) g[pp][a][b] = mine[pp] ? (float)src[a * ldg + b] : 0.f;
            }
        }
        }
        WAVE_SYNC();
        // the tile's R value rows of P2 (packed columns 0 .. DP-1; column K4 is the indicator: row sums of Tbar's value columns)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u, r = e / pch, k = (e - r * pch) * 4;
            if (e < R * pch) {
                const bool in = j0 + r < n2;
#pragma unroll
                for (int t = 0; t < 4; ++t) Xs[r * LDX + k + t] = in ? pf[u][t] : 0.f;
            }
        }
        WAVE_SYNC();
        if (lane < R) {
            const bool ok = j0 + lane < n2;
            Xs[lane * LDX + K4 + 1] = 0.f;
            Xs[lane * LDX + K4 + 2] = ok ? 1.f : 0.f;
            Xs[lane * LDX + K4 + 3] = ok ? pnrm : 0.f;
        }
        {   // the next tile of this workgroup: its P2 rows into registers, its upstream tile into LDS, under the work on this one
            const int ctn = ct + (int)gridDim.x;
            prefetch(min(ctn, ncoltiles - 1));
            if (DMA && ctn < ncoltiles && ctn * T + T <= n2q) dma_tile(ctn);
        }
        WAVE_SYNC();
        f4 t[3];
        {
            const float* pb = Xs + m16 * LDX + kg;
            auto product = [&](auto ksc) {
                constexpr int KS_ = decltype(ksc)::value;
#pragma unroll
                for (int ks = 0; ks < KS_; ++ks) {
                    const float bv = pb[ks * 4];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][ks], bv, ks == 0 ? f4{0.f, 0.f, 0.f, 0.f} : t[i], 0, 0, 0);
                }
            };
            switch (KS) {
                case 1: product(std::integral_constant<int, 1>{}); break;
                case 2: product(std::integral_constant<int, 2>{}); break;
                case 3: product(std::integral_constant<int, 3>{}); break;
                case 4: product(std::integral_constant<int, 4>{}); break;
                case 5: product(std::integral_constant<int, 5>{}); break;
                case 6: product(std::integral_constant<int, 6>{}); break;
                case 7: product(std::integral_constant<int, 7>{}); break;
                default: product(std::integral_constant<int, 8>{}); break;
            }
        }
        if (m16 < R) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f4*>(TT + m16 * CAN_LDT + i * 16 + kg * 4) = t[i];
        }
        WAVE_SYNC();
        float tv[PPL][Q];                   // Tbar's value column of this pair: Tbar[(i, a)][(j, 0)]
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            float uq[Q], w[Q];
            const float* up = TT + pj_[pp] * CAN_LDT + pi_[pp] * Q;
            const float* xp = Xs + pj_[pp] * LDX;
#pragma unroll
            for (int a = 0; a < Q; ++a) uq[a] = up[a];
#pragma unroll
            for (int b = 1; b < Q; ++b) w[b] = zc[pp][b] - xp[cidx[b]];
            const float nrm2 = xp[K4 + 3];
            const float nn = fmaxf(nrm1[pp] + nrm2 - 2.f * uq[0], 0.f);
            const float k = mine[pp] ? s * expf(-0.5f * nn) : 0.f;
            const float kil = k * il, kil2 = k * il2;
            float first = 0.f, second = 0.f, hsum = 0.f, dots = 0.f;
            float gu[Q];
#pragma unroll
            for (int b = 0; b < Q; ++b) gu[b] = 0.f;
#pragma unroll
            for (int b = 1; b < Q; ++b) first = __builtin_fmaf(g[pp][0][b], w[b], first);                 // sum g0b w_b
#pragma unroll
            for (int a = 1; a < Q; ++a) {
                const float u = -uq[a];
                const float ga0 = g[pp][a][0];
                float gw = 0.f, gt = 0.f;
#pragma unroll
                for (int b = 1; b < Q; ++b) {
                    const float gab = g[pp][a][b];
                    gw = __builtin_fmaf(gab, w[b], gw);
                    gt = __builtin_fmaf(gab, gc[pp][a][b], gt);
                    gu[b] = __builtin_fmaf(gab, u, gu[b]);
                    S[pp][a][b] = __builtin_fmaf(kil2, gab, S[pp][a][b]);                               // Tbar_ab, summed over the points of side 2
                }
                second = __builtin_fmaf(ga0, u, second);
                hsum += gt - u * gw;
                const float ubar = -(kil * ga0 + kil2 * gw);
                tv[pp][a] = -ubar;                                                                      // Tbar_a0
                dots = __builtin_fmaf(ubar, u, dots);
            }
#pragma unroll
            for (int b = 1; b < Q; ++b) {
                const float wbar = kil * g[pp][0][b] - kil2 * gu[b];
                S[pp][0][b] += wbar;                                                                    // Tbar_0b
                dots = __builtin_fmaf(wbar, w[b], dots);
            }
            const float e1 = il * (first - second), e2 = il2 * hsum;
            const float t00 = k * (g[pp][0][0] + e1 + e2);                                              // Tbar_00
            tv[pp][0] = t00;
            sK_sum += t00;
            l_acc += k * (e1 + 2.f * e2) - t00 * nn + dots;
        }
        WAVE_SYNC();                    // every lane has read its U values: Tbar's value columns take their place, [point][row]
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            if (lane + 64 * pp < NPAIR) {
                float* tp = TT + pj_[pp] * CAN_LDT + pi_[pp] * Q;
#pragma unroll
                for (int a = 0; a < Q; ++a) tp[a] = tv[pp][a];
            }
        }
        WAVE_SYNC();
        // dP1[48, NP] += Tbar[:, value columns][48, R] . X2'[R, NP]:  A operand lane (m, k) = Tbar[row m][point k] = TT[k][m]
        {
#pragma unroll
            for (int kk = 0; kk < 4 * KP; kk += 4) {
                float av[3], bv[2];
#pragma unroll
                for (int i = 0; i < 3; ++i) av[i] = TT[(kk + kg) * CAN_LDT + i * 16 + m16];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = (j < nnp) ? Xs[(kk + kg) * LDX + 16 * j + m16] : 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (j < nnp) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // ---- the column sums of Tbar's direction columns: across the R lanes of a point (pj = pid % R: xor steps below R), then into
    // packed column c_b of the point's rows through a small LDS table
    WAVE_SYNC();
#pragma unroll
    for (int pp = 0; pp < PPL; ++pp)
#pragma unroll
        for (int a = 0; a < Q; ++a)
#pragma unroll
            for (int b = 1; b < Q; ++b) {
                float x = S[pp][a][b];
#pragma unroll
                for (int off = 1; off < R; off <<= 1) x += __shfl_xor(x, off);
                S[pp][a][b] = x;
            }
#pragma unroll
    for (int pp = 0; pp < PPL; ++pp) {
        if (lane + 64 * pp < NPAIR && pj_[pp] == 0) {
#pragma unroll
            for (int a = 0; a < Q; ++a)
#pragma unroll
                for (int b = 1; b < Q; ++b) CS[(pi_[pp] * Q + a) * 8 + b] = S[pp][a][b];
        }
    }
    WAVE_SYNC();
    float* myslab = slab + ((int64_t)blockIdx.x * n1q) * NP;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j < nnp) {
                const int col = j * 16 + m16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = i * 16 + kg * 4 + r;
                    const int64_t gr = row0 + rr;
                    float x = acc[i][j][r];
#pragma unroll
                    for (int b = 1; b < Q; ++b) x += (cidx[b] == col) ? CS[rr * 8 + b] : 0.f;
                    if (gr < n1q) myslab[gr * NP + col] = x;
                }
            }
    float l_sum = -il * l_acc;
    for (int off = 32; off > 0; off >>= 1) {
        sK_sum += __shfl_down(sK_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    if (lane == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        partials[bid * 2] = sK_sum;
        partials[bid * 2 + 1] = l_sum;
    }
}

inline bool canon_ok(const Geom& g) { return (g.q == 6 || g.q == 3) && g.NP <= 32; }

}  // namespace

extern "C" int dsvgp_kernel_canon_supported(int d, int p) {
    Geom g;
    if (p < 1 || make_geom(d, p, g)) return 0;
    return canon_ok(g) ? 1 : 0;
}

// K_ZX with canonical (one-hot) directions on side 2, shared by all its points: see the block comment above kernel_fwd_canon_kernel.
// P2 / self2: the packed rows of side 2 as dsvgp_pack_points leaves them (only the value rows are read).  DSVGP_EINVAL for a geometry the
// canonical kernels do not take (q not in {3, 6} or packed width > 32): the caller uses dsvgp_kernel_fwd.
extern "C" int dsvgp_kernel_fwd_canon(dsvgp_ctx* ctx, const float* P1, const float* self1, int n1, const float* P2, const float* self2,
                                      int n2, int d, int p, const int* dir_idx, int idx_base, const float* hyp, float* out, int64_t ld) {
    if (!ctx || !P1 || !self1 || !P2 || !self2 || !hyp || !out || !dir_idx || n1 < 0 || n2 < 0) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (!canon_ok(g)) return DSVGP_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    const int n1q = n1 * g.q, n2q = n2 * g.q, R = 48 / g.q;
    if (ld < n2q) return DSVGP_EINVAL;
    const int rt = cdiv(n1q, 48), ctiles = cdiv(n2, R);
    int ns = CAN_FWD_WGS / rt;
    if (ns < 1) ns = 1;
    if (ns > ctiles) ns = ctiles;
    const size_t lds = sizeof(float) * (((16 * (size_t)(g.K4 + 5) + 3) & ~(size_t)3) + 48 * (size_t)CAN_LDT);
    const int ovec = ((ld % 4 == 0 && (uintptr_t)out % 16 == 0) ? 2 : 0) |
                     ((size_t)n1q * (size_t)n2 * g.q * sizeof(float) >= ((size_t)192 << 20) ? 4 : 0);      // (bit 2: non-temporal stores, CAN_FWD_NT)
    dim3 grid(ns, rt);
    if (g.q == 6)
        hipLaunchKernelGGL((kernel_fwd_canon_kernel<6>), grid, dim3(64), lds, ctx->stream, P1, self1, n1q, P2, self2, n2, g.K4, g.DP, dir_idx,
                           idx_base, ovec, hyp, out, ld);
    else
        hipLaunchKernelGGL((kernel_fwd_canon_kernel<3>), grid, dim3(64), lds, ctx->stream, P1, self1, n1q, P2, self2, n2, g.K4, g.DP, dir_idx,
                           idx_base, ovec, hyp, out, ld);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// backward of dsvgp_kernel_fwd_canon w.r.t. (x1, v1, lengthscale, outputscale): accumulates (+=) like dsvgp_kernel_bwd with symmetric = 0;
// workspace: dsvgp_kernel_bwd_workspace_bytes(n1, n2, d, p) bytes.
extern "C" int dsvgp_kernel_bwd_canon(dsvgp_ctx* ctx, const void* G, int64_t ldg, int g_is_double, const float* P1, const float* self1,
                                      const float* vnorm1, int n1, const float* P2, const float* self2, int n2, int d, int p,
                                      const int* dir_idx, int idx_base, const float* hyp, float* d_x1, float* d_v1, float* d_hyp,
                                      void* workspace) {
    if (!ctx || !G || !P1 || !self1 || !P2 || !self2 || !hyp || !d_x1 || !d_hyp || !workspace || !dir_idx || p < 1 || !vnorm1 || !d_v1)
        return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (!canon_ok(g)) return DSVGP_EINVAL;
    if (n1 <= 0 || n2 <= 0) return 0;
    const int n1q = n1 * g.q, n2q = n2 * g.q, R = 48 / g.q;
    if (ldg < n2q) return DSVGP_EINVAL;
    const int rt = cdiv(n1q, 48), ctiles = cdiv(n2, R);
    int ns = bwd_nsplit(n1, n2, g);               // (the slab count the workspace was sized for: the pair kernels' tiling)
    if (ns > ctiles) ns = ctiles;
    float* slab = (float*)workspace;
    float* partials = slab + (size_t)bwd_nsplit(n1, n2, g) * n1q * g.NP;
    const int esz = g_is_double ? 8 : 4;
    const int gvec = (ldg % 2 == 0) && ((uintptr_t)G % (2 * esz) == 0);
    const bool dma = CAN_BWD_DMA && !g_is_double && ldg % 4 == 0 && (uintptr_t)G % 16 == 0;       // (16-byte pieces of the upstream rows)
    const size_t lds = sizeof(float) * (((16 * (size_t)(g.NP + 1) + 3) & ~(size_t)3) + 16 * (size_t)CAN_LDT + 48 * 8 + (dma ? 48 * 48 : 0));
    dim3 grid(ns, rt);
#define DSVGP_CANON_BWD(GT_, Q_, DMA_)                                                                                                  \
    hipLaunchKernelGGL((kernel_bwd_canon_kernel<GT_, Q_, DMA_>), grid, dim3(64), lds, ctx->stream, (const GT_*)G, ldg, P1, self1, n1q, P2, \
                       self2, n2, g.K4, g.DP, g.NP, dir_idx, idx_base, gvec, hyp, slab, partials)
    if (g_is_double) { if (g.q == 6) DSVGP_CANON_BWD(double, 6, false); else DSVGP_CANON_BWD(double, 3, false); }
    else if (dma) { if (g.q == 6) DSVGP_CANON_BWD(float, 6, true); else DSVGP_CANON_BWD(float, 3, true); }
    else { if (g.q == 6) DSVGP_CANON_BWD(float, 6, false); else DSVGP_CANON_BWD(float, 3, false); }
#undef DSVGP_CANON_BWD
    DSVGP_LAUNCH_CHECK();
    return finish_points(ctx, g, slab, ns, P1, vnorm1, n1, d, p, hyp, 1.f, d_x1, d_v1, (const float*)partials, ns * rt, d_hyp);
}


// =================================================================================================
// One-hot directions on BOTH sides, the same index list for every point of both (round 6): the full-gradient SVGP (reference
// GradVariationalStrategy.py:89-99: RBFKernelGrad over cat([Z, x]) = the directional kernel with p = d and I_d at every inducing and data
// point; BASELINE config 3), or any model whose inducing directions are FIXED to the canonical columns of its data.
// With v1_a = e_{c_a} and v2_b = e_{c_b}:  u_a = delta[c_a], w_b = delta[c_b], G_ab = [a == b], delta = x1~ - x2~ -- every block is an
// elementwise function of (k, delta_c), no products:
//     K00 = k;   K0b = k delta_b / ell;   Ka0 = -k delta_a / ell;   Kab = k ([a == b] - delta_a delta_b) / ell^2;   k = s exp(-|delta|^2 / 2)
// One wave per workgroup, ONE PAIR OF POINTS PER LANE: tile = 4 points of side 1 x 16 (float) / 8 (double upstream) points of side 2.  The
// tile passes through LDS ([4 Q][16 Q] floats, dense: 256 Q^2 bytes) so that it leaves (forward) as whole 16-byte pieces of its rows and
// arrives (backward) by LDS-DMA, one tile ahead; a lane's accesses to its Q x Q block are bank-conflict free for Q = 11 (11 j + 16 i mod 64
// are 64 different banks).
// Backward: with Gm = G : K / k = G00 + (r0 - c0) / ell + (tr - quad) / ell^2  (r0 = sum_b G0b delta_b, c0 = sum_a Ga0 delta_a,
// tr = sum_a Gaa, quad = delta^T G' delta) the gradient w.r.t. the difference vector is
//     d delta_m = -k Gm delta_m  (all d coordinates, through k)  +  [m = c_c]  k ((G0c - Gc0) / ell - ((G' delta)_c + (G'^T delta)_c) / ell^2)
// The first part is the general kernels' T00-bar mechanism (w_ij = k Gm: dP1[value row] += sum_j w_ij x2~_j, its indicator column collects
// sum_j w_ij; kernel_bwd_points_kernel turns that into -sum_j w_ij x1~); the second lands in packed column c_c of the value row.  The
// direction rows of the slab are written as zeros: the directions are not parameters here (d_v1 += 0).
// Instantiated for Q = 11 (d = p = 10 <= 12); other geometries return DSVGP_EINVAL and the caller uses the general kernels.
// =================================================================================================
template <int Q, typename OT>
__global__ __launch_bounds__(64) void kernel_fwd_canon2_kernel(const float* __restrict__ P1, int n1, const float* __restrict__ P2, int n2,
                                                               int d, int DP, const int* __restrict__ dir_idx, int idx_base,
                                                               const float* __restrict__ hyp, float jitter, OT* __restrict__ out,
                                                               int64_t ld, int ovec) {
    constexpr int RP = 4, CP = 16, W = CP * Q, H = RP * Q, CPR = W / 4, NCH = H * CPR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ts = smem;                       // [H][W]
    float* X1s = Ts + H * W;                // [RP][DP] value rows of the tile's points of side 1
    float* X2s = X1s + RP * DP;             // [CP][DP]
    const int lane = threadIdx.x, i = lane >> 4, j = lane & 15;
    const int p0 = blockIdx.y * RP, j0 = blockIdx.x * CP;
    // ovec bit 3 (ctx->fwd_lower_only, K_ZZ of the one-call steps): tiles entirely to the right of the 64 x 64 blocks on the block diagonal
    // of these rows are not read by the Cholesky factorisation
    if ((ovec & 8) && (int64_t)j0 * Q > 64 * (((int64_t)p0 * Q + H - 1) / 64) + 63) return;
    for (int e = lane; e < RP * DP; e += 64) {
        const int r = e / DP, k = e - r * DP;
        X1s[e] = (p0 + r < n1) ? P1[(int64_t)(p0 + r) * Q * DP + k] : 0.f;
    }
    for (int e = lane; e < CP * DP; e += 64) {
        const int r = e / DP, k = e - r * DP;
        X2s[e] = (j0 + r < n2) ? P2[(int64_t)(j0 + r) * Q * DP + k] : 0.f;
    }
    WAVE_SYNC();
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    float dl[Q];
    dl[0] = 0.f;
#pragma unroll
    for (int b = 1; b < Q; ++b) {
        const int c = dir_idx[b - 1] - idx_base;
        dl[b] = X1s[i * DP + c] - X2s[j * DP + c];
    }
    float nn = 0.f;
    for (int k = 0; k < d; ++k) {
        const float t = X1s[i * DP + k] - X2s[j * DP + k];
        nn = __builtin_fmaf(t, t, nn);
    }
    const bool valid = p0 + i < n1 && j0 + j < n2;
    const float kv = valid ? s * expf(-0.5f * nn) : 0.f;                // postprocess_rbf, ScaleKernel
    const float kil = kv * il, kil2 = kv * il2;
    const float jd = (valid && p0 + i == j0 + j) ? jitter : 0.f;        // the global diagonal
    float* blk = Ts + (i * Q) * W + j * Q;
#pragma unroll
    for (int a = 0; a < Q; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            float v;
            if (a == 0) v = (b == 0) ? kv : dl[b] * kil;
            else if (b == 0) v = -dl[a] * kil;
            else v = ((a == b) ? kil2 : 0.f) - dl[a] * dl[b] * kil2;
            if (a == b) v += jd;
            blk[a * W + b] = v;
        }
    WAVE_SYNC();
    const int64_t row0 = (int64_t)p0 * Q, col0 = (int64_t)j0 * Q;
    const int64_t n1q = (int64_t)n1 * Q, n2q = (int64_t)n2 * Q;
    const bool full = (ovec & 1) && row0 + H <= n1q && col0 + W <= n2q;
    OT* obase = out + row0 * ld + col0;
    for (int id = lane; id < NCH; id += 64) {
        const int r = id / CPR, c4 = (id - r * CPR) * 4;
        const f4 x = *reinterpret_cast<const f4*>(Ts + r * W + c4);
        OT* o = obase + (int64_t)r * ld + c4;
        if (full) {
            if constexpr (sizeof(OT) == 4) *reinterpret_cast<f4*>(o) = x;
            else {
                using d2 = double __attribute__((ext_vector_type(2)));
                *reinterpret_cast<d2*>(o) = d2{(double)x[0], (double)x[1]};
                *reinterpret_cast<d2*>(o + 2) = d2{(double)x[2], (double)x[3]};
            }
        } else if (row0 + r < n1q) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (col0 + c4 + t < n2q) o[t] = (OT)x[t];
        }
    }
}

template <int Q, typename GT>
__global__ __launch_bounds__(64) void kernel_bwd_canon2_kernel(const GT* __restrict__ G, int64_t ldg, const float* __restrict__ P1, int n1,
                                                               const float* __restrict__ P2, int n2, int d, int K4, int DP, int NP,
                                                               const int* __restrict__ dir_idx, int idx_base,
                                                               const float* __restrict__ hyp, float* __restrict__ slab,
                                                               float* __restrict__ partials) {
    constexpr int RP = 4, CP = 64 / (int)sizeof(GT), W = CP * Q, H = RP * Q, EPC = 16 / (int)sizeof(GT), CPR = W / EPC, NCH = H * CPR;
    constexpr int NDMA = (NCH + 63) / 64;   // 1 KB pieces of the tile (the last one partly past it: the buffer is that long)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GT* Ts = reinterpret_cast<GT*>(smem);   // [H][W], dense
    float* X1s = smem + NDMA * 256;         // [RP][DP]
    float* X2s = X1s + RP * DP;             // [CP][DP]
    float* Wt = X2s + CP * DP;              // [RP][CP] w_ij of the tile
    float* rowbuf = Wt + RP * CP;           // [RP][NP] the value rows of the slab at the end
    const int lane = threadIdx.x, i = lane / CP, j = lane - i * CP;
    const bool lane_on = i < RP;            // (double upstream: 32 pairs per tile)
    const int ii = lane_on ? i : 0;
    const int p0 = blockIdx.y * RP;
    const int64_t n1q = (int64_t)n1 * Q, n2q = (int64_t)n2 * Q;
    const int ncoltiles = (n2 + CP - 1) / CP;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    int cidx[Q];
#pragma unroll
    for (int b = 1; b < Q; ++b) cidx[b] = dir_idx[b - 1] - idx_base;
    cidx[0] = 0;
    for (int e = lane; e < RP * DP; e += 64) {
        const int r = e / DP, k = e - r * DP;
        X1s[e] = (p0 + r < n1) ? P1[(int64_t)(p0 + r) * Q * DP + k] : 0.f;
    }
    const unsigned ts_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Ts;
    const int64_t ldg_last = (ldg / EPC) * EPC - EPC;       // the last whole 16-byte piece of a row
    auto dma_tile = [&](int ct_) {
        const int64_t c0 = (int64_t)ct_ * W;
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int id = min(lane + 64 * u, NCH - 1), r = id / CPR, ch = id - r * CPR;
            const GT* src = G + min((int64_t)p0 * Q + r, n1q - 1) * ldg + min(c0 + (int64_t)ch * EPC, ldg_last);
            can_dma16(reinterpret_cast<const float*>(src), ts_addr + u * 1024);
        }
    };
    // lane (im, m): column m <= K4 of the value row of point im accumulates sum_j w_ij X2'[j][m] (column K4 of a value row is 1)
    const int im = lane >> 4, m = lane & 15;
    float accm = 0.f, E[Q], sK_sum = 0.f, l_acc = 0.f;
#pragma unroll
    for (int b = 0; b < Q; ++b) E[b] = 0.f;
    dma_tile(blockIdx.x);
    for (int ct = blockIdx.x; ct < ncoltiles; ct += gridDim.x) {
        const int j0 = ct * CP;
        WAVE_SYNC();
        for (int e = lane; e < CP * DP; e += 64) {
            const int r = e / DP, k = e - r * DP;
            X2s[e] = (j0 + r < n2) ? P2[(int64_t)(j0 + r) * Q * DP + k] : 0.f;
        }
        WAVE_SYNC();
        float dl[Q];
        dl[0] = 0.f;
#pragma unroll
        for (int b = 1; b < Q; ++b) dl[b] = X1s[ii * DP + cidx[b]] - X2s[j * DP + cidx[b]];
        float nn = 0.f;
        for (int k = 0; k < d; ++k) {
            const float t = X1s[ii * DP + k] - X2s[j * DP + k];
            nn = __builtin_fmaf(t, t, nn);
        }
        const bool valid = lane_on && p0 + i < n1 && j0 + j < n2;
        const float kv = valid ? s * expf(-0.5f * nn) : 0.f;
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): this tile's DMA is in
        asm volatile("" ::: "memory");
        const GT* blk = Ts + (ii * Q) * W + j * Q;
        float r0 = 0.f, c0 = 0.f, tr = 0.f, quad = 0.f, Et[Q], colv[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) { Et[b] = 0.f; colv[b] = 0.f; }
        float g00 = 0.f;
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            float rowv = 0.f;
#pragma unroll
            for (int b = 0; b < Q; ++b) {
                const float g = (float)blk[a * W + b];
                if (a == 0 && b == 0) g00 = g;
                else if (a == 0) { r0 = __builtin_fmaf(g, dl[b], r0); Et[b] = __builtin_fmaf(il, g, Et[b]); }
                else if (b == 0) { c0 = __builtin_fmaf(g, dl[a], c0); Et[a] = __builtin_fmaf(-il, g, Et[a]); }
                else {
                    if (a == b) tr += g;
                    rowv = __builtin_fmaf(g, dl[b], rowv);
                    colv[b] = __builtin_fmaf(g, dl[a], colv[b]);
                }
            }
            if (a > 0) { quad = __builtin_fmaf(dl[a], rowv, quad); Et[a] = __builtin_fmaf(-il2, rowv, Et[a]); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads are done: the buffer may take the next tile
        if (ct + (int)gridDim.x < ncoltiles) dma_tile(ct + (int)gridDim.x);
        const float e1 = il * (r0 - c0), e2 = il2 * (tr - quad);
        const float w = kv * (g00 + e1 + e2);                       // w_ij = k Gm = T00-bar
        float dots = 0.f;
#pragma unroll
        for (int b = 1; b < Q; ++b) {
            const float ec = kv * __builtin_fmaf(-il2, colv[b], Et[b]);
            E[b] += ec;
            dots = __builtin_fmaf(ec, dl[b], dots);
        }
        sK_sum += w;
        l_acc += kv * (e1 + 2.f * e2) - w * nn + dots;
        if (lane_on) Wt[i * CP + j] = w;
        WAVE_SYNC();
        if (m <= K4) {
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) accm = __builtin_fmaf(Wt[im * CP + jj], X2s[jj * DP + m], accm);
        }
    }
    // ---- the value rows of this workgroup's slab: columns 0 .. K4 from accm, the one-hot columns take the sums of E over the lanes of a point
    WAVE_SYNC();
    for (int e = lane; e < RP * NP; e += 64) rowbuf[e] = 0.f;
    WAVE_SYNC();
    if (m <= K4) rowbuf[im * NP + m] = accm;
#pragma unroll
    for (int b = 1; b < Q; ++b) {
        float x = E[b];
#pragma unroll
        for (int off = 1; off < CP; off <<= 1) x += __shfl_xor(x, off);
        E[b] = x;
    }
    WAVE_SYNC();
    if (lane_on && j == 0) {
#pragma unroll
        for (int b = 1; b < Q; ++b) rowbuf[i * NP + cidx[b]] += E[b];
    }
    WAVE_SYNC();
    float* myslab = slab + (int64_t)blockIdx.x * n1q * NP;
    for (int e = lane; e < H * NP; e += 64) {
        const int r = e / NP, col = e - r * NP;
        const int pt = r / Q, a = r - pt * Q;
        if ((int64_t)p0 * Q + r < n1q) myslab[((int64_t)p0 * Q + r) * NP + col] = (a == 0) ? rowbuf[pt * NP + col] : 0.f;
    }
    float l_sum = -il * l_acc;
    for (int off = 32; off > 0; off >>= 1) {
        sK_sum += __shfl_down(sK_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    if (lane == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        partials[bid * 2] = sK_sum;
        partials[bid * 2 + 1] = l_sum;
    }
}

#ifndef CAN2_BWD_WGS
#define CAN2_BWD_WGS (256 * 5)
#endif
inline bool canon2_ok(const Geom& g) { return g.q == 11 && g.K4 <= 12; }

extern "C" int dsvgp_kernel_canon2_supported(int d, int p) {
    Geom g;
    if (p < 1 || make_geom(d, p, g)) return 0;
    return canon2_ok(g) ? 1 : 0;
}

// K(x1, x2) with one-hot directions e_{dir_idx[a] - idx_base} on BOTH sides (see the block comment above); float or double output, jitter on
// the global diagonal as dsvgp_kernel_fwd.  Only the value rows of the packs are read.
extern "C" int dsvgp_kernel_fwd_canon2(dsvgp_ctx* ctx, const float* P1, int n1, const float* P2, int n2, int d, int p, const int* dir_idx,
                                       int idx_base, const float* hyp, float jitter, void* out, int64_t ld, int out_is_double) {
    if (!ctx || !P1 || !P2 || !hyp || !out || !dir_idx || n1 < 0 || n2 < 0 || p < 1) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (!canon2_ok(g)) return DSVGP_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    if (ld < (int64_t)n2 * g.q) return DSVGP_EINVAL;
    const int ovec = ((out_is_double ? (ld % 2 == 0 && (uintptr_t)out % 16 == 0) : (ld % 4 == 0 && (uintptr_t)out % 16 == 0)) ? 1 : 0) |
                     ((ctx->fwd_lower_only && P1 == P2 && n1 == n2) ? 8 : 0);
    const size_t lds = sizeof(float) * (64 * (size_t)g.q * g.q + 20 * (size_t)g.DP);
    dim3 grid(cdiv(n2, 16), cdiv(n1, 4));
    if (out_is_double)
        hipLaunchKernelGGL((kernel_fwd_canon2_kernel<11, double>), grid, dim3(64), lds, ctx->stream, P1, n1, P2, n2, d, g.DP, dir_idx, idx_base,
                           hyp, jitter, (double*)out, ld, ovec);
    else
        hipLaunchKernelGGL((kernel_fwd_canon2_kernel<11, float>), grid, dim3(64), lds, ctx->stream, P1, n1, P2, n2, d, g.DP, dir_idx, idx_base,
                           hyp, jitter, (float*)out, ld, ovec);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// backward of dsvgp_kernel_fwd_canon2 w.r.t. (x1, lengthscale, outputscale): accumulates (+=) like dsvgp_kernel_bwd (symmetric != 0: x1 == x2
// and G symmetric -- the K_ZZ case -- point gradients doubled); d_v1 receives nothing (the directions are fixed).  G's rows must be 16-byte
// pieces (ldg a multiple of 4 floats / 2 doubles, base 16-byte aligned): DSVGP_EINVAL otherwise -- the caller uses dsvgp_kernel_bwd.
// workspace: dsvgp_kernel_bwd_workspace_bytes(n1, n2, d, p).
extern "C" int dsvgp_kernel_bwd_canon2(dsvgp_ctx* ctx, const void* G, int64_t ldg, int g_is_double, const float* P1, const float* vnorm1,
                                       int n1, const float* P2, int n2, int d, int p, const int* dir_idx, int idx_base, const float* hyp,
                                       int symmetric, float* d_x1, float* d_v1, float* d_hyp, void* workspace) {
    if (!ctx || !G || !P1 || !P2 || !hyp || !d_x1 || !d_hyp || !workspace || !dir_idx || p < 1 || !vnorm1 || !d_v1) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (!canon2_ok(g)) return DSVGP_EINVAL;
    if (n1 <= 0 || n2 <= 0) return 0;
    const int64_t n1q = (int64_t)n1 * g.q, n2q = (int64_t)n2 * g.q;
    const int epc = g_is_double ? 2 : 4;
    if (ldg < n2q || ldg % epc != 0 || (uintptr_t)G % 16 != 0) return DSVGP_EINVAL;
    const int cp = g_is_double ? 8 : 16;
    const int rt = cdiv(n1, 4), ctiles = cdiv(n2, cp);
    // the slab count and the partial slots the workspace was sized for (the general kernels' tiling: dsvgp_kernel_bwd_workspace_bytes)
    const int cap_ns = bwd_nsplit(n1, n2, g);
    int tr_, tc_, wgs_;
    bwd_tiles(g, tr_, tc_, wgs_);
    const int64_t cap_part = (int64_t)cap_ns * cdiv(n1q, tr_);
    int ns = CAN2_BWD_WGS / rt;
    if (ns < 1) ns = 1;
    if (ns > ctiles) ns = ctiles;
    if (ns > cap_ns) ns = cap_ns;
    if ((int64_t)ns * rt > cap_part) ns = (int)(cap_part / rt);
    if (ns < 1) return DSVGP_EINVAL;
    float* slab = (float*)workspace;
    float* partials = slab + (size_t)cap_ns * n1q * g.NP;
    const int nch = 16 * g.q * g.q, ndma = (nch + 63) / 64;
    const size_t lds = sizeof(float) * ((size_t)ndma * 256 + (4 + cp) * (size_t)g.DP + 4 * cp + 4 * (size_t)g.NP);
    dim3 grid(ns, rt);
    if (g_is_double)
        hipLaunchKernelGGL((kernel_bwd_canon2_kernel<11, double>), grid, dim3(64), lds, ctx->stream, (const double*)G, ldg, P1, n1, P2, n2, d,
                           g.K4, g.DP, g.NP, dir_idx, idx_base, hyp, slab, partials);
    else
        hipLaunchKernelGGL((kernel_bwd_canon2_kernel<11, float>), grid, dim3(64), lds, ctx->stream, (const float*)G, ldg, P1, n1, P2, n2, d,
                           g.K4, g.DP, g.NP, dir_idx, idx_base, hyp, slab, partials);
    DSVGP_LAUNCH_CHECK();
    return finish_points(ctx, g, slab, ns, P1, vnorm1, n1, d, p, hyp, symmetric ? 2.f : 1.f, d_x1, d_v1, (const float*)partials, ns * rt, d_hyp);
}

// One kernel_bwd_points launch over the slab sets the kernel backwards have noted while ctx->defer_points was set (at most two: the one-call
// step's K_ZX-bar and K_ZZ-bar), with the step's scalar tail folded in when `scal` is given (PointsTail above).  Internal (csrc/step.hip).
int kernel_bwd_points_flush(dsvgp_ctx* ctx, const float* P1, const float* vnorm1, int n1, int d, int p, const float* hyp, float* d_x1,
                            float* d_v1, float* d_hyp, const float* scal, const float* kl0, double rows, double num_data, const float* rl,
                            const float* rs, const float* rn, float* drl, float* drs, float* drn, float* dconst, float* loss) {
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    const int n = ctx->n_deferred;
    ctx->n_deferred = 0;
    if (n < 1) return DSVGP_EINVAL;
    PointsTail tail{};
    if (scal) tail = PointsTail{scal, kl0, (float)(1.0 / rows), (float)(1.0 / num_data), rl, rs, rn, drl, drs, drn, dconst, loss};
    const dsvgp_ctx::PointsJob& a = ctx->deferred[0];
    if (n == 1) return launch_points(ctx->stream, g, a.slab, a.ns, P1, vnorm1, n1, d, p, hyp, a.sym, d_x1, d_v1, a.partials, a.nparts, d_hyp,
                                     nullptr, 0, 0.f, nullptr, 0, tail);
    const dsvgp_ctx::PointsJob& b = ctx->deferred[1];
    return launch_points(ctx->stream, g, a.slab, a.ns, P1, vnorm1, n1, d, p, hyp, a.sym, d_x1, d_v1, a.partials, a.nparts, d_hyp, b.slab, b.ns,
                         b.sym, b.partials, b.nparts, tail);
}

extern "C" int dsvgp_packed_width(int d) { return ((d + 3) & ~3) + 4; }

// column means + the constrained hyper-parameters (dsvgp_hyp_forward) in one launch: block 0 also writes hyp
__global__ void column_mean_hyp_kernel(const float* __restrict__ x, int n, int d, float* __restrict__ out, const float* rl,
                                       const float* rs, const float* rn, float* hyp) {
    __shared__ double part[256];
    const int k = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[(int64_t)i * d + k];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[k] = (float)(part[0] / (double)n);
        if (k == 0) {     // softplus constraints of gpytorch Positive / GreaterThan(1e-4) (csrc/elbo.hip: hyp_forward_kernel)
            auto sp = [](float v) { return v > 20.f ? v : log1pf(expf(v)); };
            hyp[0] = sp(rl[0]); hyp[1] = sp(rs[0]); hyp[2] = sp(rn[0]) + 1e-4f; hyp[3] = 0.f;
        }
    }
}
// centre + constrained hyper-parameters + the packed rows of (Z, V) and of (x, D) in one launch (pack_both_kernel)
int launch_pack_both(hipStream_t st, const float* Z, const float* V, int M, const float* X, const float* D, int B, int d, int p,
                     const float* rl, const float* rs, const float* rn, float* hyp, float* center, float* PZ, float* sZ, float* vZ,
                     float* PX, float* sX, float* vX) {
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    const int nbz = cdiv((int64_t)M * g.q, 256), nbx = cdiv((int64_t)B * g.q, 256);
    hipLaunchKernelGGL(pack_both_kernel, dim3(nbz + nbx), dim3(256), sizeof(float) * (d + 1), st, Z, V, M, X, D, B, d, p, rl, rs, rn, hyp,
                       center, PZ, sZ, vZ, PX, sX, vX, g.K4, g.DP, nbz);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

int launch_column_mean_hyp(hipStream_t st, const float* x, int n, int d, float* center, const float* rl, const float* rs,
                           const float* rn, float* hyp) {
    hipLaunchKernelGGL(column_mean_hyp_kernel, dim3(d), dim3(256), 0, st, x, n, d, center, rl, rs, rn, hyp);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_column_mean(dsvgp_ctx* ctx, const float* x, int n, int d, float* out) {
    if (!ctx || !x || !out || n < 1 || d < 1) return DSVGP_EINVAL;
    hipLaunchKernelGGL(column_mean_kernel, dim3(d), dim3(256), 0, ctx->stream, x, n, d, out);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_pack_points(dsvgp_ctx* ctx, const float* x, const float* v, int n, int d, int p,
                                 const float* hyp, const float* center, float* P, float* self, float* vnorm) {
    if (!ctx || !x || !hyp || !P || !self || n < 0 || (p > 0 && (!v || !vnorm))) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (n == 0) return 0;
    const int rows = n * g.q;
    hipLaunchKernelGGL(pack_points_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, ctx->stream, x, v, n, d, p, hyp, center,
                       P, self, vnorm, g.K4, g.DP);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_kernel_fwd(dsvgp_ctx* ctx, const float* P1, const float* self1, int n1, const float* P2,
                                const float* self2, int n2, int d, int p, const float* hyp, float jitter, void* out,
                                int64_t ld, int out_is_double) {
    if (!ctx || !P1 || !self1 || !P2 || !self2 || !hyp || !out || n1 < 0 || n2 < 0) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (n1 == 0 || n2 == 0) return 0;
    const int n1q = n1 * g.q, n2q = n2 * g.q;
    if (ld < n2q) return DSVGP_EINVAL;
    if (FWD_RUN && (g.q == 6 || g.q == 3) && g.NP <= 32 && n1q % g.q == 0) {
        // "run" kernel (16-byte stores): needs 16-byte aligned rows of the output
        const int esz = out_is_double ? 8 : 4;
        const bool aligned = (ld * esz) % 16 == 0 && (uintptr_t)out % 16 == 0;
        if (aligned) {
            const int TC = g.q == 6 ? 96 : 48;
            const int rt = cdiv(n1q, 48), ctiles = cdiv(n2q, TC);
            int ns = FWD_RUN_WGS_ / rt;
            if (ns < 1) ns = 1;
            if (ns > ctiles) ns = ctiles;
            const size_t p2w = (size_t)TC * (g.K4 + 5), ttw = (size_t)48 * (TC + 4);
            const size_t lds = sizeof(float) * (48 * (size_t)(g.K4 + 5) + (p2w > ttw ? p2w : ttw));
            dim3 grid(ns, rt);
#define DSVGP_FWD_RUN(OT_, Q_)                                                                                       \
            hipLaunchKernelGGL((kernel_fwd_run_kernel<OT_, Q_>), grid, dim3(64), lds, ctx->stream, P1, self1, n1q, P2, self2, \
                               n2q, g.K4, g.DP, hyp, jitter, (OT_*)out, ld)
            if (out_is_double) { if (g.q == 6) DSVGP_FWD_RUN(double, 6); else DSVGP_FWD_RUN(double, 3); }
            else { if (g.q == 6) DSVGP_FWD_RUN(float, 6); else DSVGP_FWD_RUN(float, 3); }
#undef DSVGP_FWD_RUN
            DSVGP_LAUNCH_CHECK();
            return 0;
        }
    }
#ifndef FWD_NO_SPLIT
    if (g.q == 11 && g.K4 <= 12) {      // full-gradient SVGP at d <= 12 (BASELINE config 3): the split-row kernel
        const int T = 44;
        const int rt = cdiv(n1q, T), ctiles = cdiv(n2q, T);
        int ns = FWD_PAIR_WGS_ / rt;
        if (ns < 1) ns = 1;
        if (ns > ctiles) ns = ctiles;
        const size_t p2w_ = 49 * (size_t)(g.K4 + 5), ttw_ = 48 * (size_t)FWD_PAIR_LDT;
        const size_t lds = sizeof(float) * (p2w_ > ttw_ ? p2w_ : ttw_);
        const int ovec = (!out_is_double && ld % 4 == 0 && (uintptr_t)out % 16 == 0) ? 2 : 0;
        dim3 grid(ns, rt);
        if (out_is_double)
            hipLaunchKernelGGL((kernel_fwd_split_kernel<double, 11, 4>), grid, dim3(64), lds, ctx->stream, P1, self1, n1q, P2, self2, n2q,
                               g.K4, g.DP, ovec, hyp, jitter, (double*)out, ld);
        else
            hipLaunchKernelGGL((kernel_fwd_split_kernel<float, 11, 4>), grid, dim3(64), lds, ctx->stream, P1, self1, n1q, P2, self2, n2q,
                               g.K4, g.DP, ovec, hyp, jitter, (float*)out, ld);
        DSVGP_LAUNCH_CHECK();
        return 0;
    }
#endif
    if ((g.q == 6 || g.q == 3) && g.NP <= 64) {
        const int T = (48 / g.q) * g.q;
        const int rt = cdiv(n1q, T), ctiles = cdiv(n2q, T);
        int ns = FWD_PAIR_WGS_ / rt;
        if (ns < 1) ns = 1;
        if (ns > ctiles) ns = ctiles;
        const size_t p2w_ = 49 * (size_t)(g.K4 + 5), ttw_ = 48 * (size_t)FWD_PAIR_LDT;        // (49: the scratch row of the predicate-free staging)
        const size_t lds = FWDP_LDS_EXTRA + sizeof(float) * (FWDP_OVERLAY ? (FWDP_AREG ? 0 : 48 * (size_t)(g.K4 + 5)) + (p2w_ > ttw_ ? p2w_ : ttw_)
                                                         : 2 * 48 * (size_t)(g.K4 + 5) + 48 * (size_t)FWD_PAIR_LDT);
        const int esz = out_is_double ? 8 : 4;
        // bit 0: 2-wide stores of the micro-block rows; bit 1: 16-byte stores of lane pairs (float output, 16-byte aligned rows)
        const int ovec = (((ld % 2 == 0) && ((uintptr_t)out % (2 * esz) == 0)) ? 1 : 0) |
                         ((!out_is_double && ld % 4 == 0 && (uintptr_t)out % 16 == 0 && FWDP_ST16) ? 2 : 0) |
                         ((ctx->fwd_lower_only && P1 == P2 && n1 == n2) ? 8 : 0);       // (bit 3: the block triangle the Cholesky factorisation reads)
        dim3 grid(ns, rt);
#define DSVGP_FWD_PAIR(OT_, Q_, KSM_)                                                                                \
        hipLaunchKernelGGL((kernel_fwd_pair_kernel<OT_, Q_, KSM_>), grid, dim3(64), lds, ctx->stream, P1, self1, n1q, P2, self2, \
                           n2q, g.K4, g.DP, ovec, hyp, jitter, (OT_*)out, ld)
        if (g.NP <= 32) {
            if (out_is_double) { if (g.q == 6) DSVGP_FWD_PAIR(double, 6, 8); else DSVGP_FWD_PAIR(double, 3, 8); }
            else { if (g.q == 6) DSVGP_FWD_PAIR(float, 6, 8); else DSVGP_FWD_PAIR(float, 3, 8); }
        } else {
            if (out_is_double) { if (g.q == 6) DSVGP_FWD_PAIR(double, 6, 16); else DSVGP_FWD_PAIR(double, 3, 16); }
            else { if (g.q == 6) DSVGP_FWD_PAIR(float, 6, 16); else DSVGP_FWD_PAIR(float, 3, 16); }
        }
#undef DSVGP_FWD_PAIR
        DSVGP_LAUNCH_CHECK();
        return 0;
    }
    const int Rc = g.R, Rr = g.R >= 2 ? g.R / 2 : g.R;
    const int Tr = Rr * g.q, Tc = Rc * g.q, Trp = (Tr + 15) & ~15, Tcp = (Tc + 15) & ~15;
    const size_t lds = sizeof(float) * ((size_t)(Trp + Tcp) * (g.K4 + 1) + (size_t)Trp * LDT + Trp + Tcp + (size_t)Rr * Rc);
    dim3 grid(cdiv(n2q, Tc), cdiv(n1q, Tr));
    (void)hipFuncSetAttribute((const void*)kernel_fwd_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)kernel_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (out_is_double)
        hipLaunchKernelGGL(kernel_fwd_kernel<double>, grid, dim3(256), lds, ctx->stream, P1, self1, n1q, P2, self2,
                           n2q, g.q, Rr, Rc, g.K4, g.DP, hyp, jitter, (double*)out, ld);
    else
        hipLaunchKernelGGL(kernel_fwd_kernel<float>, grid, dim3(256), lds, ctx->stream, P1, self1, n1q, P2, self2,
                           n2q, g.q, Rr, Rc, g.K4, g.DP, hyp, jitter, (float*)out, ld);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_kernel_diag(dsvgp_ctx* ctx, int n, int p, const float* hyp, float* out) {
    if (!ctx || !hyp || !out || n < 0 || p < 0) return DSVGP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(kernel_diag_kernel, dim3(cdiv((int64_t)n * (p + 1), 256)), dim3(256), 0, ctx->stream, n, p,
                       hyp, out);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t dsvgp_kernel_bwd_workspace_bytes(int n1, int n2, int d, int p) {
    Geom g;
    if (make_geom(d, p, g) || n1 <= 0 || n2 <= 0) return 0;
    const int ns = bwd_nsplit(n1, n2, g);
    int tr, tc, wgs;
    bwd_tiles(g, tr, tc, wgs);
    const int rt = cdiv((int64_t)n1 * g.q, tr);
    return sizeof(float) * ((size_t)ns * n1 * g.q * g.NP + (size_t)2 * ns * rt + 64);
}

extern "C" int dsvgp_kernel_bwd(dsvgp_ctx* ctx, const void* G, int64_t ldg, int g_is_double, const float* P1,
                                const float* self1, const float* vnorm1, int n1, const float* P2,
                                const float* self2, int n2, int d, int p, const float* hyp, int symmetric,
                                float* d_x1, float* d_v1, float* d_hyp, void* workspace) {
    if (!ctx || !G || !P1 || !self1 || !P2 || !self2 || !hyp || !d_x1 || !d_hyp || !workspace) return DSVGP_EINVAL;
    if (p > 0 && (!vnorm1 || !d_v1)) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (n1 <= 0 || n2 <= 0) return 0;
    const int n1q = n1 * g.q, n2q = n2 * g.q;
    if (ldg < n2q) return DSVGP_EINVAL;
    const int ns = bwd_nsplit(n1, n2, g);
    int tr_, tc_, wgs_;
    bwd_tiles(g, tr_, tc_, wgs_);
    const int rt = cdiv(n1q, tr_);
    float* slab = (float*)workspace;
    float* partials = slab + (size_t)ns * n1q * g.NP;
    if (bwd_use_split(g)) {
        const size_t lds = sizeof(float) * (((49 * (size_t)(g.NP + 1) + 3) & ~(size_t)3) + 48 * (size_t)PAIR_LDT + 44 * (size_t)PAIR_LDT);
        const int esz = g_is_double ? 8 : 4;
        const int gvec = (ldg % 4 == 0) && ((uintptr_t)G % (4 * esz) == 0);     // 4-wide loads of the upstream tile rows (tile origins are multiples of 44)
        dim3 grid(ns, rt);
        if (g_is_double)
            hipLaunchKernelGGL((kernel_bwd_split_kernel<double, 11, 4>), grid, dim3(64), lds, ctx->stream, (const double*)G, ldg, P1, self1, n1q,
                               P2, self2, n2q, g.K4, g.DP, g.NP, gvec, hyp, slab, partials);
        else
            hipLaunchKernelGGL((kernel_bwd_split_kernel<float, 11, 4>), grid, dim3(64), lds, ctx->stream, (const float*)G, ldg, P1, self1, n1q,
                               P2, self2, n2q, g.K4, g.DP, g.NP, gvec, hyp, slab, partials);
        DSVGP_LAUNCH_CHECK();
    } else if (bwd_use_pair(g)) {
        const size_t lds = sizeof(float) * (((49 * (size_t)(g.NP + 1) + 3) & ~(size_t)3) + 48 * (size_t)PAIR_LDT);      // (49: the scratch row of the predicate-free staging)
        const int esz = g_is_double ? 8 : 4;
        const int gvec = (ldg % 2 == 0) && ((uintptr_t)G % (2 * esz) == 0);    // 2-wide loads of the micro-block rows
        dim3 grid(ns, rt);
#define DSVGP_PAIR_LAUNCH(GT_, Q_, KSM_)                                                                                \
        hipLaunchKernelGGL((kernel_bwd_pair_kernel<GT_, Q_, KSM_>), grid, dim3(64), lds, ctx->stream, (const GT_*)G, ldg, P1, self1, \
                           n1q, P2, self2, n2q, g.K4, g.DP, g.NP, gvec, hyp, slab, partials)
        if (g.NP <= 32) {
            if (g_is_double) { if (g.q == 6) DSVGP_PAIR_LAUNCH(double, 6, 8); else DSVGP_PAIR_LAUNCH(double, 3, 8); }
            else { if (g.q == 6) DSVGP_PAIR_LAUNCH(float, 6, 8); else DSVGP_PAIR_LAUNCH(float, 3, 8); }
        } else {
            if (g_is_double) { if (g.q == 6) DSVGP_PAIR_LAUNCH(double, 6, 16); else DSVGP_PAIR_LAUNCH(double, 3, 16); }
            else { if (g.q == 6) DSVGP_PAIR_LAUNCH(float, 6, 16); else DSVGP_PAIR_LAUNCH(float, 3, 16); }
        }
#undef DSVGP_PAIR_LAUNCH
        DSVGP_LAUNCH_CHECK();
    } else {
    const int Trp = (g.Tr + 15) & ~15, Tcp = (g.T + 15) & ~15;
    const size_t lds = sizeof(float) * ((size_t)Trp * (g.K4 + 1) + (size_t)Tcp * (g.NP + 1) + 2 * (size_t)Trp * LDT + Trp + Tcp +
                                        (size_t)g.Rr * g.R + 2 * (size_t)g.Tr * g.R + 2 * BWD_NW);
    // 4-wide loads of the upstream tile need 4-element aligned tile origins and rows
    const int gvec = (g.T % 4 == 0) && (ldg % 4 == 0) && ((uintptr_t)G % 32 == 0);
    dim3 grid(ns, rt);
    if (g_is_double)
        dispatch_bwd<double>(ctx->stream, grid, lds, (const double*)G, ldg, P1, self1, n1q, P2, self2, n2q, g, gvec, hyp, slab, partials);
    else
        dispatch_bwd<float>(ctx->stream, grid, lds, (const float*)G, ldg, P1, self1, n1q, P2, self2, n2q, g, gvec, hyp, slab, partials);
    DSVGP_LAUNCH_CHECK();
    }
    const float sym = symmetric ? 2.f : 1.f;
    return finish_points(ctx, g, slab, ns, P1, vnorm1, n1, d, p, hyp, sym, d_x1, d_v1, (const float*)partials, ns * rt, d_hyp);
}
