// Block-kernel assembly for RBFKernelDirectionalGrad (reference
// directionalvi/RBFKernelDirectionalGrad.py:41-119) and its backward, gfx950.
//
// Formulation.  For every point pack the (p+1) rows  [x/ell ; v_1 ; ... ; v_p]  (unit directions)
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

#ifndef ASM_ABLATE
#define ASM_ABLATE 0      // tools only: 1 = no global stores, 2 = stores of a constant (no MFMA / epilogue math)
#endif

namespace {

constexpr int TMAX = 96;   // tile rows/cols of the interleaved matrix handled per workgroup
constexpr int LDT = 100;   // LDS row stride of the T / G tiles (100 % 32 == 4 -> acc writes conflict-free per half)
constexpr int MAXACC = 9;  // backward: max 16x16 output tiles per wave (6 x ceil(DP/16) / 4)

using f4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fdiv_small(int e, float inv) { return (int)(((float)e + 0.5f) * inv); }

// ---- pack -------------------------------------------------------------------------------------
__global__ void pack_points_kernel(const float* __restrict__ x, const float* __restrict__ v, int n, int d,
                                   int p, const float* __restrict__ hyp, float* __restrict__ P,
                                   float* __restrict__ self, float* __restrict__ vnorm, int K4, int DP) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = p + 1;
    if (row >= n * q) return;
    const int i = row / q, a = row - i * q;
    const float ell = hyp[0];
    float* Pr = P + (int64_t)row * DP;
    const float* xi = x + (int64_t)i * d;
    if (a == 0) {
        float acc = 0.f;
        for (int k = 0; k < d; ++k) {
            const float xt = xi[k] / ell;          // x.div(lengthscale), :67-68
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
            acc = __builtin_fmaf(vh, xi[k] / ell, acc);
        }
        for (int k = d; k < DP; ++k) Pr[k] = 0.f;
        self[row] = acc;
        vnorm[(int64_t)i * p + (a - 1)] = nrm;
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

__global__ void kernel_diag_kernel(int n, int p, const float* __restrict__ hyp, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * (p + 1)) return;
    const float ell = hyp[0], s = hyp[1];
    out[idx] = (idx % (p + 1) == 0) ? s : s / (ell * ell);
}

// ---- backward -------------------------------------------------------------------------------------
// grid = (nsplit, row tiles).  Workgroup (sx, by) sweeps column tiles sx, sx+nsplit, ... of row tile by.
template <typename GT>
__global__ __launch_bounds__(256) void kernel_bwd_kernel(const GT* __restrict__ G, int64_t ldg,
                                                         const float* __restrict__ P1, const float* __restrict__ self1,
                                                         int n1q, const float* __restrict__ P2,
                                                         const float* __restrict__ self2, int n2q, int q, int R,
                                                         int K4, int DP, int NP, const float* __restrict__ hyp,
                                                         float* __restrict__ slab, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = NP + 1;
    float* P1s = smem;
    float* P2s = P1s + TMAX * LDP;
    float* Ts = P2s + TMAX * LDP;
    float* Gs = Ts + TMAX * LDT;
    float* s1 = Gs + TMAX * LDT;
    float* s2 = s1 + TMAX;
    float* red = s2 + TMAX;  // 2*4 floats

    const int T = R * q, Tp = (T + 15) & ~15;
    const int ntr = Tp / 16, nnp = NP / 16;
    const int row0 = blockIdx.y * T;
    const int ncoltiles = (n2q + T - 1) / T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float ell = hyp[0], s = hyp[1];
    const float il = 1.f / ell, il2 = il * il;
    const int p = q - 1;
    const float invR = 1.f / (float)R;

    stage_pack(P1s, s1, P1, self1, row0, T, n1q, Tp, DP, K4, LDP);

    f4 acc[MAXACC];
#pragma unroll
    for (int i = 0; i < MAXACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float sK_sum = 0.f, l_sum = 0.f;

    for (int ct = blockIdx.x; ct < ncoltiles; ct += gridDim.x) {
        const int col0 = ct * T;
        __syncthreads();  // previous iteration's MFMA reads of P2s / Gs are done
        stage_pack(P2s, s2, P2, self2, col0, T, n2q, Tp, DP, NP, LDP);
        for (int e = threadIdx.x; e < Tp * Tp; e += 256) {
            const int r = e / Tp, c = e - r * Tp;
            const int64_t gr = row0 + r, gc = col0 + c;
            Gs[r * LDT + c] = (r < T && c < T && gr < n1q && gc < n2q) ? (float)G[gr * ldg + gc] : 0.f;
        }
        __syncthreads();
        mfma_T(Ts, P1s, P2s, ntr, ntr, K4, LDP);
        __syncthreads();

        // one thread per (point i, point j) pair: Gbar micro-block -> Tbar micro-block, in place in Gs
        for (int pid = threadIdx.x; pid < R * R; pid += 256) {
            const int pi = fdiv_small(pid, invR), pj = pid - pi * R;
            const int r0 = pi * q, c0 = pj * q;
            float* g0 = Gs + r0 * LDT + c0;
            const float* t0 = Ts + r0 * LDT + c0;
            const float nn = fmaxf(s1[r0] + s2[c0] - 2.f * t0[0], 0.f);
            const float k = s * expf(-0.5f * nn);
            const float g00 = g0[0];
            float kbar = g00, e1 = 0.f, e2 = 0.f, dotw = 0.f, dotu = 0.f;
            for (int b = 1; b <= p; ++b) {
                const float w = t0[b] - s2[c0 + b];
                const float g = g0[b];
                kbar += g * w * il;
                e1 += g * w * k * il;
            }
            for (int a = 1; a <= p; ++a) {
                const float u = s1[r0 + a] - t0[a * LDT];
                const float ga = g0[a * LDT];
                kbar -= ga * u * il;
                e1 -= ga * u * k * il;
                float accu = -ga * il;
                for (int b = 1; b <= p; ++b) {
                    const float w = t0[b] - s2[c0 + b];
                    const float gab = g0[a * LDT + b];
                    const float h = (t0[a * LDT + b] - u * w) * il2;
                    kbar += gab * h;
                    e2 += gab * h * k;
                    accu -= gab * w * il2;
                    g0[a * LDT + b] = k * gab * il2;          // Tbar_ab
                }
                const float ubar = k * accu;
                g0[a * LDT] = -ubar;                            // Tbar_a0
                dotu += ubar * u;
            }
            for (int b = 1; b <= p; ++b) {
                const float w = t0[b] - s2[c0 + b];
                float wbar = g0[b] * il * k;
                for (int a = 1; a <= p; ++a) {
                    const float u = s1[r0 + a] - t0[a * LDT];
                    wbar -= g0[a * LDT + b] * u;                // Tbar_ab * u_a
                }
                g0[b] = wbar;                                   // Tbar_0b
                dotw += wbar * w;
            }
            const float nbar = -0.5f * k * kbar;
            g0[0] = k * kbar;                                   // Tbar_00 = -2 nbar
            sK_sum += g00 * k + e1 + e2;
            l_sum -= (e1 + 2.f * e2 + 2.f * nbar * nn + dotw + dotu) * il;
        }
        __syncthreads();

        // dP1[Tp, NP] += Tbar[Tp, Tp] . P2ext[Tp, NP]   (A from Gs, B from P2s)
#pragma unroll
        for (int si = 0; si < MAXACC; ++si) {
            const int id = wave + 4 * si;
            if (id < ntr * nnp) {
                const int tr = id / nnp, tn = id - tr * nnp;
                const float* pa = Gs + (tr * 16 + (lane & 15)) * LDT + (lane >> 4);
                const float* pb = P2s + (lane >> 4) * LDP + tn * 16 + (lane & 15);
                f4 c = acc[si];
                for (int kk = 0; kk < Tp; kk += 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk], pb[kk * LDP], c, 0, 0, 0);
                acc[si] = c;
            }
        }
    }

    // partial slab: slab[split][row][NP]
    float* myslab = slab + ((int64_t)blockIdx.x * n1q) * NP;
#pragma unroll
    for (int si = 0; si < MAXACC; ++si) {
        const int id = wave + 4 * si;
        if (id < ntr * nnp) {
            const int tr = id / nnp, tn = id - tr * nnp;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = tr * 16 + (lane >> 4) * 4 + r;
                const int64_t gr = row0 + rr;
                if (rr < T && gr < n1q) myslab[gr * NP + tn * 16 + (lane & 15)] = acc[si][r];
            }
        }
    }
    // block reduce the two scalars
    for (int off = 32; off > 0; off >>= 1) {
        sK_sum += __shfl_down(sK_sum, off);
        l_sum += __shfl_down(l_sum, off);
    }
    __syncthreads();
    if (lane == 0) { red[wave * 2] = sK_sum; red[wave * 2 + 1] = l_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        partials[bid * 2] = red[0] + red[2] + red[4] + red[6];
        partials[bid * 2 + 1] = red[1] + red[3] + red[5] + red[7];
    }
}

// one 64-thread block per point: slabs -> d_x1, d_v1 (through the x/ell scaling and the direction normalisation)
__global__ __launch_bounds__(64) void kernel_bwd_points_kernel(const float* __restrict__ slab, int nsplit,
                                                               const float* __restrict__ P1,
                                                               const float* __restrict__ vnorm1, int n1, int d, int p,
                                                               int K4, int DP, int NP, const float* __restrict__ hyp,
                                                               float sym, float* __restrict__ d_x1,
                                                               float* __restrict__ d_v1) {
    extern __shared__ float dPs[];          // [q][DP] summed over the split slabs, then [q] dots
    const int i = blockIdx.x, t = threadIdx.x;
    const int q = p + 1;
    const int64_t n1q = (int64_t)n1 * q;
    const float ell = hyp[0];
    for (int e = t; e < q * DP; e += 64) {
        const int a = e / DP, col = e - a * DP;
        float sum = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) sum += slab[((int64_t)sp * n1q + (int64_t)i * q + a) * NP + col];
        dPs[e] = sum;
    }
    __syncthreads();
    const float* xt = P1 + (int64_t)i * q * DP;
    float* dots = dPs + q * DP;
    // vhat-bar_a = dP[a,:] + alphabar_a x~ ; dots[a] = vhat_a . vhat-bar_a ; alphabar_a = -dP[a,K4]
    for (int a = 1 + t; a <= p; a += 64) {
        const float* vh = P1 + ((int64_t)i * q + a) * DP;
        const float ab = -dPs[a * DP + K4];
        float dot = 0.f;
        for (int k = 0; k < d; ++k) dot += vh[k] * (dPs[a * DP + k] + ab * xt[k]);
        dots[a] = dot;
    }
    __syncthreads();
    const float nbar = -0.5f * dPs[K4];
    for (int k = t; k < d; k += 64) {
        // x~bar = dP[0,:] + 2 nbar x~ + sum_a alphabar_a vhat_a
        float xb = dPs[k] + 2.f * nbar * xt[k];
        for (int a = 1; a <= p; ++a) xb += -dPs[a * DP + K4] * P1[((int64_t)i * q + a) * DP + k];
        d_x1[(int64_t)i * d + k] += sym * xb / ell;
    }
    for (int e = t; e < p * d; e += 64) {
        const int a = 1 + e / d, k = e - (a - 1) * d;
        const float* vh = P1 + ((int64_t)i * q + a) * DP;
        const float vb = dPs[a * DP + k] - dPs[a * DP + K4] * xt[k];
        const float inv = 1.f / vnorm1[(int64_t)i * p + (a - 1)];
        d_v1[((int64_t)i * p + (a - 1)) * d + k] += sym * (vb - vh[k] * dots[a]) * inv;   // normalisation Jacobian
    }
}

__global__ void kernel_bwd_scalars_kernel(const float* __restrict__ partials, int nblocks,
                                          const float* __restrict__ hyp, float* __restrict__ d_hyp) {
    __shared__ double r0[256], r1[256];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    r0[threadIdx.x] = a; r1[threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { r0[threadIdx.x] += r0[threadIdx.x + off]; r1[threadIdx.x] += r1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d_hyp[1] += (float)(r0[0] / (double)hyp[1]);   // d outputscale = sum(G o K)/s
        d_hyp[0] += (float)r1[0];                       // d lengthscale
    }
}

struct Geom { int q, R, T, K4, DP, NP; };
inline int make_geom(int d, int p, Geom& g) {
    g.q = p + 1;
    if (d < 1 || p < 0 || g.q > TMAX) return DSVGP_EINVAL;
    g.R = TMAX / g.q;
    g.T = g.R * g.q;
    g.K4 = (d + 3) & ~3;
    g.DP = g.K4 + 4;
    g.NP = (g.DP + 15) & ~15;
    if (6 * (g.NP / 16) > 4 * MAXACC) return DSVGP_EINVAL;   // d <= 88
    return 0;
}
inline int bwd_nsplit(int n1, int n2, const Geom& g) {
    const int rt = cdiv((int64_t)n1 * g.q, g.T), ctiles = cdiv((int64_t)n2 * g.q, g.T);
    int ns = 1024 / rt;
    if (ns < 1) ns = 1;
    if (ns > ctiles) ns = ctiles;
    return ns;
}

}  // namespace

extern "C" int dsvgp_packed_width(int d) { return ((d + 3) & ~3) + 4; }

extern "C" int dsvgp_pack_points(dsvgp_ctx* ctx, const float* x, const float* v, int n, int d, int p,
                                 const float* hyp, float* P, float* self, float* vnorm) {
    if (!ctx || !x || !hyp || !P || !self || n < 0 || (p > 0 && (!v || !vnorm))) return DSVGP_EINVAL;
    Geom g;
    if (int rc = make_geom(d, p, g)) return rc;
    if (n == 0) return 0;
    const int rows = n * g.q;
    hipLaunchKernelGGL(pack_points_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, ctx->stream, x, v, n, d, p, hyp, P,
                       self, vnorm, g.K4, g.DP);
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
    const int rt = cdiv((int64_t)n1 * g.q, g.T);
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
    const int rt = cdiv(n1q, g.T);
    float* slab = (float*)workspace;
    float* partials = slab + (size_t)ns * n1q * g.NP;
    const size_t lds = sizeof(float) * (2 * TMAX * (g.NP + 1) + 2 * TMAX * LDT + 2 * TMAX + 8);
    dim3 grid(ns, rt);
    if (g_is_double) {
        (void)hipFuncSetAttribute((const void*)kernel_bwd_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel_bwd_kernel<double>, grid, dim3(256), lds, ctx->stream, (const double*)G, ldg, P1, self1,
                           n1q, P2, self2, n2q, g.q, g.R, g.K4, g.DP, g.NP, hyp, slab, partials);
    } else {
        (void)hipFuncSetAttribute((const void*)kernel_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel_bwd_kernel<float>, grid, dim3(256), lds, ctx->stream, (const float*)G, ldg, P1, self1,
                           n1q, P2, self2, n2q, g.q, g.R, g.K4, g.DP, g.NP, hyp, slab, partials);
    }
    DSVGP_LAUNCH_CHECK();
    const float sym = symmetric ? 2.f : 1.f;
    hipLaunchKernelGGL(kernel_bwd_points_kernel, dim3(n1), dim3(64), sizeof(float) * (g.q * g.DP + g.q + 1), ctx->stream,
                       slab, ns, P1, vnorm1, n1, d, p, g.K4, g.DP, g.NP, hyp, sym, d_x1, d_v1);
    DSVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(kernel_bwd_scalars_kernel, dim3(1), dim3(256), 0, ctx->stream, partials, ns * rt, hyp, d_hyp);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
