// ELBO terms, KL, Cholesky-backward helper, minibatch gather and fused Adam (gfx950).
// All kernels here are HBM-bound streaming passes over [M', B'] or [M', M'] operands or tiny
// reductions; they use coalesced row-major sweeps (lane -> consecutive column).
#include "common.h"

namespace {

constexpr float KXX_JITTER = 1e-4f;     // data_data_covar.add_jitter(1e-4), reference DGVS.py:197,203
constexpr float NOISE_FLOOR = 1e-4f;    // GaussianLikelihood noise constraint GreaterThan(1e-4)
constexpr float MIN_VARIANCE = 1e-6f;   // gpytorch settings.min_variance (float)

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void hyp_forward_kernel(const float* rl, const float* rs, const float* rn, float* hyp) {
    if (threadIdx.x == 0) {
        hyp[0] = softplusf(rl[0]);
        hyp[1] = softplusf(rs[0]);
        hyp[2] = softplusf(rn[0]) + NOISE_FLOOR;
        hyp[3] = 0.f;
    }
}
__global__ void hyp_backward_kernel(const float* rl, const float* rs, const float* rn, const float* dh, float* drl,
                                    float* drs, float* drn) {
    if (threadIdx.x == 0) {
        drl[0] += dh[0] * sigmoidf(rl[0]);
        drs[0] += dh[1] * sigmoidf(rs[0]);
        drn[0] += dh[2] * sigmoidf(rn[0]);
    }
}

// everything the step does with its scalar accumulators in ONE launch: fold the data-term scalars into d_hyp,
// softplus chain rule to the raw parameters, d constant, and loss = -ll / rows + KL / num_data
__global__ void step_epilogue_kernel(const float* scal, const float* kl0, float inv_rows, float inv_num_data,
                                     const float* rl, const float* rs, const float* rn, float* dh, float* drl,
                                     float* drs, float* drn, float* dconst, float* loss) {
    if (threadIdx.x == 0) {
        const float d0 = dh[0] + scal[4], d1 = dh[1] + scal[3], d2 = dh[2] + scal[1];
        dh[0] = d0; dh[1] = d1; dh[2] = d2;
        drl[0] += d0 * sigmoidf(rl[0]);
        drs[0] += d1 * sigmoidf(rs[0]);
        drn[0] += d2 * sigmoidf(rn[0]);
        dconst[0] += scal[2];
        loss[0] = -scal[0] * inv_rows + kl0[0] * inv_num_data;
    }
}

// partial column statistics over a row chunk: part[chunk][0][j] = sum_i A_ij m_i ; part[chunk][1][j] = sum_i W^2 - A^2
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ W, int64_t ldw, int Mp, int ncols,
                                                       const float* __restrict__ m, int rows_per_chunk,
                                                       float* __restrict__ part) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * rows_per_chunk, i1 = min(Mp, i0 + rows_per_chunk);
    if (j >= ncols) return;
    float sm = 0.f, sq = 0.f;
    if (W == A) {                 // ELBO fast path / zero middle term: only mu is wanted, A is streamed once
#pragma unroll 4
        for (int i = i0; i < i1; ++i) sm = fmaf(A[(int64_t)i * lda + j], m[i], sm);
    } else {
        for (int i = i0; i < i1; ++i) {
            const float a = A[(int64_t)i * lda + j], w = W[(int64_t)i * ldw + j];
            sm = fmaf(a, m[i], sm);
            sq += w * w - a * a;
        }
    }
    part[((int64_t)blockIdx.y * 2) * ncols + j] = sm;
    part[((int64_t)blockIdx.y * 2 + 1) * ncols + j] = sq;
}
__global__ void colstats_finish_kernel(const float* __restrict__ part, int nchunk, int ncols, int p,
                                       const float* __restrict__ constant, const float* __restrict__ hyp,
                                       float* __restrict__ mu, float* __restrict__ var) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    float sm = 0.f, sq = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        sm += part[((int64_t)c * 2) * ncols + j];
        sq += part[((int64_t)c * 2 + 1) * ncols + j];
    }
    const float ell = hyp[0], s = hyp[1];
    const float dg = (j % (p + 1) == 0) ? s : s / (ell * ell);
    mu[j] = sm + constant[0];                 // test_mean is the constant for ALL rows, DGVS.py:126
    var[j] = dg + KXX_JITTER + sq;
}

// scalars out: 0 sum_ll, 1 d_noise, 2 d_constant, 3 d_outputscale(diag), 4 d_lengthscale(diag)
__global__ __launch_bounds__(256) void likelihood_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                         const float* __restrict__ y, int ncols, int p,
                                                         const float* __restrict__ hyp, int mll_type, float inv_rows,
                                                         float* __restrict__ mu_bar, float* __restrict__ var_bar,
                                                         float* __restrict__ varn_out, float* __restrict__ scal, float* __restrict__ partials) {
    __shared__ float red[5][4];
    const float ell = hyp[0], s = hyp[1], noise = hyp[2];
    const float LOG2PI = 1.8378770664093453f;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = blockIdx.x * 256 + threadIdx.x; j < ncols; j += gridDim.x * 256) {
        const float mj = mu[j], r = y[j] - mj;
        const float vraw = var[j] + noise;
        const bool clamped = vraw < MIN_VARIANCE;
        const float vn = clamped ? MIN_VARIANCE : vraw;        // likelihood(model(x)).variance
        float ll, dmu, dvn, dnoise;                              // derivatives of ll (not yet of the loss)
        if (mll_type == 0) {   // expected_log_prob: -0.5[((y-mu)^2 + vn)/noise + log noise + log 2pi]
            ll = -0.5f * ((r * r + vn) / noise + logf(noise) + LOG2PI);
            dmu = r / noise;
            dvn = -0.5f / noise;
            dnoise = 0.5f * (r * r + vn) / (noise * noise) - 0.5f / noise;
        } else {               // log_marginal: log N(y; mu, vn + noise)
            const float tot = fmaxf(vn + noise, 1e-8f);
            ll = -0.5f * (r * r / tot + logf(tot) + LOG2PI);
            dmu = r / tot;
            dvn = 0.5f * (r * r / (tot * tot) - 1.f / tot);
            dnoise = dvn;
        }
        const float dvar = clamped ? 0.f : dvn;                 // vn = var + noise (unless clamped)
        dnoise += dvar;
        // loss = -(sum ll)/rows + KL/num_data
        const float mb = -dmu * inv_rows, vb = -dvar * inv_rows;
        mu_bar[j] = mb;
        var_bar[j] = vb;
        varn_out[j] = vn;
        const bool isf = (j % (p + 1)) == 0;
        acc[0] += ll;
        acc[1] += -dnoise * inv_rows;
        acc[2] += mb;
        acc[3] += vb * (isf ? 1.f : 1.f / (ell * ell));          // var = s*dg + ...
        acc[4] += isf ? 0.f : vb * (-2.f * s / (ell * ell * ell));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        float v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[q][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const float v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (partials) partials[blockIdx.x * 8 + threadIdx.x] = v;     // deterministic mode: summed in block order afterwards
        else atomicAdd(&scal[threadIdx.x], v);
    }
}
// out[q] = sum over blocks of partials[b][q] in the fixed order b = 0, 1, ... (q < nq <= 8; one wave)
__global__ void partials_reduce_kernel(const float* __restrict__ partials, int nblocks, int nq, float* __restrict__ out) {
    const int q = threadIdx.x;
    if (q >= nq) return;
    double s = 0;
    for (int b = 0; b < nblocks; ++b) s += partials[b * 8 + q];
    out[q] = (float)s;
}

__global__ __launch_bounds__(256) void abar_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ U,
                                                   int64_t ldu, int Mp, int ncols, const float* __restrict__ m,
                                                   const float* __restrict__ mu_bar, const float* __restrict__ var_bar,
                                                   float* __restrict__ Ab, int64_t ldab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    const float mb = mu_bar[j], vb2 = 2.f * var_bar[j];
    for (int i = blockIdx.y; i < Mp; i += gridDim.y)
        Ab[(int64_t)i * ldab + j] = m[i] * mb + vb2 * (U[(int64_t)i * ldu + j] - A[(int64_t)i * lda + j]);
}

__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ A, int64_t lda, int Mp, int ncols,
                                                     const float* __restrict__ vec, float* __restrict__ out) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    float s = 0.f;
    const float* row = A + (int64_t)i * lda;
    if ((lda & 3) == 0 && (ncols & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)vec & 15) == 0) {
        using f4 = float __attribute__((ext_vector_type(4)));       // 16-byte loads: the row is streamed once
        for (int j = threadIdx.x * 4; j < ncols; j += 1024) {
            const f4 a = *reinterpret_cast<const f4*>(row + j), v = *reinterpret_cast<const f4*>(vec + j);
            s = fmaf(a[0], v[0], fmaf(a[1], v[1], fmaf(a[2], v[2], fmaf(a[3], v[3], s))));
        }
    } else {
        for (int j = threadIdx.x; j < ncols; j += 256) s = fmaf(row[j], vec[j], s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[i] += red[0] + red[1] + red[2] + red[3];
}

__global__ void sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// G[i][j] (i<j) = G[j][i]; lower part and diagonal unchanged: Phi(G)+Phi(G)^T
__global__ void phi_sym_kernel(double* __restrict__ G, int n, int64_t ldg) {
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;                       // handle upper block (bi, bj), bj >= bi, from lower block (bj, bi)
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8) {
        const int gi = bj * 32 + r, gj = bi * 32 + tx;   // lower block element (gi, gj)
        tile[r][tx] = (gi < n && gj < n) ? G[(int64_t)gi * ldg + gj] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int gi = bi * 32 + r, gj = bj * 32 + tx;   // upper block element (gi, gj) <- lower (gj, gi)
        if (gi < n && gj < n && gj > gi) G[(int64_t)gi * ldg + gj] = tile[tx][r];
    }
}

// dst (fp64, full symmetric) = mirror of the lower triangle (diagonal included) of src (fp32): the widening of an fp32 product's
// lower-triangular result and Phi(.) + Phi(.)^T of it in ONE pass (csrc/step.hip, chol_tail)
__global__ void widen_sym_kernel(const float* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int n) {
    __shared__ float tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;                       // lower block (bi, bj), bj <= bi -> dst blocks (bi, bj) and (bj, bi)
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8) {
        const int gi = bi * 32 + r, gj = bj * 32 + tx;
        const float v = (gi < n && gj < n) ? src[(int64_t)gi * lds + gj] : 0.f;
        tile[r][tx] = v;
        if (gi < n && gj < n && gj <= gi) dst[(int64_t)gi * ldd + gj] = (double)v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int gi = bj * 32 + r, gj = bi * 32 + tx;   // upper element (gi, gj) <- lower (gj, gi)
        if (gi < n && gj < n && gj > gi) dst[(int64_t)gi * ldd + gj] = (double)tile[tx][r];
    }
}
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ in, int64_t ldi, int rows, int cols,
                                 T* __restrict__ out, int64_t ldo) {
    __shared__ T tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(int64_t)(r0 + r) * ldi + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < cols && r0 + tx < rows) out[(int64_t)(c0 + r) * ldo + r0 + tx] = tile[tx][r];
}

__global__ void add_diag_kernel(double* A, int n, int64_t lda, double delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(int64_t)i * lda + i] += delta;
}

__global__ void gather_kernel(const float* __restrict__ X, const float* __restrict__ Y, const int64_t* __restrict__ idx,
                              int nb, int d, int ycols, const int* __restrict__ cols, int p, float* __restrict__ xb,
                              float* __restrict__ yb, const float* __restrict__ E, float* __restrict__ Db) {
    const int b = blockIdx.x;
    const int64_t src = idx[b];
    for (int k = threadIdx.x; k < d; k += blockDim.x) xb[(int64_t)b * d + k] = X[src * d + k];
    for (int c = threadIdx.x; c <= p; c += blockDim.x) yb[(int64_t)b * (p + 1) + c] = Y[src * ycols + cols[c]];
    if (E)      // the batch's derivative directions: row a of point b = row cols[a + 1] - 1 of E (the canonical basis, :238)
        for (int e = threadIdx.x; e < p * d; e += blockDim.x) {
            const int a = e / d, k = e - a * d;
            Db[((int64_t)b * p + a) * d + k] = E[(int64_t)(cols[a + 1] - 1) * d + k];
        }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = grad[i];
        const float mi = b1 * m[i] + (1.f - b1) * g;
        const float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;      // torch.optim.Adam (single tensor path)
        param[i] -= (lr / bc1) * (mi / denom);
    }
}

// all tensors of one optimizer in ONE launch: workgroup b belongs to the tensor whose [first, first + blocks) holds b
struct AdamTable {
    float* param[DSVGP_ADAM_MAX_TENSORS];
    const float* grad[DSVGP_ADAM_MAX_TENSORS];
    float* m[DSVGP_ADAM_MAX_TENSORS];
    float* v[DSVGP_ADAM_MAX_TENSORS];
    int64_t n[DSVGP_ADAM_MAX_TENSORS];
    int first[DSVGP_ADAM_MAX_TENSORS + 1];
    int count;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable t, float lr, float b1, float b2, float eps,
                                                         float bc1, float bc2_sqrt, const int* __restrict__ guard) {
    if (guard && *guard != 0) return;       // (the Cholesky status word of the step whose gradients these are: a failed factorisation must not touch the parameters)
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first[k + 1]) ++k;
    const int nb = t.first[k + 1] - t.first[k];
    float* __restrict__ param = t.param[k];
    const float* __restrict__ grad = t.grad[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const int64_t n = t.n[k];
    auto update = [&](int64_t i) {
        const float g = grad[i];
        const float mi = b1 * m[i] + (1.f - b1) * g;
        const float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param[i] -= (lr / bc1) * (mi / denom);
    };
    if (n < 0) {
        // a square matrix of which only the lower triangle is a parameter (chol_variational_covar: the reference masks the rest in its
        // forward, gradient and both moments are exactly zero there and the update is exactly zero): rows dealt cyclically to the
        // workgroups (the triangular row lengths balance), the strict upper triangle is not touched -- half the traffic of this launch
        const int64_t dim = -n;
        for (int64_t r = blockIdx.x - t.first[k]; r < dim; r += nb)
            for (int64_t c = threadIdx.x; c <= r; c += 256) update(r * dim + c);
        return;
    }
    for (int64_t i = (int64_t)(blockIdx.x - t.first[k]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) update(i);
}

// Graph-capturable variants: every per-step scalar comes from device memory.
//  adam_multi_dev_kernel: hp = {lr, step} (floats, written by a captured host-to-device copy), guard (may be null): the
//    update is skipped while *guard != 0 (the Cholesky status word: a failed factorisation must not touch the parameters).
//  scale3_kernel: x_k *= 1 / (noise rows) = 2 vbar for up to three arrays (Z-bar, V-bar, d_hyp[0..1] of the ELBO fast path,
//    whose K_ZX-bar / L-bar products run unscaled when the noise is not known on the host).
__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const AdamTable t, const float* __restrict__ hp, float b1, float b2,
                                                             float eps, const int* __restrict__ guard) {
    if (guard && *guard != 0) return;
    const float lr = hp[0];
    const double step = (double)hp[1];
    const float bc1 = (float)(1.0 - pow((double)b1, step)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first[k + 1]) ++k;
    const int nb = t.first[k + 1] - t.first[k];
    float* __restrict__ param = t.param[k];
    const float* __restrict__ grad = t.grad[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const int64_t n = t.n[k];
    for (int64_t i = (int64_t)(blockIdx.x - t.first[k]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        const float g = grad[i];
        const float mi = b1 * m[i] + (1.f - b1) * g;
        const float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param[i] -= (lr / bc1) * (mi / denom);
    }
}
__global__ __launch_bounds__(256) void scale3_kernel(float* __restrict__ x0, int64_t n0, float* __restrict__ x1, int64_t n1,
                                                     float* __restrict__ x2, int64_t n2, const float* __restrict__ hyp,
                                                     float inv_rows) {
    const float s = inv_rows / hyp[2];
    const int64_t tot = n0 + n1 + n2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        if (i < n0) x0[i] *= s;
        else if (i < n0 + n1) x1[i - n0] *= s;
        else x2[i - n0 - n1] *= s;
    }
}

// scale3_kernel (Z-bar, V-bar, d_hyp[0..1] *= 2 vbar) + step_epilogue_kernel in one launch: the first thread of the grid scales
// d_hyp[0..1] itself and goes on with the scalar tail
__global__ __launch_bounds__(256) void scale_epilogue_kernel(float* __restrict__ x0, int64_t n0, float* __restrict__ x1, int64_t n1,
                                                             const float* __restrict__ hyp, float inv_rows, const float* scal,
                                                             const float* kl0, float inv_num_data, const float* rl, const float* rs,
                                                             const float* rn, float* dh, float* drl, float* drs, float* drn,
                                                             float* dconst, float* loss) {
    const float s = inv_rows / hyp[2];
    const int64_t tot = n0 + n1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        if (i < n0) x0[i] *= s;
        else x1[i - n0] *= s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float d0 = dh[0] * s + scal[4], d1 = dh[1] * s + scal[3], d2 = dh[2] + scal[1];
        dh[0] = d0; dh[1] = d1; dh[2] = d2;
        drl[0] += d0 * sigmoidf(rl[0]);
        drs[0] += d1 * sigmoidf(rs[0]);
        drn[0] += d2 * sigmoidf(rn[0]);
        dconst[0] += scal[2];
        loss[0] = -scal[0] * inv_rows + kl0[0] * inv_num_data;
    }
}

// scal layout of likelihood_kernel: 0 sum_ll, 1 d_noise, 2 d_constant, 3 d_outputscale(diag), 4 d_lengthscale(diag)
__device__ __forceinline__ void elbo_fast_finalize_body(const float* __restrict__ sums, const float* __restrict__ hyp, int npts,
                                                        int p, float inv_rows, float* __restrict__ scal) {
    const float ell = hyp[0], s = hyp[1], noise = hyp[2];
    const float LOG2PI = 1.8378770664093453f;
    const float nrow = (float)npts * (float)(p + 1);
    const float sum_r2 = sums[0], sum_mubar = sums[1];
    const float sum_prior = (float)npts * s * (1.f + (float)p / (ell * ell)) + nrow * 1e-4f;   // s*diag + K_XX jitter
    const float sum_var = sum_prior + sums[2] - sums[3];
    const float vbar = 0.5f / noise * inv_rows;                                                 // dLoss/dvar_j (constant)
    scal[0] = -0.5f * ((sum_r2 + sum_var) / noise + nrow * (1.f + logf(noise) + LOG2PI));
    scal[1] = 0.5f * inv_rows * (-(sum_r2 + sum_var) / (noise * noise) + nrow / noise);
    scal[2] = sum_mubar;
    scal[3] = vbar * (float)npts * (1.f + (float)p / (ell * ell));
    scal[4] = vbar * (float)npts * (float)p * (-2.f * s / (ell * ell * ell));
    scal[5] = scal[6] = scal[7] = 0.f;
}

// ---- one pass over the lower triangles of L_S and L_S-bar (round 3) ---------------------------------------------------
// Replaces kl_kernel / kl_scaled_kernel / trace_kernel (one workgroup per row, one scalar load per thread: 0.6 TB/s, 147 +
// 92 us at M' = 3000) by ONE WAVE per row with 16-byte loads / stores where the rows allow, rows dealt cyclically to the
// waves (the triangular row lengths balance), every reduction finished inside the wave (no LDS, no barrier) and NO atomics:
// per-row partial sums go to caller scratch and a one-workgroup pass adds them in a fixed order (run-to-run identical).
//   flags bit 0 (SCALE): dLS <- dLS * inv_rows / noise before the KL gradient is added (the fast path's unscaled G L_S)
//         bit 1 (KL):    rowkl[i] = 1/2 (m_i^2 + sum_{j<=i} L_ij^2 - 1 - log L_ii^2), dLS += dKL/dL_S / num_data, d_m += m / num_data
//         bit 2 (TRACE): rowtr[i] = sum_{j<=i} L_ij T_ij with T = dLS as it is READ (before scaling)
// The strict upper triangle of dLS is written as zero (the masked part of the parameter carries no gradient).
template <bool VEC>
__global__ __launch_bounds__(256) void ls_rows_kernel(const float* __restrict__ m, const float* __restrict__ LS, int64_t ldls, int Mp,
                                                      float inv_nd, int flags, const float* __restrict__ hyp, float inv_rows,
                                                      float* __restrict__ rowkl, float* __restrict__ rowtr,
                                                      const float* __restrict__ dm_src, float* __restrict__ d_m,
                                                      float* __restrict__ dLS, int64_t lddls) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const bool scale = flags & 1, kl = flags & 2, trace = flags & 4;
    const float sc = scale ? inv_rows / hyp[2] : 1.f;
    for (int i = wave; i < Mp; i += nwaves) {
        const float* __restrict__ l = LS + (int64_t)i * ldls;
        float* __restrict__ t = dLS + (int64_t)i * lddls;
        float sll = 0.f, slt = 0.f;
        if constexpr (VEC) {
            for (int j = lane * 4; j < Mp; j += 256) {          // (Mp % 4 == 0 on this path)
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j <= i) {
                    const float4 lv = *(const float4*)(l + j);
                    const float4 tv = *(const float4*)(t + j);
                    const float la[4] = {lv.x, lv.y, lv.z, lv.w}, ta[4] = {tv.x, tv.y, tv.z, tv.w};
                    float oa[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool in = j + c <= i;
                        const float lc = in ? la[c] : 0.f;
                        sll = fmaf(lc, lc, sll);
                        slt = fmaf(lc, in ? ta[c] : 0.f, slt);
                        const float g = kl ? ((j + c == i) ? (lc - 1.f / lc) * inv_nd : lc * inv_nd) : 0.f;
                        oa[c] = in ? fmaf(ta[c], sc, g) : 0.f;
                    }
                    o = make_float4(oa[0], oa[1], oa[2], oa[3]);
                }
                *(float4*)(t + j) = o;
            }
        } else {
            for (int j = lane; j < Mp; j += 64) {
                float o = 0.f;
                if (j <= i) {
                    const float lc = l[j], tc = t[j];
                    sll = fmaf(lc, lc, sll);
                    slt = fmaf(lc, tc, slt);
                    const float g = kl ? ((j == i) ? (lc - 1.f / lc) * inv_nd : lc * inv_nd) : 0.f;
                    o = fmaf(tc, sc, g);
                }
                t[j] = o;
            }
        }
        for (int off = 32; off > 0; off >>= 1) { sll += __shfl_down(sll, off); slt += __shfl_down(slt, off); }
        if (lane == 0) {
            if (kl) {
                const float lii = l[i], mi = m[i];
                rowkl[i] = 0.5f * (mi * mi + sll - 1.f - logf(lii * lii));
                d_m[i] = (dm_src ? dm_src[i] : d_m[i]) + mi * inv_nd;       // (dm_src: the data part b = A mu_bar, copied on the way)
            } else {
                if (rowkl) rowkl[i] = 0.f;
                if (dm_src) d_m[i] = dm_src[i];
            }
            if (trace) rowtr[i] = slt;
        }
    }
}
// fixed-order sums of the per-row partials (double accumulation): out_kl[0] = sum rowkl; sums[2] = t1_scale sum rowtr,
// sums[3] = trace(G) (either group optional)
__global__ __launch_bounds__(256) void ls_rows_reduce_kernel(const float* __restrict__ rowkl, const float* __restrict__ rowtr,
                                                             const float* __restrict__ G, int64_t ldg, int n, float t1_scale,
                                                             float* __restrict__ out_kl, float* __restrict__ sums,
                                                             const float* __restrict__ hyp, int fin_npts, int fin_p, float inv_rows,
                                                             float* __restrict__ fin_scal, int mute = 0) {
    __shared__ double red[3][256];
    double a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (rowkl) a += rowkl[i];
        if (rowtr) { b += rowtr[i]; c += G[(int64_t)i * ldg + i]; }
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
#pragma unroll
            for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // mute: a data-parallel rank whose G is the GLOBAL Gram matrix but which is not the rank that counts the trace terms and the
        // KL value (they enter the summed loss once; the gradients were formed regardless)
        if (rowkl && out_kl) out_kl[0] = mute ? 0.f : (float)red[0][0];
        if (rowtr && sums) { sums[2] = mute ? 0.f : t1_scale * (float)red[1][0]; sums[3] = mute ? 0.f : (float)red[2][0]; }
        if (fin_scal) elbo_fast_finalize_body(sums, hyp, fin_npts, fin_p, inv_rows, fin_scal);   // (one launch fewer in the one-call step)
    }
}
static int launch_ls_rows(hipStream_t st, const float* m, const float* LS, int64_t ldls, int Mp, float inv_nd, int flags,
                          const float* hyp, float inv_rows, float* rowkl, float* rowtr, float* d_m, float* dLS, int64_t lddls,
                          const float* dm_src = nullptr) {
    const bool vec = Mp % 4 == 0 && ldls % 4 == 0 && lddls % 4 == 0 && ((uintptr_t)LS % 16 == 0) && ((uintptr_t)dLS % 16 == 0);
    int blocks = cdiv(Mp, 4);
    if (blocks > 2048) blocks = 2048;
    if (vec) hipLaunchKernelGGL(ls_rows_kernel<true>, dim3(blocks), dim3(256), 0, st, m, LS, ldls, Mp, inv_nd, flags, hyp, inv_rows,
                                rowkl, rowtr, dm_src, d_m, dLS, lddls);
    else hipLaunchKernelGGL(ls_rows_kernel<false>, dim3(blocks), dim3(256), 0, st, m, LS, ldls, Mp, inv_nd, flags, hyp, inv_rows,
                            rowkl, rowtr, dm_src, d_m, dLS, lddls);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// ---- ELBO fast path (constant dLoss/dvar): residuals, traces, scalar assembly -----------------------
__global__ __launch_bounds__(256) void residual_kernel(const float* __restrict__ mu, const float* __restrict__ y,
                                                       int ncols, const float* __restrict__ hyp, float inv_rows,
                                                       float* __restrict__ mu_bar, float* __restrict__ sums,
                                                       float* __restrict__ partials) {
    __shared__ float red[2][4];
    const float noise = hyp[2];
    float a0 = 0.f, a1 = 0.f;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < ncols; j += gridDim.x * 256) {
        const float r = y[j] - mu[j];
        const float mb = -r / noise * inv_rows;
        mu_bar[j] = mb;
        a0 = fmaf(r, r, a0);
        a1 += mb;
    }
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (partials) partials[blockIdx.x * 8 + threadIdx.x] = v;     // deterministic mode
        else atomicAdd(&sums[threadIdx.x], v);
    }
}
// colstats_finish_kernel + residual_kernel in one launch (the one-call step: two ~5 us launches at M' = 600): mu_j = sum of the
// chunk partials + c, r_j = y_j - mu_j, mu-bar_j, sums[0] += r^2, sums[1] += mu-bar
__global__ __launch_bounds__(256) void colstats_residual_kernel(const float* __restrict__ part, int nchunk, int ncols, int p,
                                                                const float* __restrict__ constant, const float* __restrict__ hyp,
                                                                float* __restrict__ mu, float* __restrict__ var,
                                                                const float* __restrict__ y, float inv_rows,
                                                                float* __restrict__ mu_bar, float* __restrict__ sums) {
    __shared__ float red[2][4];
    const float ell = hyp[0], s = hyp[1], noise = hyp[2], c0 = constant[0];
    float a0 = 0.f, a1 = 0.f;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < ncols; j += gridDim.x * 256) {
        float sm = 0.f, sq = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            sm += part[((int64_t)c * 2) * ncols + j];
            sq += part[((int64_t)c * 2 + 1) * ncols + j];
        }
        const float dg = (j % (p + 1) == 0) ? s : s / (ell * ell);
        const float mj = sm + c0;
        mu[j] = mj;
        var[j] = dg + KXX_JITTER + sq;
        const float r = y[j] - mj;
        const float mb = -r / noise * inv_rows;
        mu_bar[j] = mb;
        a0 = fmaf(r, r, a0);
        a1 += mb;
    }
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) atomicAdd(&sums[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
// sums[2] += sum_{i>=j} LS_ij T1_ij  (= |L_S^T A|_F^2) ; sums[3] += trace(G)
__global__ __launch_bounds__(256) void trace_kernel(const float* __restrict__ LS, int64_t ldls,
                                                    const float* __restrict__ T1, int64_t ldt,
                                                    const float* __restrict__ G, int64_t ldg, int n,
                                                    float t1_scale, float* __restrict__ sums) {
    // rows i = blockIdx.x, blockIdx.x + gridDim.x, ... (interleaved: the triangular row lengths balance), ONE pair of
    // atomics per workgroup (one pair per row serialised 2 n atomics on two addresses: 79 us at n = 3000)
    __shared__ float red[4];
    float s = 0.f, gd = 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float* __restrict__ l = LS + (int64_t)i * ldls;
        const float* __restrict__ t = T1 + (int64_t)i * ldt;
        for (int j = threadIdx.x; j <= i; j += 256) s = fmaf(l[j], t[j], s);
        if (threadIdx.x == 0) gd += G[(int64_t)i * ldg + i];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2], t1_scale * (red[0] + red[1] + red[2] + red[3]));
        atomicAdd(&sums[3], gd);
    }
}
// scal layout of likelihood_kernel: 0 sum_ll, 1 d_noise, 2 d_constant, 3 d_outputscale(diag), 4 d_lengthscale(diag)
__global__ void elbo_fast_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ hyp, int npts,
                                          int p, float inv_rows, float* __restrict__ scal) {
    if (threadIdx.x == 0) elbo_fast_finalize_body(sums, hyp, npts, p, inv_rows, scal);
}
__global__ void mirror_lower_f32_kernel(float* __restrict__ G, int n, int64_t ldg) {
    __shared__ float tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8) {
        const int gi = bj * 32 + r, gj = bi * 32 + tx;
        tile[r][tx] = (gi < n && gj < n) ? G[(int64_t)gi * ldg + gj] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int gi = bi * 32 + r, gj = bj * 32 + tx;
        if (gi < n && gj < n && gj > gi) G[(int64_t)gi * ldg + gj] = tile[tx][r];
    }
}
__global__ void add_diag_f32_kernel(float* A, int n, int64_t lda, float delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(int64_t)i * lda + i] += delta;
}

// [S | .] -> [S - I | m noise rows]: the right-hand side of the [Q' | a / (2 vbar)] solve in one pass (2 vbar = 1 / (noise rows))
__global__ void sminus_i_col_kernel(float* A, int n, int64_t lda, const float* __restrict__ m, const float* __restrict__ hyp,
                                    float rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        A[(int64_t)i * lda + i] -= 1.f;
        A[(int64_t)i * lda + n] = m[i] * (hyp[2] * rows);
    }
}

// Packed lower triangle (row i at offset i(i+1)/2, i+1 entries) + `nextra` trailing floats: the data-parallel all-reduce
// operand ([tril(G) ; b^T] or [tril(L_S-bar) ; m-bar]) at half the dense volume.  One workgroup per row.
__global__ void tril_pack_f32_kernel(const float* __restrict__ src, int64_t ld, int n, const float* __restrict__ extra,
                                     int nextra, float* __restrict__ dst) {
    const int i = blockIdx.x;
    if (i < n) {
        const float* r = src + (int64_t)i * ld;
        float* o = dst + (int64_t)i * (i + 1) / 2;
        for (int j = threadIdx.x; j <= i; j += blockDim.x) o[j] = r[j];
    } else {
        float* o = dst + (int64_t)n * (n + 1) / 2;
        for (int j = threadIdx.x + (i - n) * blockDim.x; j < nextra; j += blockDim.x * (gridDim.x - n)) o[j] = extra[j];
    }
}
__global__ void tril_unpack_f32_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int64_t ld,
                                       float* __restrict__ extra, int nextra) {
    const int i = blockIdx.x;
    if (i < n) {
        const float* r = src + (int64_t)i * (i + 1) / 2;
        float* o = dst + (int64_t)i * ld;
        for (int j = threadIdx.x; j <= i; j += blockDim.x) o[j] = r[j];
    } else {
        const float* r = src + (int64_t)n * (n + 1) / 2;
        for (int j = threadIdx.x + (i - n) * blockDim.x; j < nextra; j += blockDim.x * (gridDim.x - n)) extra[j] = r[j];
    }
}

}  // namespace

extern "C" int dsvgp_hyp_forward(dsvgp_ctx* ctx, const float* rl, const float* rs, const float* rn, float* hyp) {
    if (!ctx || !rl || !rs || !rn || !hyp) return DSVGP_EINVAL;
    hipLaunchKernelGGL(hyp_forward_kernel, dim3(1), dim3(64), 0, ctx->stream, rl, rs, rn, hyp);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_hyp_backward(dsvgp_ctx* ctx, const float* rl, const float* rs, const float* rn, const float* dh,
                                  float* drl, float* drs, float* drn) {
    if (!ctx || !rl || !rs || !rn || !dh || !drl || !drs || !drn) return DSVGP_EINVAL;
    hipLaunchKernelGGL(hyp_backward_kernel, dim3(1), dim3(64), 0, ctx->stream, rl, rs, rn, dh, drl, drs, drn);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_step_epilogue(dsvgp_ctx* ctx, const float* scal, const float* kl0, double rows, double num_data,
                                   const float* rl, const float* rs, const float* rn, float* dh, float* drl,
                                   float* drs, float* drn, float* dconst, float* loss) {
    if (!ctx || !scal || !kl0 || !rl || !rs || !rn || !dh || !drl || !drs || !drn || !dconst || !loss || rows <= 0 ||
        num_data <= 0)
        return DSVGP_EINVAL;
    hipLaunchKernelGGL(step_epilogue_kernel, dim3(1), dim3(64), 0, ctx->stream, scal, kl0, (float)(1.0 / rows),
                       (float)(1.0 / num_data), rl, rs, rn, dh, drl, drs, drn, dconst, loss);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

static inline int stats_chunks(int Mp) { int c = cdiv(Mp, 128); return c > 32 ? 32 : (c < 1 ? 1 : c); }

extern "C" size_t dsvgp_stats_workspace_bytes(int Mp, int ncols) {
    if (Mp <= 0 || ncols <= 0) return 0;
    return sizeof(float) * (size_t)2 * stats_chunks(Mp) * ncols;
}
extern "C" int dsvgp_predictive_stats(dsvgp_ctx* ctx, const float* A, int64_t lda, const float* W, int64_t ldw, int Mp,
                                      int ncols, int p, const float* m, const float* constant, const float* hyp,
                                      float* mu, float* var, void* workspace) {
    if (!ctx || !A || !W || !m || !constant || !hyp || !mu || !var || !workspace || Mp <= 0 || ncols < 0 || p < 0)
        return DSVGP_EINVAL;
    if (ncols == 0) return 0;
    const int nch = stats_chunks(Mp), rpc = cdiv(Mp, nch);
    hipLaunchKernelGGL(colstats_kernel, dim3(cdiv(ncols, 256), nch), dim3(256), 0, ctx->stream, A, lda, W, ldw, Mp,
                       ncols, m, rpc, (float*)workspace);
    DSVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(colstats_finish_kernel, dim3(cdiv(ncols, 256)), dim3(256), 0, ctx->stream,
                       (const float*)workspace, nch, ncols, p, constant, hyp, mu, var);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_likelihood_terms(dsvgp_ctx* ctx, const float* mu, const float* var, const float* y, int ncols,
                                      int p, const float* hyp, int mll_type, double global_rows, float* mu_bar,
                                      float* var_bar, float* varn_out, float* out_scalars) {
    if (!ctx || !mu || !var || !y || !hyp || !mu_bar || !var_bar || !varn_out || !out_scalars || ncols < 0 ||
        global_rows <= 0 || (mll_type != 0 && mll_type != 1))
        return DSVGP_EINVAL;
    hipError_t e = hipMemsetAsync(out_scalars, 0, 8 * sizeof(float), ctx->stream);
    if (e != hipSuccess) return 1000 + (int)e;
    if (ncols == 0) return 0;
    int blocks = cdiv(ncols, 256);
    if (blocks > 512) blocks = 512;
    float* partials = (ctx->det_slab && ctx->det_bytes >= (size_t)blocks * 8 * sizeof(float)) ? (float*)ctx->det_slab : nullptr;
    hipLaunchKernelGGL(likelihood_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mu, var, y, ncols, p, hyp, mll_type,
                       (float)(1.0 / global_rows), mu_bar, var_bar, varn_out, out_scalars, partials);
    DSVGP_LAUNCH_CHECK();
    if (partials) {
        hipLaunchKernelGGL(partials_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float*)partials, blocks, 5, out_scalars);
        DSVGP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dsvgp_abar(dsvgp_ctx* ctx, const float* A, int64_t lda, const float* U, int64_t ldu, int Mp, int ncols,
                          const float* m, const float* mu_bar, const float* var_bar, float* Abar, int64_t ldab) {
    if (!ctx || !A || !U || !m || !mu_bar || !var_bar || !Abar || Mp <= 0 || ncols < 0) return DSVGP_EINVAL;
    if (ncols == 0) return 0;
    int gy = Mp < 64 ? Mp : 64;
    hipLaunchKernelGGL(abar_kernel, dim3(cdiv(ncols, 256), gy), dim3(256), 0, ctx->stream, A, lda, U, ldu, Mp, ncols, m,
                       mu_bar, var_bar, Abar, ldab);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_rowdot(dsvgp_ctx* ctx, const float* A, int64_t lda, int Mp, int ncols, const float* vec,
                            float* out_accum) {
    if (!ctx || !A || !vec || !out_accum || Mp <= 0 || ncols < 0) return DSVGP_EINVAL;
    hipLaunchKernelGGL(rowdot_kernel, dim3(Mp), dim3(256), 0, ctx->stream, A, lda, Mp, ncols, vec, out_accum);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_kl_terms(dsvgp_ctx* ctx, const float* m, const float* LS, int64_t ldls, int Mp, double num_data,
                              float* kl_out, float* d_m, float* d_LS, int64_t lddls) {
    if (!ctx || !m || !LS || !kl_out || !d_m || !d_LS || Mp <= 0 || num_data <= 0) return DSVGP_EINVAL;
    // kl_out must have room for 1 + Mp floats: [0] = KL, [1..Mp] = per-row scratch
    int rc = launch_ls_rows(ctx->stream, m, LS, ldls, Mp, (float)(1.0 / num_data), 2, nullptr, 0.f, kl_out + 1, nullptr, d_m, d_LS, lddls);
    if (rc) return rc;
    hipLaunchKernelGGL(ls_rows_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)(kl_out + 1), (const float*)nullptr,
                       (const float*)nullptr, (int64_t)0, Mp, 0.f, kl_out, (float*)nullptr, (const float*)nullptr, 0, 0, 0.f, (float*)nullptr);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_phi_symmetrize(dsvgp_ctx* ctx, double* G, int n, int64_t ldg) {
    if (!ctx || !G || n <= 0 || ldg < n) return DSVGP_EINVAL;
    const int nb = cdiv(n, 32);
    hipLaunchKernelGGL(phi_sym_kernel, dim3(nb, nb), dim3(32, 8), 0, ctx->stream, G, n, ldg);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_transpose_f64(dsvgp_ctx* ctx, const double* in, int64_t ldi, int rows, int cols, double* out,
                                   int64_t ldo) {
    if (!ctx || !in || !out || in == out || rows <= 0 || cols <= 0 || ldi < cols || ldo < rows) return DSVGP_EINVAL;
    hipLaunchKernelGGL(transpose_kernel<double>, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, ctx->stream, in,
                       ldi, rows, cols, out, ldo);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_transpose_f32(dsvgp_ctx* ctx, const float* in, int64_t ldi, int rows, int cols, float* out,
                                   int64_t ldo) {
    if (!ctx || !in || !out || in == out || rows <= 0 || cols <= 0 || ldi < cols || ldo < rows) return DSVGP_EINVAL;
    hipLaunchKernelGGL(transpose_kernel<float>, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, ctx->stream, in,
                       ldi, rows, cols, out, ldo);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// ---- fp64 matrix-vector products of the float64 model mode (_step64.py: mu = A^T m, b = A mu-bar on the [M', B'] panel) ----
// y = A x: one wave per row, 16-byte loads, butterfly sum.   y = A^T x: a thread per pair of columns, rows in chunks over
// blockIdx.y, fp64 atomics onto the zeroed y.  (As N = 1 products on the 128 x 128 GEMM kernel they took 0.53 ms each at C4: 1.1 TB/s.)
__global__ __launch_bounds__(256) void gemv64_rows_kernel(const double* __restrict__ A, int64_t lda, int M, int N,
                                                          const double* __restrict__ x, double* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const double* a = A + (int64_t)row * lda;
    double s = 0.0;
    const bool vec = (lda % 2 == 0) && (((uintptr_t)A % 16) == 0) && (((uintptr_t)x % 16) == 0);
    if (vec) {
        const int n2 = N / 2;
        for (int j = lane; j < n2; j += 64) {
            const double2 av = *reinterpret_cast<const double2*>(a + 2 * j);
            const double2 xv = *reinterpret_cast<const double2*>(x + 2 * j);
            s = fma(av.x, xv.x, s);
            s = fma(av.y, xv.y, s);
        }
        if ((N & 1) && lane == 0) s = fma(a[N - 1], x[N - 1], s);
    } else {
        for (int j = lane; j < N; j += 64) s = fma(a[j], x[j], s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) y[row] = s;
}
__global__ __launch_bounds__(256) void gemv64_cols_kernel(const double* __restrict__ A, int64_t lda, int M, int N,
                                                          const double* __restrict__ x, int rows_per_chunk,
                                                          double* __restrict__ y) {
    const int j = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const int i0 = blockIdx.y * rows_per_chunk, i1 = min(M, i0 + rows_per_chunk);
    const bool two = j + 1 < N, vec = two && (lda % 2 == 0) && (((uintptr_t)A % 16) == 0);
    double s0 = 0.0, s1 = 0.0;
    if (vec) {
#pragma unroll 4
        for (int i = i0; i < i1; ++i) {
            const double2 av = *reinterpret_cast<const double2*>(A + (int64_t)i * lda + j);
            const double xi = x[i];
            s0 = fma(av.x, xi, s0);
            s1 = fma(av.y, xi, s1);
        }
    } else {
        for (int i = i0; i < i1; ++i) {
            const double xi = x[i];
            s0 = fma(A[(int64_t)i * lda + j], xi, s0);
            if (two) s1 = fma(A[(int64_t)i * lda + j + 1], xi, s1);
        }
    }
    atomicAdd(&y[j], s0);
    if (two) atomicAdd(&y[j + 1], s1);
}
extern "C" int dsvgp_gemv_f64(dsvgp_ctx* ctx, int trans, const double* A, int64_t lda, int M, int N, const double* x,
                              double* y) {
    if (!ctx || !A || !x || !y || M <= 0 || N <= 0 || lda < N) return DSVGP_EINVAL;
    if (!trans) {
        hipLaunchKernelGGL(gemv64_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, ctx->stream, A, lda, M, N, x, y);
    } else {
        hipError_t e = hipMemsetAsync(y, 0, (size_t)N * sizeof(double), ctx->stream);
        if (e != hipSuccess) return 1000 + (int)e;
        int nch = cdiv(M, 64);
        if (nch > 64) nch = 64;
        const int rpc = cdiv(M, nch);
        hipLaunchKernelGGL(gemv64_cols_kernel, dim3(cdiv(N, 512), cdiv(M, rpc)), dim3(256), 0, ctx->stream, A, lda, M, N, x, rpc, y);
    }
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_add_diag(dsvgp_ctx* ctx, double* A, int n, int64_t lda, double delta) {
    if (!ctx || !A || n <= 0) return DSVGP_EINVAL;
    hipLaunchKernelGGL(add_diag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, A, n, lda, delta);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_gather_batch(dsvgp_ctx* ctx, const float* X, const float* Y, const int64_t* idx, int nb, int d,
                                  int ycols, const int* cols, int p, float* xb, float* yb, const float* E, float* Db) {
    if (!ctx || !X || !Y || !idx || !cols || !xb || !yb || nb < 0 || d <= 0 || p < 0 || ycols <= p || (E && !Db))
        return DSVGP_EINVAL;
    if (nb == 0) return 0;
    hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(64), 0, ctx->stream, X, Y, idx, nb, d, ycols, cols, p, xb, yb, E, Db);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_adam_step(dsvgp_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               int64_t n, float lr, float beta1, float beta2, float eps, int step) {
    if (!ctx || !param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return DSVGP_EINVAL;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    int blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, ctx->stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// the float64 model mode's update (the reference's experiment scripts train under torch.set_default_dtype(torch.float64),
// experiments/synthetic/exp_script.py:56): the same rule in double
struct AdamTable64 {
    double* param[DSVGP_ADAM_MAX_TENSORS];
    const double* grad[DSVGP_ADAM_MAX_TENSORS];
    double* m[DSVGP_ADAM_MAX_TENSORS];
    double* v[DSVGP_ADAM_MAX_TENSORS];
    int64_t n[DSVGP_ADAM_MAX_TENSORS];
    int first[DSVGP_ADAM_MAX_TENSORS + 1];
    int count;
};
__global__ __launch_bounds__(256) void adam_multi_f64_kernel(const AdamTable64 t, double lr, double b1, double b2, double eps,
                                                             double bc1, double bc2_sqrt) {
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first[k + 1]) ++k;
    const int nb = t.first[k + 1] - t.first[k];
    double* __restrict__ param = t.param[k];
    const double* __restrict__ grad = t.grad[k];
    double* __restrict__ m = t.m[k];
    double* __restrict__ v = t.v[k];
    const int64_t n = t.n[k];
    for (int64_t i = (int64_t)(blockIdx.x - t.first[k]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        const double g = grad[i];
        const double mi = b1 * m[i] + (1.0 - b1) * g;
        const double vi = b2 * v[i] + (1.0 - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        const double denom = sqrt(vi) / bc2_sqrt + eps;
        param[i] -= (lr / bc1) * (mi / denom);
    }
}
extern "C" int dsvgp_adam_step_multi_f64(dsvgp_ctx* ctx, int count, double* const* params, const double* const* grads,
                                         double* const* exp_avgs, double* const* exp_avg_sqs, const int64_t* sizes, double lr,
                                         double beta1, double beta2, double eps, int step) {
    if (!ctx || count < 0 || count > DSVGP_ADAM_MAX_TENSORS || step < 1) return DSVGP_EINVAL;
    if (count && (!params || !grads || !exp_avgs || !exp_avg_sqs || !sizes)) return DSVGP_EINVAL;
    AdamTable64 t{};
    int nblocks = 0;
    for (int k = 0; k < count; ++k) {
        if (sizes[k] < 0 || (sizes[k] > 0 && (!params[k] || !grads[k] || !exp_avgs[k] || !exp_avg_sqs[k]))) return DSVGP_EINVAL;
        if (sizes[k] == 0) continue;
        const int c = t.count++;
        t.param[c] = params[k]; t.grad[c] = grads[k]; t.m[c] = exp_avgs[k]; t.v[c] = exp_avg_sqs[k]; t.n[c] = sizes[k];
        int b = cdiv(sizes[k], 256);
        if (b > 2048) b = 2048;
        t.first[c] = nblocks;
        nblocks += b;
    }
    if (!t.count) return 0;
    t.first[t.count] = nblocks;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(adam_multi_f64_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, t, lr, beta1, beta2, eps, bc1, sqrt(bc2));
    DSVGP_LAUNCH_CHECK();
    return 0;
}

static int adam_step_multi_impl(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads, float* const* exp_avgs,
                                float* const* exp_avg_sqs, const int64_t* sizes, float lr, float beta1, float beta2, float eps, int step,
                                const int* guard_dev);
extern "C" int dsvgp_adam_step_multi(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads,
                                     float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* sizes, float lr,
                                     float beta1, float beta2, float eps, int step) {
    return adam_step_multi_impl(ctx, count, params, grads, exp_avgs, exp_avg_sqs, sizes, lr, beta1, beta2, eps, step, nullptr);
}
// the same update skipped ON THE DEVICE while *guard_dev != 0 (round 6): guard_dev = the status word of the one-call step that produced the
// gradients (dsvgp_elbo_step_locate which = 5) -- the host then need not wait for that status before it queues the update
extern "C" int dsvgp_adam_step_multi_guarded(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads,
                                             float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* sizes, float lr,
                                             float beta1, float beta2, float eps, int step, const int* guard_dev) {
    return adam_step_multi_impl(ctx, count, params, grads, exp_avgs, exp_avg_sqs, sizes, lr, beta1, beta2, eps, step, guard_dev);
}
static int adam_step_multi_impl(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads, float* const* exp_avgs,
                                float* const* exp_avg_sqs, const int64_t* sizes, float lr, float beta1, float beta2, float eps, int step,
                                const int* guard_dev) {
    if (!ctx || count < 0 || count > DSVGP_ADAM_MAX_TENSORS || step < 1) return DSVGP_EINVAL;
    if (count && (!params || !grads || !exp_avgs || !exp_avg_sqs || !sizes)) return DSVGP_EINVAL;
    AdamTable t{};
    int nblocks = 0;
    for (int k = 0; k < count; ++k) {
        if (sizes[k] != 0 && (!params[k] || !grads[k] || !exp_avgs[k] || !exp_avg_sqs[k])) return DSVGP_EINVAL;
        if (sizes[k] == 0) continue;
        const int c = t.count++;
        t.param[c] = params[k]; t.grad[c] = grads[k]; t.m[c] = exp_avgs[k]; t.v[c] = exp_avg_sqs[k]; t.n[c] = sizes[k];
        // (sizes[k] = -n: an n x n matrix whose lower triangle is the parameter: one workgroup per row, cyclically)
        int b = sizes[k] < 0 ? (int)(-sizes[k] < 2048 ? -sizes[k] : 2048) : cdiv(sizes[k], 256);
        if (b > 2048) b = 2048;
        t.first[c] = nblocks;
        nblocks += b;
    }
    if (!t.count) return 0;
    t.first[t.count] = nblocks;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, t, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), guard_dev);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// ---- fused launches of the one-call step (csrc/step.hip) ---------------------------------------------
int launch_stats_residual(dsvgp_ctx* ctx, const float* A, int64_t lda, int Mp, int ncols, int p, const float* m,
                          const float* constant, const float* hyp, float* mu, float* var, void* workspace, const float* y,
                          double global_rows, float* mu_bar, float* sums) {
    if (ctx->det_slab || ncols == 0) {          // deterministic mode keeps its per-workgroup partials: the two public calls
        int rc = dsvgp_predictive_stats(ctx, A, lda, A, lda, Mp, ncols, p, m, constant, hyp, mu, var, workspace);
        if (rc) return rc;
        return dsvgp_residual_terms(ctx, mu, y, ncols, hyp, global_rows, mu_bar, sums);
    }
    if (!ctx->prezeroed) {
        hipError_t e = hipMemsetAsync(sums, 0, 4 * sizeof(float), ctx->stream);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    const int nch = stats_chunks(Mp), rpc = cdiv(Mp, nch);
    hipLaunchKernelGGL(colstats_kernel, dim3(cdiv(ncols, 256), nch), dim3(256), 0, ctx->stream, A, lda, A, lda, Mp, ncols, m, rpc,
                       (float*)workspace);
    DSVGP_LAUNCH_CHECK();
    int blocks = cdiv(ncols, 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(colstats_residual_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)workspace, nch, ncols, p,
                       constant, hyp, mu, var, y, (float)(1.0 / global_rows), mu_bar, sums);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
int launch_scale_epilogue(dsvgp_ctx* ctx, float* x0, int64_t n0, float* x1, int64_t n1, const float* hyp, double rows,
                          const float* scal, const float* kl0, double num_data, const float* rl, const float* rs, const float* rn,
                          float* dh, float* drl, float* drs, float* drn, float* dconst, float* loss) {
    const int64_t tot = n0 + n1;
    int blocks = (int)((tot + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_epilogue_kernel, dim3(blocks), dim3(256), 0, ctx->stream, x0, n0, x1, n1, hyp, (float)(1.0 / rows), scal,
                       kl0, (float)(1.0 / num_data), rl, rs, rn, dh, drl, drs, drn, dconst, loss);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// ---- ELBO fast path entry points ---------------------------------------------------------------------
extern "C" int dsvgp_residual_terms(dsvgp_ctx* ctx, const float* mu, const float* y, int ncols, const float* hyp,
                                    double global_rows, float* mu_bar, float* sums) {
    if (!ctx || !mu || !y || !hyp || !mu_bar || !sums || ncols < 0 || global_rows <= 0) return DSVGP_EINVAL;
    if (!ctx->prezeroed) {
        hipError_t e = hipMemsetAsync(sums, 0, 4 * sizeof(float), ctx->stream);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    if (ncols == 0) return 0;
    int blocks = cdiv(ncols, 256);
    if (blocks > 512) blocks = 512;
    float* partials = (ctx->det_slab && ctx->det_bytes >= (size_t)blocks * 8 * sizeof(float)) ? (float*)ctx->det_slab : nullptr;
    hipLaunchKernelGGL(residual_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mu, y, ncols, hyp,
                       (float)(1.0 / global_rows), mu_bar, sums, partials);
    DSVGP_LAUNCH_CHECK();
    if (partials) {
        hipLaunchKernelGGL(partials_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float*)partials, blocks, 2, sums);
        DSVGP_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int dsvgp_trace_terms(dsvgp_ctx* ctx, const float* LS, int64_t ldls, const float* T1, int64_t ldt,
                                 const float* G, int64_t ldg, int n, float t1_scale, float* sums) {
    if (!ctx || !LS || !T1 || !G || !sums || n <= 0) return DSVGP_EINVAL;
    hipLaunchKernelGGL(trace_kernel, dim3(n < 1024 ? n : 1024), dim3(256), 0, ctx->stream, LS, ldls, T1, ldt, G, ldg, n, t1_scale, sums);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_elbo_fast_finalize(dsvgp_ctx* ctx, const float* sums, const float* hyp, int npts, int p,
                                        double global_rows, float* out_scalars) {
    if (!ctx || !sums || !hyp || !out_scalars || npts < 0 || p < 0 || global_rows <= 0) return DSVGP_EINVAL;
    hipLaunchKernelGGL(elbo_fast_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream, sums, hyp, npts, p,
                       (float)(1.0 / global_rows), out_scalars);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_mirror_lower_f32(dsvgp_ctx* ctx, float* G, int n, int64_t ldg) {
    if (!ctx || !G || n <= 0 || ldg < n) return DSVGP_EINVAL;
    const int nb = cdiv(n, 32);
    hipLaunchKernelGGL(mirror_lower_f32_kernel, dim3(nb, nb), dim3(32, 8), 0, ctx->stream, G, n, ldg);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_add_diag_f32(dsvgp_ctx* ctx, float* A, int n, int64_t lda, float delta) {
    if (!ctx || !A || n <= 0) return DSVGP_EINVAL;
    hipLaunchKernelGGL(add_diag_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, A, n, lda, delta);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_sminus_i_col(dsvgp_ctx* ctx, float* A, int n, int64_t lda, const float* m, const float* hyp, double rows) {
    if (!ctx || !A || !m || !hyp || n <= 0 || lda < n + 1) return DSVGP_EINVAL;
    hipLaunchKernelGGL(sminus_i_col_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, A, n, lda, m, hyp, (float)rows);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// mirror the lower triangle of A[n, n] onto its upper triangle AND A[i][i] -= 1, A[i][n] = m[i] noise rows: dsvgp_mirror_lower_f32
// + dsvgp_sminus_i_col in one launch (the diagonal blocks do the extra work after their mirror)
// W (optional, round 6): the fp64 copy [S - I ; m^T noise rows], (n + 1) x n with row stride ldw, written on the way (S - I is symmetric:
// its rows go as they lie, the extra column becomes row n) -- two widening launches fewer per step
__global__ void mirror_sminus_kernel(float* __restrict__ G, int n, int64_t ldg, const float* __restrict__ m,
                                     const float* __restrict__ hyp, float rows, double* __restrict__ W, int64_t ldw) {
    __shared__ float tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int r = ty; r < 32; r += 8) {
        const int gi = bj * 32 + r, gj = bi * 32 + tx;
        const float v = (gi < n && gj < n) ? G[(int64_t)gi * ldg + gj] : 0.f;
        tile[r][tx] = v;
        if (W && gi < n && gj < n && gj <= gi) W[(int64_t)gi * ldw + gj] = (double)(gi == gj ? v - 1.f : v);     // the lower tile as it lies
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int gi = bi * 32 + r, gj = bj * 32 + tx;
        if (gi < n && gj < n && gj > gi) {
            G[(int64_t)gi * ldg + gj] = tile[tx][r];
            if (W) W[(int64_t)gi * ldw + gj] = (double)tile[tx][r];
        }
    }
    if (bi == bj && ty == 0) {
        const int i = bi * 32 + tx;
        if (i < n) {
            const float col = m[i] * (hyp[2] * rows);
            G[(int64_t)i * ldg + i] -= 1.f;
            G[(int64_t)i * ldg + n] = col;
            if (W) W[(int64_t)n * ldw + i] = (double)col;
        }
    }
}
int launch_widen_sym_f32_f64(hipStream_t st, const float* src, int64_t lds, double* dst, int64_t ldd, int n) {
    if (n <= 0) return 0;
    const int nb = cdiv(n, 32);
    hipLaunchKernelGGL(widen_sym_kernel, dim3(nb, nb), dim3(32, 8), 0, st, src, lds, dst, ldd, n);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

int launch_mirror_sminus_i_col(hipStream_t st, float* A, int n, int64_t lda, const float* m, const float* hyp, float rows, double* W,
                               int64_t ldw) {
    const int nb = cdiv(n, 32);
    hipLaunchKernelGGL(mirror_sminus_kernel, dim3(nb, nb), dim3(32, 8), 0, st, A, n, lda, m, hyp, rows, W, ldw);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_tril_pack_f32(dsvgp_ctx* ctx, const float* src, int64_t ld, int n, const float* extra, int nextra,
                                   float* dst) {
    if (!ctx || n < 0 || nextra < 0 || (n > 0 && (!src || ld < n)) || (nextra > 0 && !extra)) return DSVGP_EINVAL;
    if (n == 0 && nextra == 0) return 0;
    if (!dst) return DSVGP_EINVAL;
    const int xb = nextra > 0 ? (cdiv(nextra, 256 * 8) < 64 ? cdiv(nextra, 256 * 8) : 64) : 0;
    hipLaunchKernelGGL(tril_pack_f32_kernel, dim3(n + xb), dim3(256), 0, ctx->stream, src, ld, n, extra, nextra, dst);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_tril_unpack_f32(dsvgp_ctx* ctx, const float* src, int n, float* dst, int64_t ld, float* extra,
                                     int nextra) {
    if (!ctx || n < 0 || nextra < 0 || (n > 0 && (!dst || ld < n)) || (nextra > 0 && !extra)) return DSVGP_EINVAL;
    if (n == 0 && nextra == 0) return 0;
    if (!src) return DSVGP_EINVAL;
    const int xb = nextra > 0 ? (cdiv(nextra, 256 * 8) < 64 ? cdiv(nextra, 256 * 8) : 64) : 0;
    hipLaunchKernelGGL(tril_unpack_f32_kernel, dim3(n + xb), dim3(256), 0, ctx->stream, src, n, dst, ld, extra, nextra);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

// ---- graph-capturable entry points (per-step scalars in device memory) ---------------------------------
extern "C" int dsvgp_adam_step_multi_dev(dsvgp_ctx* ctx, int count, float* const* params, const float* const* grads,
                                         float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* sizes,
                                         const float* lr_step_dev, float beta1, float beta2, float eps, const int* guard_dev) {
    if (!ctx || count < 0 || count > DSVGP_ADAM_MAX_TENSORS || !lr_step_dev) return DSVGP_EINVAL;
    if (count && (!params || !grads || !exp_avgs || !exp_avg_sqs || !sizes)) return DSVGP_EINVAL;
    AdamTable t{};
    int nblocks = 0;
    for (int k = 0; k < count; ++k) {
        if (sizes[k] < 0 || (sizes[k] > 0 && (!params[k] || !grads[k] || !exp_avgs[k] || !exp_avg_sqs[k]))) return DSVGP_EINVAL;
        if (sizes[k] == 0) continue;
        const int c = t.count++;
        t.param[c] = params[k]; t.grad[c] = grads[k]; t.m[c] = exp_avgs[k]; t.v[c] = exp_avg_sqs[k]; t.n[c] = sizes[k];
        int b = cdiv(sizes[k], 256);
        if (b > 2048) b = 2048;
        t.first[c] = nblocks;
        nblocks += b;
    }
    if (!t.count) return 0;
    t.first[t.count] = nblocks;
    hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, t, lr_step_dev, beta1, beta2, eps,
                       guard_dev);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_scale_by_vbar(dsvgp_ctx* ctx, float* x0, int64_t n0, float* x1, int64_t n1, float* x2, int64_t n2,
                                   const float* hyp, double global_rows) {
    if (!ctx || !hyp || n0 < 0 || n1 < 0 || n2 < 0 || global_rows <= 0) return DSVGP_EINVAL;
    if ((n0 && !x0) || (n1 && !x1) || (n2 && !x2)) return DSVGP_EINVAL;
    const int64_t tot = n0 + n1 + n2;
    if (tot == 0) return 0;
    int blocks = cdiv(tot, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale3_kernel, dim3(blocks), dim3(256), 0, ctx->stream, x0, n0, x1, n1, x2, n2, hyp,
                       (float)(1.0 / global_rows));
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_kl_terms_scaled(dsvgp_ctx* ctx, const float* m, const float* LS, int64_t ldls, int Mp, double num_data,
                                     int add_kl, const float* hyp, double global_rows, float* kl_out, float* d_m, float* d_LS,
                                     int64_t lddls) {
    if (!ctx || !m || !LS || !kl_out || !d_m || !d_LS || !hyp || Mp <= 0 || num_data <= 0 || global_rows <= 0) return DSVGP_EINVAL;
    int rc = launch_ls_rows(ctx->stream, m, LS, ldls, Mp, (float)(1.0 / num_data), 1 | (add_kl ? 2 : 0), hyp, (float)(1.0 / global_rows),
                            kl_out + 1, nullptr, d_m, d_LS, lddls);
    if (rc) return rc;
    hipLaunchKernelGGL(ls_rows_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)(kl_out + 1), (const float*)nullptr,
                       (const float*)nullptr, (int64_t)0, Mp, 0.f, kl_out, (float*)nullptr, (const float*)nullptr, 0, 0, 0.f, (float*)nullptr);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
// The variational block of the ELBO fast path in ONE pass over (L_S, T = tril(G L_S)) -- trace terms, optional scaling by
// 2 vbar = 1 / (noise rows), optional KL value + gradient (flags as ls_rows_kernel: 1 scale, 2 KL; the trace terms always):
//   sums[2] = t1_scale * sum_{i>=j} L_S,ij T_ij (= |L_S^T A|_F^2), sums[3] = trace(G), kl_out[0] = KL (0 without the KL flag),
//   d_LS <- [scale] T + dKL/dL_S / num_data, d_m += m / num_data.   kl_out: 1 + 2 Mp floats (value + per-row scratch).
// No atomics: per-row partials + a fixed-order reduction (bitwise reproducible).
int launch_variational_terms(hipStream_t st, const float* m, const float* LS, int64_t ldls, int Mp, double num_data, int flags,
                             const float* hyp, double global_rows, const float* G, int64_t ldg, float t1_scale, float* kl_out,
                             float* sums, const float* dm_src, float* d_m, float* d_LS, int64_t lddls, int fin_npts, int fin_p,
                             float* fin_scal) {
    // flags bit 3 (8): mute the scalar outputs (trace terms, KL value), see ls_rows_reduce_kernel
    int rc = launch_ls_rows(st, m, LS, ldls, Mp, (float)(1.0 / num_data), (flags & 3) | 4, hyp, (float)(1.0 / global_rows),
                            kl_out + 1, kl_out + 1 + Mp, d_m, d_LS, lddls, dm_src);
    if (rc) return rc;
    hipLaunchKernelGGL(ls_rows_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)(kl_out + 1),
                       (const float*)(kl_out + 1 + Mp), G, ldg, Mp, t1_scale, kl_out, sums, hyp, fin_npts, fin_p,
                       (float)(1.0 / global_rows), fin_scal, (flags & 8) ? 1 : 0);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_variational_terms(dsvgp_ctx* ctx, const float* m, const float* LS, int64_t ldls, int Mp, double num_data,
                                       int flags, const float* hyp, double global_rows, const float* G, int64_t ldg,
                                       float t1_scale, float* kl_out, float* sums, float* d_m, float* d_LS, int64_t lddls) {
    if (!ctx || !m || !LS || !kl_out || !sums || !d_m || !d_LS || !G || Mp <= 0 || num_data <= 0 || global_rows <= 0 ||
        ((flags & 1) && !hyp) || (flags & ~3))
        return DSVGP_EINVAL;
    return launch_variational_terms(ctx->stream, m, LS, ldls, Mp, num_data, flags, hyp, global_rows, G, ldg, t1_scale, kl_out, sums,
                                    nullptr, d_m, d_LS, lddls, 0, 0, nullptr);
}
