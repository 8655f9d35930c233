// fp64 model mode of the block-kernel assembly (reference experiments run with torch.set_default_dtype(torch.float64),
// experiments/synthetic/exp_script.py:56): RBFKernelDirectionalGrad.forward (directionalvi/RBFKernelDirectionalGrad.py:41-119)
// and its backward in double precision.
//
// Same formulation as assemble.hip -- pack every point as the (p+1) rows [(x - c)/ell ; v_1 ; ... ; v_p] (unit directions),
// T = P1 P2^T is already laid out like the interleaved kernel matrix and holds every inner product the four block types
// need -- but the heavy contractions are left to the fp64 MFMA GEMM (dsvgp_gemm): T = P1 P2^T and dP1 = Tbar P2, and only the
// per-pair micro-block transforms live here, as plain one-thread-per-point-pair kernels working IN PLACE on the T buffer:
//   forward   T -> K:   |r|^2 = nrm1 + nrm2 - 2 T00, k = s exp(-|r|^2/2), u_a = alpha_a - T_a0, w_b = T_0b - beta_b,
//                       K00 = k, K0b = w_b k/ell, Ka0 = -u_a k/ell, Kab = (T_ab - u_a w_b) k/ell^2
//   backward  (Gbar, T) -> Tbar, d outputscale, d lengthscale (SURVEY.md Appendix A in the T formulation):
//                       q = dL/dk, Tbar_00 = k q, Tbar_0b = wbar_b, Tbar_a0 = -ubar_a, Tbar_ab = k Gbar_ab / ell^2
//   points    dP1 = Tbar [P2 | indicator] -> d_x1, d_v1 through x/ell and the direction normalisation (the indicator
//             column collects the row sums that carry the self-term gradients nrm-bar, alpha-bar).
// Not a throughput path (fp64 mode is the reference's experiment setting, not the benchmark): clarity over tiling.
#include "common.h"

namespace {

constexpr int PMAX = 16;                      // directions per point supported by the per-thread register arrays

__global__ void pack64_kernel(const double* __restrict__ x, const double* __restrict__ v, int n, int d, int p,
                              const double* __restrict__ hyp, const double* __restrict__ center, double* __restrict__ P,
                              double* __restrict__ self, double* __restrict__ vnorm, int K4, int DP) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = p + 1;
    if (row >= n * q) return;
    const int i = row / q, a = row - i * q;
    const double ell = hyp[0];
    double* Pr = P + (int64_t)row * DP;
    const double* xi = x + (int64_t)i * d;
    for (int k = 0; k < DP; ++k) Pr[k] = 0.0;
    if (a == 0) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) {
            const double xt = (xi[k] - (center ? center[k] : 0.0)) / ell;    // x.div(lengthscale), :67-68
            Pr[k] = xt;
            acc = fma(xt, xt, acc);
        }
        Pr[K4] = 1.0;                                   // indicator column (row sums in the backward)
        self[row] = acc;
    } else {
        const double* vi = v + ((int64_t)i * p + (a - 1)) * d;
        double ss = 0.0;
        for (int k = 0; k < d; ++k) ss = fma(vi[k], vi[k], ss);
        const double nrm = sqrt(ss);                    // :57-58
        double acc = 0.0;
        for (int k = 0; k < d; ++k) {
            const double vh = vi[k] / nrm;
            Pr[k] = vh;
            acc = fma(vh, (xi[k] - (center ? center[k] : 0.0)) / ell, acc);
        }
        self[row] = acc;
        vnorm[(int64_t)i * p + (a - 1)] = nrm;
    }
}

// T -> K in place; one thread per point pair (consecutive threads: consecutive pairs of one point row)
__global__ __launch_bounds__(256) void fwd_transform64_kernel(double* __restrict__ T, int64_t ld, const double* __restrict__ self1,
                                                              int n1, const double* __restrict__ self2, int n2, int p,
                                                              const double* __restrict__ hyp, double jitter) {
    const int64_t pid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pid >= (int64_t)n1 * n2) return;
    const int i = (int)(pid / n2), j = (int)(pid - (int64_t)i * n2);
    const int q = p + 1;
    const double ell = hyp[0], s = hyp[1], il = 1.0 / ell, il2 = il * il;
    double* blk = T + (int64_t)i * q * ld + (int64_t)j * q;
    const double* s1 = self1 + (int64_t)i * q;
    const double* s2 = self2 + (int64_t)j * q;
    const double nn = fmax(s1[0] + s2[0] - 2.0 * blk[0], 0.0);          // covar_dist clamps at 0
    const double k = s * exp(-0.5 * nn);                                // postprocess_rbf, ScaleKernel
    const bool diag = jitter != 0.0 && i == j;
    double w[PMAX];
    for (int b = 1; b < q; ++b) w[b - 1] = blk[b] - s2[b];              // r . v2_b
    for (int a = 1; a < q; ++a) {
        double* row = blk + (int64_t)a * ld;
        const double u = s1[a] - row[0];                                // r . v1_a
        for (int b = 1; b < q; ++b) row[b] = (row[b] - u * w[b - 1]) * k * il2 + ((diag && a == b) ? jitter : 0.0);
        row[0] = -u * k * il;
    }
    for (int b = 1; b < q; ++b) blk[b] = w[b - 1] * k * il;
    blk[0] = k + (diag ? jitter : 0.0);
}

// (Gbar, T) -> Tbar in place of T; block partial sums of <Gbar, K>/s and of the lengthscale gradient go to d_hyp by atomics
__global__ __launch_bounds__(256) void bwd_transform64_kernel(const double* __restrict__ G, int64_t ldg, double* __restrict__ T,
                                                              int64_t ldt, const double* __restrict__ self1, int n1,
                                                              const double* __restrict__ self2, int n2, int p,
                                                              const double* __restrict__ hyp, double* __restrict__ d_hyp) {
    __shared__ double red[2][4];
    const int64_t pid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double ds = 0.0, dl = 0.0;
    if (pid < (int64_t)n1 * n2) {
        const int i = (int)(pid / n2), j = (int)(pid - (int64_t)i * n2);
        const int q = p + 1;
        const double ell = hyp[0], s = hyp[1], il = 1.0 / ell, il2 = il * il;
        double* tb = T + (int64_t)i * q * ldt + (int64_t)j * q;
        const double* gb = G + (int64_t)i * q * ldg + (int64_t)j * q;
        const double* s1 = self1 + (int64_t)i * q;
        const double* s2 = self2 + (int64_t)j * q;
        const double nn = fmax(s1[0] + s2[0] - 2.0 * tb[0], 0.0);
        const double k = s * exp(-0.5 * nn);
        double w[PMAX], wbar[PMAX];
        double qq = gb[0];                      // dL/dk
        double gk = gb[0] * k;                  // <Gbar, K> of the block
        double first = 0.0, second = 0.0, hess = 0.0, dots = 0.0;
        for (int b = 1; b < q; ++b) {
            w[b - 1] = tb[b] - s2[b];
            wbar[b - 1] = 0.0;
            first = fma(gb[b], w[b - 1], first);                        // sum_b G0b w_b
        }
        for (int a = 1; a < q; ++a) {
            double* trow = tb + (int64_t)a * ldt;
            const double* grow = gb + (int64_t)a * ldg;
            const double u = s1[a] - trow[0];
            const double ga0 = grow[0];
            double gw = 0.0, gt = 0.0;
            for (int b = 1; b < q; ++b) {
                const double gab = grow[b];
                gw = fma(gab, w[b - 1], gw);
                gt = fma(gab, trow[b], gt);
                wbar[b - 1] = fma(gab, u, wbar[b - 1]);                 // sum_a Gab u_a
                trow[b] = k * il2 * gab;                                // Tbar_ab
            }
            second = fma(ga0, u, second);                               // sum_a Ga0 u_a
            hess += gt - u * gw;                                        // sum_ab Gab (T_ab - u_a w_b)
            const double ubar = k * (-ga0 * il - gw * il2);
            trow[0] = -ubar;                                            // Tbar_a0
            dots = fma(ubar, u, dots);
        }
        qq += first * il - second * il + hess * il2;
        for (int b = 1; b < q; ++b) {
            const double wb = k * (gb[b] * il - wbar[b - 1] * il2);     // wbar_b
            dots = fma(wb, w[b - 1], dots);
            tb[b] = wb;                                                 // Tbar_0b
        }
        tb[0] = k * qq;                                                 // Tbar_00 = -2 nn-bar
        gk += k * (first * il - second * il + hess * il2);              // <Gbar, K> = k q
        ds = gk / s;
        // d ell: -(rbar . r)/ell - (sum G0b K0b + sum Ga0 Ka0)/ell - 2 sum Gab Kab / ell,  rbar . r = -k q |r|^2 + dots
        dl = -(-k * qq * nn + dots) * il - k * (first * il - second * il) * il - 2.0 * k * hess * il2 * il;
    }
    for (int off = 32; off > 0; off >>= 1) { ds += __shfl_down(ds, off); dl += __shfl_down(dl, off); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = ds; red[1][wave] = dl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&d_hyp[1], red[0][0] + red[0][1] + red[0][2] + red[0][3]);     // d outputscale
        atomicAdd(&d_hyp[0], red[1][0] + red[1][1] + red[1][2] + red[1][3]);     // d lengthscale
    }
}

// one 64-thread block per point: dP (its q rows, DP columns) -> d_x1, d_v1 (through x/ell and the direction normalisation)
__global__ __launch_bounds__(64) void bwd_points64_kernel(const double* __restrict__ dP, const double* __restrict__ P1,
                                                          const double* __restrict__ vnorm1, int n1, int d, int p, int K4,
                                                          int DP, const double* __restrict__ hyp, double sym,
                                                          double* __restrict__ d_x1, double* __restrict__ d_v1) {
    extern __shared__ double dots64[];       // [q]
    const int i = blockIdx.x, t = threadIdx.x;
    const int q = p + 1;
    const double ell = hyp[0];
    const double* dPi = dP + (int64_t)i * q * DP;
    const double* xt = P1 + (int64_t)i * q * DP;
    // vhat-bar_a = dP[a,:] + alphabar_a x~ ; dots[a] = vhat_a . vhat-bar_a ; alphabar_a = -dP[a,K4]
    for (int a = 1 + t; a <= p; a += 64) {
        const double* vh = P1 + ((int64_t)i * q + a) * DP;
        const double ab = -dPi[a * DP + K4];
        double dot = 0.0;
        for (int k = 0; k < d; ++k) dot += vh[k] * (dPi[a * DP + k] + ab * xt[k]);
        dots64[a] = dot;
    }
    __syncthreads();
    const double nbar = -0.5 * dPi[K4];
    for (int k = t; k < d; k += 64) {
        // x~bar = dP[0,:] + 2 nbar x~ + sum_a alphabar_a vhat_a
        double xb = dPi[k] + 2.0 * nbar * xt[k];
        for (int a = 1; a <= p; ++a) xb += -dPi[a * DP + K4] * P1[((int64_t)i * q + a) * DP + k];
        d_x1[(int64_t)i * d + k] += sym * xb / ell;
    }
    for (int e = t; e < p * d; e += 64) {
        const int a = 1 + e / d, k = e - (a - 1) * d;
        const double* vh = P1 + ((int64_t)i * q + a) * DP;
        const double vb = dPi[a * DP + k] - dPi[a * DP + K4] * xt[k];
        const double inv = 1.0 / vnorm1[(int64_t)i * p + (a - 1)];
        d_v1[((int64_t)i * p + (a - 1)) * d + k] += sym * (vb - vh[k] * dots64[a]) * inv;     // normalisation Jacobian
    }
}

// mu_j = sum_i A[i,j] m[i] (+ constant added by the caller), cs_j = sum_i (W[i,j]^2 - A[i,j]^2): the two column reductions of
// DGVS.py:188,192-205 over the fp64 interpolation matrices.  64 columns x 4 row lanes per block, row range split over
// gridDim.y, fp64 atomics into zeroed outputs.
__global__ __launch_bounds__(256) void colstats64_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ W,
                                                         int64_t ldw, const double* __restrict__ m, int Mp, int Bp,
                                                         double* __restrict__ mu, double* __restrict__ cs) {
    __shared__ double red[2][4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    const int per = (Mp + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(Mp, r0 + per);
    double am = 0.0, sq = 0.0;
    if (j < Bp)
        for (int i = r0 + g; i < r1; i += 4) {
            const double a = A[(int64_t)i * lda + j];
            am = fma(a, m[i], am);
            if (W) {
                const double w = W[(int64_t)i * ldw + j];
                sq += w * w - a * a;
            }
        }
    red[0][g][c] = am;
    red[1][g][c] = sq;
    __syncthreads();
    if (g == 0 && j < Bp) {
        atomicAdd(mu + j, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        if (W) atomicAdd(cs + j, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

// Abar = m mu_bar^T + 2 (U - A) diag(var_bar)  and  Av = 2 A diag(var_bar)   (the backward of mu = A^T m, var = colsum(W^2 - A^2))
__global__ __launch_bounds__(256) void abar64_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ U,
                                                     int64_t ldu, const double* __restrict__ m, const double* __restrict__ mu_bar,
                                                     const double* __restrict__ var_bar, int Mp, int Bp, double* __restrict__ Abar,
                                                     int64_t ldo, double* __restrict__ Av, int64_t ldv) {
    const int j = blockIdx.y * 256 + threadIdx.x;
    const int i = blockIdx.x;
    if (j >= Bp) return;
    const double a = A[(int64_t)i * lda + j];
    const double vb2 = 2.0 * var_bar[j];
    const double u = U ? U[(int64_t)i * ldu + j] : a;
    Abar[(int64_t)i * ldo + j] = fma(m[i], mu_bar[j], (u - a) * vb2);
    if (Av) Av[(int64_t)i * ldv + j] = a * vb2;
}

// GaussianLikelihood + VariationalELBO / PredictiveLogLikelihood terms of the fp64 model (directional_vi.py:245-246, gpytorch
// expected_log_prob / log_marginal): per output j  mu = mu0 + c,  var = s dg_j + 1e-4 + cs_j,  vn = max(var + noise, 1e-6);
// writes mu, vn, mu_bar = d loss / d mu0, var_bar = d loss / d cs and accumulates
// scal: 0 sum_ll, 1 d loss / d noise, 2 d / d constant, 3 d / d outputscale (prior diagonal), 4 d / d lengthscale (prior diagonal)
__global__ __launch_bounds__(256) void likelihood64_kernel(const double* __restrict__ mu0, const double* __restrict__ cs,
                                                           const double* __restrict__ y, const double* __restrict__ constant,
                                                           int ncols, int p, const double* __restrict__ hyp, int mll_type,
                                                           double inv_rows, double* __restrict__ mu_out,
                                                           double* __restrict__ varn_out, double* __restrict__ mu_bar,
                                                           double* __restrict__ var_bar, double* __restrict__ scal) {
    __shared__ double red[5][4];
    const double ell = hyp[0], s = hyp[1], noise = hyp[2], c = constant[0];
    const double LOG2PI = 1.8378770664093454835606594728112;
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int j = blockIdx.x * 256 + threadIdx.x; j < ncols; j += gridDim.x * 256) {
        const bool isf = (j % (p + 1)) == 0;
        const double mj = mu0[j] + c, r = y[j] - mj;
        const double var = s * (isf ? 1.0 : 1.0 / (ell * ell)) + 1e-4 + cs[j];
        const double vraw = var + noise;
        const bool clamped = vraw < 1e-6;
        const double vn = clamped ? 1e-6 : vraw;
        double ll, dmu, dvn, dnoise;
        if (mll_type == 0) {
            ll = -0.5 * ((r * r + vn) / noise + log(noise) + LOG2PI);
            dmu = r / noise;
            dvn = -0.5 / noise;
            dnoise = 0.5 * (r * r + vn) / (noise * noise) - 0.5 / noise;
        } else {
            const double tot = fmax(vn + noise, 1e-8);
            ll = -0.5 * (r * r / tot + log(tot) + LOG2PI);
            dmu = r / tot;
            dvn = (vn + noise < 1e-8) ? 0.0 : 0.5 * (r * r / (tot * tot) - 1.0 / tot);
            dnoise = dvn;
        }
        const double dvar = clamped ? 0.0 : dvn;
        dnoise += dvar;
        const double mb = -dmu * inv_rows, vb = -dvar * inv_rows;
        mu_out[j] = mj;
        varn_out[j] = vn;
        mu_bar[j] = mb;
        var_bar[j] = vb;
        acc[0] += ll;
        acc[1] += -dnoise * inv_rows;
        acc[2] += mb;
        acc[3] += vb * (isf ? 1.0 : 1.0 / (ell * ell));
        acc[4] += isf ? 0.0 : vb * (-2.0 * s / (ell * ell * ell));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[q][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&scal[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// Scalar tail of the fp64 ELBO fast path (ElboEngine64._elbo_fast64): the variance enters the ELBO only through its sum, so
// per output just  mu = mu0 + c,  mu_bar = d loss / d mu0 = -(y - mu) / (noise rows)  and the sums of r^2 and r (scal[6..7]) ...
__global__ __launch_bounds__(256) void fast_tail64_kernel(const double* __restrict__ mu0, const double* __restrict__ y,
                                                          const double* __restrict__ constant, int ncols,
                                                          const double* __restrict__ hyp, double inv_rows,
                                                          double* __restrict__ mu_out, double* __restrict__ mu_bar,
                                                          double* __restrict__ scal) {
    __shared__ double red[2][4];
    const double noise = hyp[2], c = constant[0];
    const double k = -inv_rows / noise;
    double a0 = 0.0, a1 = 0.0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < ncols; j += gridDim.x * 256) {
        const double mj = mu0[j] + c, r = y[j] - mj;
        mu_out[j] = mj;
        mu_bar[j] = k * r;
        a0 += r * r;
        a1 += r;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); }
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) atomicAdd(&scal[6 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
// ... and the closed forms (one thread): sum_j var_j + noise = s npts (1 + pd / ell^2) + ncols (1e-4 + noise) + tvar,
// ll = -1/2 [(sum r^2 + that) / noise + ncols (log noise + log 2 pi)]  (expected_log_prob summed, directional_vi.py:245-246);
// scal = {ll, d loss / d noise, d / d constant, d / d outputscale, d / d lengthscale, vbar = d loss / d tvar, sum r^2, sum r}
__global__ void fast_tail64_final_kernel(const double* __restrict__ hyp, const double* __restrict__ tvar, int ncols, int npts,
                                         int pd, double inv_rows, double* __restrict__ scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ell = hyp[0], s = hyp[1], noise = hyp[2];
    const double LOG2PI = 1.8378770664093454835606594728112;
    const double SS = scal[6], SR = scal[7];
    const double dgsum = (double)npts * (1.0 + (double)pd / (ell * ell));
    const double sum_varn = s * dgsum + (double)ncols * (1e-4 + noise) + tvar[0];
    const double vbar = 0.5 * inv_rows / noise;
    scal[0] = -0.5 * ((SS + sum_varn) / noise + (double)ncols * (log(noise) + LOG2PI));
    scal[1] = -(0.5 * (SS + sum_varn) / (noise * noise) - (double)ncols / noise) * inv_rows;
    scal[2] = -SR * inv_rows / noise;
    scal[3] = vbar * dgsum;
    scal[4] = vbar * s * (double)npts * (double)pd * (-2.0 / (ell * ell * ell));
    scal[5] = vbar;
}

}  // namespace

extern "C" int dsvgp_elbo_fast_tail_f64(dsvgp_ctx* ctx, const double* mu0, const double* y, const double* constant, int ncols,
                                        int npts, int pd, const double* hyp, const double* tvar, double rows, double* mu,
                                        double* mu_bar, double* scal) {
    if (!ctx || !mu0 || !y || !constant || !hyp || !tvar || !mu || !mu_bar || !scal || ncols <= 0 || npts <= 0 || pd < 0 ||
        ncols != npts * (pd + 1) || rows <= 0.0)
        return DSVGP_EINVAL;
    hipError_t e = hipMemsetAsync(scal, 0, sizeof(double) * 8, ctx->stream);
    if (e != hipSuccess) return 1000 + (int)e;
    int blocks = cdiv(ncols, 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(fast_tail64_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mu0, y, constant, ncols, hyp, 1.0 / rows, mu,
                       mu_bar, scal);
    DSVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(fast_tail64_final_kernel, dim3(1), dim3(64), 0, ctx->stream, hyp, tvar, ncols, npts, pd, 1.0 / rows, scal);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_likelihood_terms_f64(dsvgp_ctx* ctx, const double* mu0, const double* cs, const double* y, const double* constant,
                                          int ncols, int p, const double* hyp, int mll_type, double rows, double* mu, double* varn,
                                          double* mu_bar, double* var_bar, double* scal) {
    if (!ctx || !mu0 || !cs || !y || !constant || !hyp || !mu || !varn || !mu_bar || !var_bar || !scal || ncols <= 0 || p < 0 ||
        rows <= 0.0 || (mll_type != 0 && mll_type != 1))
        return DSVGP_EINVAL;
    hipError_t e = hipMemsetAsync(scal, 0, sizeof(double) * 8, ctx->stream);
    if (e != hipSuccess) return 1000 + (int)e;
    int blocks = cdiv(ncols, 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(likelihood64_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mu0, cs, y, constant, ncols, p, hyp, mll_type,
                       1.0 / rows, mu, varn, mu_bar, var_bar, scal);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_pack_points_f64(dsvgp_ctx* ctx, const double* x, const double* v, int n, int d, int p, const double* hyp,
                                     const double* center, double* P, double* self, double* vnorm) {
    if (!ctx || !x || !hyp || !P || !self || n < 0 || d <= 0 || p < 0 || p > 16 || (p > 0 && (!v || !vnorm))) return DSVGP_EINVAL;
    if (n == 0) return 0;
    const int K4 = (d + 3) & ~3, DP = K4 + 4;
    const int rows = n * (p + 1);
    hipLaunchKernelGGL(pack64_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, ctx->stream, x, v, n, d, p, hyp, center, P, self,
                       vnorm, K4, DP);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_kernel_transform_f64(dsvgp_ctx* ctx, double* T, int64_t ld, const double* self1, int n1,
                                          const double* self2, int n2, int p, const double* hyp, double jitter) {
    if (!ctx || !T || !self1 || !self2 || !hyp || n1 < 0 || n2 < 0 || p < 0 || p > 16 || ld < (int64_t)n2 * (p + 1)) return DSVGP_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    hipLaunchKernelGGL(fwd_transform64_kernel, dim3(cdiv((int64_t)n1 * n2, 256)), dim3(256), 0, ctx->stream, T, ld, self1, n1,
                       self2, n2, p, hyp, jitter);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_kernel_bwd_transform_f64(dsvgp_ctx* ctx, const double* G, int64_t ldg, double* T, int64_t ldt,
                                              const double* self1, int n1, const double* self2, int n2, int p,
                                              const double* hyp, double* d_hyp) {
    if (!ctx || !G || !T || !self1 || !self2 || !hyp || !d_hyp || n1 < 0 || n2 < 0 || p < 0 || p > 16) return DSVGP_EINVAL;
    if (ldg < (int64_t)n2 * (p + 1) || ldt < (int64_t)n2 * (p + 1)) return DSVGP_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    hipLaunchKernelGGL(bwd_transform64_kernel, dim3(cdiv((int64_t)n1 * n2, 256)), dim3(256), 0, ctx->stream, G, ldg, T, ldt,
                       self1, n1, self2, n2, p, hyp, d_hyp);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_kernel_bwd_points_f64(dsvgp_ctx* ctx, const double* dP, const double* P1, const double* vnorm1, int n1,
                                           int d, int p, const double* hyp, int symmetric, double* d_x1, double* d_v1) {
    if (!ctx || !dP || !P1 || !hyp || !d_x1 || n1 < 0 || d <= 0 || p < 0 || p > 16 || (p > 0 && (!vnorm1 || !d_v1))) return DSVGP_EINVAL;
    if (n1 == 0) return 0;
    const int K4 = (d + 3) & ~3, DP = K4 + 4;
    hipLaunchKernelGGL(bwd_points64_kernel, dim3(n1), dim3(64), sizeof(double) * (p + 2), ctx->stream, dP, P1, vnorm1, n1, d, p,
                       K4, DP, hyp, symmetric ? 2.0 : 1.0, d_x1, d_v1);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_colstats_f64(dsvgp_ctx* ctx, const double* A, int64_t lda, const double* W, int64_t ldw, const double* m,
                                  int Mp, int Bp, double* mu, double* cs) {
    if (!ctx || !A || !m || !mu || (W && !cs) || Mp < 0 || Bp < 0 || lda < Bp || (W && ldw < Bp)) return DSVGP_EINVAL;
    if (Bp == 0) return 0;
    hipError_t e;
    if ((e = hipMemsetAsync(mu, 0, sizeof(double) * Bp, ctx->stream)) != hipSuccess) return 1000 + (int)e;
    if (W && (e = hipMemsetAsync(cs, 0, sizeof(double) * Bp, ctx->stream)) != hipSuccess) return 1000 + (int)e;
    if (Mp == 0) return 0;
    const int splits = max(1, min(64, Mp / 64));
    hipLaunchKernelGGL(colstats64_kernel, dim3(cdiv(Bp, 64), splits), dim3(256), 0, ctx->stream, A, lda, W, ldw, m, Mp, Bp, mu, cs);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
extern "C" int dsvgp_abar_f64(dsvgp_ctx* ctx, const double* A, int64_t lda, const double* U, int64_t ldu, const double* m,
                              const double* mu_bar, const double* var_bar, int Mp, int Bp, double* Abar, int64_t ldo, double* Av,
                              int64_t ldv) {
    if (!ctx || !A || !m || !mu_bar || !var_bar || !Abar || Mp < 0 || Bp < 0 || lda < Bp || ldo < Bp || (U && ldu < Bp) ||
        (Av && ldv < Bp))
        return DSVGP_EINVAL;
    if (Mp == 0 || Bp == 0) return 0;
    hipLaunchKernelGGL(abar64_kernel, dim3(Mp, cdiv(Bp, 256)), dim3(256), 0, ctx->stream, A, lda, U, ldu, m, mu_bar, var_bar, Mp, Bp,
                       Abar, ldo, Av, ldv);
    DSVGP_LAUNCH_CHECK();
    return 0;
}
