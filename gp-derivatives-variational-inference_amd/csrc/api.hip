// Context, rocSOLVER potrf, panel triangular solve and GEMM entry points of the C ABI.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "common.h"

extern "C" const char* dsvgp_version(void) { return "dsvgp-hip 0.1 (gfx950)"; }

extern "C" int dsvgp_create(dsvgp_ctx** out) {
    if (!out) return DSVGP_EINVAL;
    dsvgp_ctx* c = new dsvgp_ctx();
    rocblas_handle h = nullptr;
    rocblas_status st = rocblas_create_handle(&h);
    if (st != rocblas_status_success) { delete c; return 2000 + (int)st; }
    c->blas = h;
    *out = c;
    return 0;
}
extern "C" int dsvgp_destroy(dsvgp_ctx* ctx) {
    if (!ctx) return DSVGP_EINVAL;
    if (ctx->blas) rocblas_destroy_handle((rocblas_handle)ctx->blas);
    delete ctx;
    return 0;
}
// Deterministic mode: with a scratch buffer set, every split-K product stores its K slices to slabs in `scratch` and adds them in a
// fixed order (no floating-point atomics), the scalar reductions of the step go through per-workgroup partials: results are
// bitwise reproducible run to run.  scratch == NULL switches back to atomics.  The scratch is caller-owned, used by the launches
// queued on the context's CURRENT stream only (one stream at a time), and sized by the caller: a product whose slices do not
// fit uses fewer, longer slices (no split at all below two).
extern "C" int dsvgp_set_deterministic(dsvgp_ctx* ctx, void* scratch, size_t bytes) {
    if (!ctx || (scratch && bytes < 4096) || ((uintptr_t)scratch % 16)) return DSVGP_EINVAL;
    ctx->det_slab = scratch;
    ctx->det_bytes = scratch ? bytes : 0;
    return 0;
}
extern "C" int dsvgp_set_stream(dsvgp_ctx* ctx, void* stream) {
    if (!ctx) return DSVGP_EINVAL;
    ctx->stream = (hipStream_t)stream;
    rocblas_status st = rocblas_set_stream((rocblas_handle)ctx->blas, ctx->stream);
    return st == rocblas_status_success ? 0 : 2000 + (int)st;
}

// algo 0: rocSOLVER dpotrf.  Row-major lower Cholesky == column-major upper factorisation of the same
// buffer: dpotrf(upper) reads A(i,j), i<=j in column-major = the row-major lower triangle, and writes U with
// U_colmajor(i,j) = L_rowmajor(j,i).   algo 1: blocked Cholesky, one fused MFMA launch per block column (potrf.hip).
extern "C" size_t dsvgp_potrf_workspace_bytes(int n, int algo) {
    return (algo == 1 && n > 0) ? potrf_blocked_workspace_bytes(n) : 0;
}
extern "C" int dsvgp_potrf(dsvgp_ctx* ctx, double* A, int n, int64_t lda, int* info_dev, int algo, void* workspace) {
    if (!ctx || !A || !info_dev || n <= 0 || lda < n) return DSVGP_EINVAL;
    if (algo == 1) {
        if (!workspace) return DSVGP_EINVAL;
        return launch_potrf_blocked(ctx->stream, A, n, lda, info_dev, (double*)workspace, nullptr, 0, nullptr);
    }
    if (algo != 0) return DSVGP_EINVAL;
    rocblas_status st = rocsolver_dpotrf((rocblas_handle)ctx->blas, rocblas_fill_upper, n, A, (rocblas_int)lda, info_dev);
    return st == rocblas_status_success ? 0 : 2000 + (int)st;
}

extern "C" int dsvgp_widen_f32_f64(dsvgp_ctx* ctx, const float* src, int64_t ld, double* dst, int64_t ldd, int M, int N) {
    if (!ctx || !src || !dst || M <= 0 || N <= 0 || ld < N || ldd < N) return DSVGP_EINVAL;
    launch_widen_f32_f64(ctx->stream, src, ld, dst, ldd, M, N);
    DSVGP_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsvgp_gemm(dsvgp_ctx* ctx, int is_double, int flags, int M, int N, int K, double alpha, const void* A,
                          int64_t lda, const void* B, int64_t ldb, double beta, const void* Cin, int64_t ldcin, void* C,
                          int64_t ldc, float* C32, int64_t ldc32, const float* kscale) {
    if (!ctx || !A || !B || !C || M < 0 || N < 0 || K < 0) return DSVGP_EINVAL;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = A; g.B = B; g.Cin = Cin; g.C = C; g.C32 = C32; g.kscale = kscale;
    g.lda = lda; g.ldb = ldb; g.ldcin = ldcin; g.ldc = ldc; g.ldc32 = ldc32;
    g.alpha = alpha; g.beta = beta; g.flags = flags; g.batch = 1; g.splitk = 1;
    g.slab = ctx->det_slab; g.slab_bytes = ctx->det_bytes;
    if (ctx->prezeroed) g.flags |= DSVGP_GEMM_C_ZEROED;
    // split-K / tril zero-fill policy lives in launch_gemm
    return launch_gemm(ctx->stream, is_double, g);
}

// Plain dense fp32 product through rocBLAS (row-major C = alpha op(A) op(B) + beta C): the library's tuned Tensile kernel
// (MT128x128x64, 16x16x1 MFMA blocks, 129 TF at the C4 shape) for products with NO structure or fused epilogue -- in the
// step that is K_ZX-bar = [Q' | a] [A ; mu_bar^T].  Everything triangular / fused / fp64 stays on gemm.hip.
extern "C" int dsvgp_gemm_lib_f32(dsvgp_ctx* ctx, int flags, int M, int N, int K, float alpha, const float* A, int64_t lda,
                                  const float* B, int64_t ldb, float beta, float* C, int64_t ldc) {
    if (!ctx || !A || !B || !C || M <= 0 || N <= 0 || K <= 0) return DSVGP_EINVAL;
    if (flags & ~(DSVGP_GEMM_TRANS_A | DSVGP_GEMM_TRANS_B)) return DSVGP_EINVAL;
    // row-major C[M,N] = op(A) op(B)  <=>  column-major C^T[N,M] = op(B)^T op(A)^T; a row-major matrix read column-major
    // IS its transpose, so an untransposed row-major operand enters as "no transpose" and a stored-transposed one as "T"
    const rocblas_operation ob = (flags & DSVGP_GEMM_TRANS_B) ? rocblas_operation_transpose : rocblas_operation_none;
    const rocblas_operation oa = (flags & DSVGP_GEMM_TRANS_A) ? rocblas_operation_transpose : rocblas_operation_none;
    rocblas_status st = rocblas_sgemm((rocblas_handle)ctx->blas, ob, oa, N, M, K, &alpha, B, (rocblas_int)ldb, A,
                                      (rocblas_int)lda, &beta, C, (rocblas_int)ldc);
    return st == rocblas_status_success ? 0 : 2000 + (int)st;
}

// -------------------------------------------------------------------------------------------------
// Panel triangular solve: op(L) X = B, L lower fp64.
//   workspace = Dinv [n, n]  (inverted nb x nb diagonal blocks, indexed like L)
//             | DinvT [n, n] (its transpose: the fp64 GEMM streams an mn-contiguous A operand faster than a
//                             k-contiguous one (3.8 vs 4.6 ms at C4), so many-column forward solves read DinvT)
//             | tmp  [n + nb, nb/2]  (trtri scratch)
//             | T    [nb, nrhs]      (right-hand side of the current block row after the update)
// -------------------------------------------------------------------------------------------------
static inline int trsm_nb(int n, int nb) {
    int b = 64;
    while (b < nb) b *= 2;          // power of two >= 64
    while (b / 2 >= n && b > 64) b /= 2;
    return b;
}
extern "C" size_t dsvgp_trsm_workspace_bytes(int n, int nrhs, int nb) {
    if (n <= 0 || nrhs < 0) return 0;
    const int b = trsm_nb(n, nb);
    return sizeof(double) * ((size_t)2 * n * n + (size_t)(n + b) * (b / 2) + (size_t)b * (nrhs > 0 ? nrhs : 1)) + 256;
}

// Factorisation AND explicit inverse in the same launches (blocked Cholesky with the fused forward elimination of the
// identity, potrf.hip): L in place, L^-1 / its transpose into the Dinv / DinvT slots of the trsm `workspace`, which
// later solves use with reuse_inverse = 1.  Needs the single-block regime nb >= n.
extern "C" int dsvgp_potrf_inverse(dsvgp_ctx* ctx, double* A, int n, int64_t lda, int* info_dev, void* potrf_workspace,
                                   int nb, void* workspace) {
    if (!ctx || !A || !info_dev || !potrf_workspace || !workspace || n <= 0 || lda < n) return DSVGP_EINVAL;
    if (trsm_nb(n, nb) < n) return DSVGP_EINVAL;
    double* Dinv = (double*)workspace;
    double* DinvT = Dinv + (size_t)n * n;
    // both images of L^-1 come out of the factorisation launches (the inverse tiles are written straight and transposed)
    return launch_potrf_blocked(ctx->stream, A, n, lda, info_dev, (double*)potrf_workspace, Dinv, n, DinvT, ctx->prezeroed);
}

// First phase of dsvgp_trsm on its own: Dinv / DinvT of `workspace` from L.  potrf_workspace (may be NULL): the
// workspace dsvgp_potrf(algo 1) factored THIS L with -- its inverted 64 x 64 diagonal blocks seed the recursion.
extern "C" int dsvgp_trtri(dsvgp_ctx* ctx, const double* L, int64_t ldl, int n, int nb, const void* potrf_workspace,
                           void* workspace) {
    if (!ctx || !L || !workspace || n <= 0 || ldl < n) return DSVGP_EINVAL;
    const int b = trsm_nb(n, nb);
    double* Dinv = (double*)workspace;
    double* DinvT = Dinv + (size_t)n * n;
    double* tmp = DinvT + (size_t)n * n;
    int rc = launch_trtri_blocks(ctx->stream, L, ldl, n, b, Dinv, n, tmp, (const double*)potrf_workspace);
    if (rc) return rc;
    return dsvgp_transpose_f64(ctx, Dinv, n, n, n, DinvT, n);
}

extern "C" int dsvgp_trsm(dsvgp_ctx* ctx, const double* L, int64_t ldl, int n, int trans, const void* B, int64_t ldb,
                          int b_is_double, int nrhs, double* X64, int64_t ldx64, float* X32, int64_t ldx32, int nb,
                          void* workspace, int reuse_inverse) {
    if (!ctx || !L || !B || !workspace || n <= 0 || nrhs < 0 || ldl < n || ldb < nrhs || (X64 && ldx64 < nrhs))
        return DSVGP_EINVAL;
    if (X32 && ldx32 < nrhs) return DSVGP_EINVAL;
    if (X64 && !b_is_double && (const void*)B == (const void*)X64) return DSVGP_EINVAL;
    const int b = trsm_nb(n, nb);
    // X64 == NULL (only the fp32 copy is wanted) is possible when the whole solve is ONE product with the explicit
    // inverse (nb >= n): with several block rows the fp64 result of earlier rows feeds the later ones
    if (!X64 && (!X32 || b < n)) return DSVGP_EINVAL;
    double* Dinv = (double*)workspace;
    double* DinvT = Dinv + (size_t)n * n;
    double* tmp = DinvT + (size_t)n * n;
    double* T = tmp + (size_t)(n + b) * (b / 2);
    hipStream_t st = ctx->stream;
    if (!reuse_inverse) {
        int rc = launch_trtri_blocks(st, L, ldl, n, b, Dinv, n, tmp, nullptr);
        if (rc) return rc;
        rc = dsvgp_transpose_f64(ctx, Dinv, n, n, n, DinvT, n);
        if (rc) return rc;
    }
    if (nrhs == 0) return 0;
    const int nblk = cdiv(n, b);
    const size_t bsz = b_is_double ? 8 : 4;
    for (int s = 0; s < nblk; ++s) {
        const int I = trans ? (nblk - 1 - s) : s;
        const int r0 = I * b, nr = (n - r0 < b) ? (n - r0) : b;
        const void* Bi = (const char*)B + bsz * (size_t)r0 * ldb;
        const int kdone = trans ? (n - (r0 + nr)) : r0;      // rows of X already solved that feed this block row
        const void* rhs = Bi; int64_t ldrhs = ldb; bool rhs_float = !b_is_double;
        if (kdone == 0 && (const void*)B == (const void*)X64) {
            // in-place solve: the block row is both read (all of it) and written by the GEMM below
            hipError_t e = hipMemcpy2DAsync(T, sizeof(double) * (size_t)nrhs, Bi, sizeof(double) * (size_t)ldb,
                                            sizeof(double) * (size_t)nrhs, (size_t)nr, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return 1000 + (int)e;
            rhs = T; ldrhs = nrhs;
        }
        if (kdone > 0) {
            // T = B_I - op(L)[I, done] X[done]
            GemmArgs g{};
            g.batch = 1; g.splitk = 1;
            g.M = nr; g.N = nrhs; g.K = kdone;
            if (!trans) { g.A = L + (size_t)r0 * ldl; g.flags = 0; }
            else        { g.A = L + (size_t)(r0 + nr) * ldl + r0; g.flags = DSVGP_GEMM_TRANS_A; }
            g.lda = ldl;
            g.B = trans ? (X64 + (size_t)(r0 + nr) * ldx64) : X64; g.ldb = ldx64;
            g.Cin = Bi; g.ldcin = ldb; g.beta = 1.0; g.alpha = -1.0;
            if (!b_is_double) g.flags |= DSVGP_GEMM_CIN_IS_FLOAT;
            g.C = T; g.ldc = nrhs;
            g.slab = ctx->det_slab; g.slab_bytes = ctx->det_bytes;
            int rc = launch_gemm(st, 1, g);
            if (rc) return rc;
            rhs = T; ldrhs = nrhs; rhs_float = false;
        }
        // X_I = op(Dinv_I) rhs
        GemmArgs f{};
        f.batch = 1; f.splitk = 1;
        f.M = nr; f.N = nrhs; f.K = nr;
        f.lda = n;
        if (trans)            { f.A = Dinv + (size_t)r0 * n + r0;  f.flags = DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER; }
        else if (nrhs >= 512) { f.A = DinvT + (size_t)r0 * n + r0; f.flags = DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_LOWER; }
        else                  { f.A = Dinv + (size_t)r0 * n + r0;  f.flags = DSVGP_GEMM_A_LOWER; }
        if (rhs_float) f.flags |= DSVGP_GEMM_B_IS_FLOAT;
        f.B = rhs; f.ldb = ldrhs;
        f.alpha = 1.0; f.beta = 0.0;
        f.lean_classic = ctx->lean_classic ? 1 : 0;
        f.C = X64 ? X64 + (size_t)r0 * ldx64 : nullptr; f.ldc = ldx64;
        if (X32) { f.C32 = X32 + (size_t)r0 * ldx32; f.ldc32 = ldx32; }
        if (!X64 && (int64_t)cdiv(nr, 64) * cdiv(nrhs, 64) < 1024) {
            // small solve, fp32 result only (b >= n here: T is unused scratch of n x nrhs doubles): give the product an fp64
            // target so that it may split K -- few tiles, each a chain of n / 16 dependent stages -- and convert afterwards
            f.C = T; f.ldc = nrhs;
        }
        f.slab = ctx->det_slab; f.slab_bytes = ctx->det_bytes;
        if (ctx->prezeroed && nblk == 1) f.flags |= DSVGP_GEMM_C_ZEROED;     // (one block row: the target is used once)
        int rc = launch_gemm(st, 1, f);
        if (rc) return rc;
    }
    return 0;
}

// ---- measurement aid: what the matrix pipes of THIS card sustain -------------------------------------------------
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_32x32x2_f32 back to back from registers (no memory traffic) on every CU, long enough
// for the clocks to settle: the roof a GEMM kernel can reach under the card's power management (the data-sheet peaks are at
// 2.4 GHz; MI355X boxes hold 2.0-2.1 GHz under sustained matrix load, and differ by a few per cent among themselves).
namespace {
using d4v = double __attribute__((ext_vector_type(4)));
using f16v = float __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_rate_f64_kernel(double* out, int iters) {
    d4v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4v{0, 0, 0, 0};
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    // thread 0 of a workgroup: shader cycles and 100 MHz ticks spent in the loop (the in-kernel clock = their ratio x 100 MHz), in the slots
    // of threads 1 / 2, which write nothing
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double* o = out + (size_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) { o[0] = s; o[1] = (double)(c1 - c0); o[2] = (double)(r1 - r0); }
    else if (threadIdx.x > 2) o[threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_rate_f32_kernel(float* out, int iters) {
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float* o = out + (size_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) { o[0] = s; o[1] = (float)(c1 - c0); o[2] = (float)(r1 - r0); }
    else if (threadIdx.x > 2) o[threadIdx.x] = s;
}
}  // namespace

// mode bit 0: 1 = v_mfma_f64_16x16x4_f64, 0 = v_mfma_f32_32x32x2_f32; bit 1: ONE wave per SIMD (one 256-thread workgroup per CU: the form the
// hardware guide's 155 TF fp32 figure was measured in) instead of four.  Runs ~`millis` ms of launches on the context's stream (synchronises
// it) and returns the rate of the second half in *tflops; burst_tflops (may be null): the rate of the very FIRST launch (~2 ms from an idle
// card, before the power management has lowered the clock); clock_ghz (may be null): the in-kernel clock of the last launch, median over
// workgroups of delta s_memtime / delta s_memrealtime x 100 MHz.  scratch: 256 * 1024 * 8 bytes.
static int mfma_rate_impl(dsvgp_ctx* ctx, int mode, int millis, void* scratch, double* tflops, double* burst_tflops, double* clock_ghz) {
    if (!ctx || !scratch || !tflops || millis < 2 || millis > 2000 || mode < 0 || mode > 3) return DSVGP_EINVAL;
    const bool is_double = mode & 1;
    hipStream_t st = ctx->stream;
    const int per_cu = (mode & 2) ? 1 : 4;
    const int grid = 256 * per_cu, iters = 2000 * (4 / per_cu);  // 4 workgroups (16 waves) or 1 (4 waves) per CU; ~1.7 ms (f64) per launch
    const double flop_per_launch = is_double ? (double)grid * 4 * iters * 32 * 2048.0 : (double)grid * 4 * iters * 32 * 4096.0;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1000 + (int)hipGetLastError();
    auto launch = [&]() {
        if (is_double) hipLaunchKernelGGL(mfma_rate_f64_kernel, dim3(grid), dim3(256), 0, st, (double*)scratch, iters);
        else hipLaunchKernelGGL(mfma_rate_f32_kernel, dim3(grid), dim3(256), 0, st, (float*)scratch, iters);
    };
    // calibrate the launch count on one launch, warm up for half the budget, time the other half
    (void)hipEventRecord(e0, st); launch(); (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) return 1000 + (int)hipGetLastError();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (burst_tflops) *burst_tflops = flop_per_launch / ((ms > 1e-3f ? ms : 1e-3f) * 1e-3) / 1e12;
    int n = (int)(0.5 * millis / (ms > 1e-3f ? ms : 1e-3f));
    n = n < 1 ? 1 : (n > 2000 ? 2000 : n);
    for (int i = 0; i < n; ++i) launch();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) launch();
    (void)hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return 1000 + (int)e;
    *tflops = flop_per_launch * n / (ms * 1e-3) / 1e12;
    if (clock_ghz) {
        // (grid <= 1024 workgroups: the stamps of all of them; median of the ratios)
        double ratio[1024];
        const size_t esz = is_double ? 8 : 4;
        char* host = (char*)malloc((size_t)grid * 256 * esz);
        if (!host) return DSVGP_EINVAL;
        e = hipMemcpy(host, scratch, (size_t)grid * 256 * esz, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(host); return 1000 + (int)e; }
        for (int b = 0; b < grid; ++b) {
            const double cyc = is_double ? ((double*)host)[(size_t)b * 256 + 1] : (double)((float*)host)[(size_t)b * 256 + 1];
            const double tck = is_double ? ((double*)host)[(size_t)b * 256 + 2] : (double)((float*)host)[(size_t)b * 256 + 2];
            ratio[b] = tck > 0 ? cyc / tck : 0.0;
        }
        free(host);
        for (int i = 1; i < grid; ++i) {                         // (insertion sort: 1024 values, once per probe)
            const double v = ratio[i];
            int j = i - 1;
            while (j >= 0 && ratio[j] > v) { ratio[j + 1] = ratio[j]; --j; }
            ratio[j + 1] = v;
        }
        *clock_ghz = ratio[grid / 2] * 0.1;                      // cycles per 10 ns tick -> GHz
    }
    return 0;
}
extern "C" int dsvgp_mfma_rate(dsvgp_ctx* ctx, int is_double, int millis, void* scratch, double* tflops) {
    if (is_double != 0 && is_double != 1) return DSVGP_EINVAL;
    return mfma_rate_impl(ctx, is_double, millis, scratch, tflops, nullptr, nullptr);
}
extern "C" int dsvgp_mfma_rate2(dsvgp_ctx* ctx, int mode, int millis, void* scratch, double* tflops, double* burst_tflops, double* clock_ghz) {
    return mfma_rate_impl(ctx, mode, millis, scratch, tflops, burst_tflops, clock_ghz);
}
