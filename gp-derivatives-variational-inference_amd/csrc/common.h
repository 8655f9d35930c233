// Internal declarations shared by the HIP translation units of libdsvgp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "dsvgp.h"

struct dsvgp_ctx {
    hipStream_t stream = nullptr;
    void* blas = nullptr;  // rocblas_handle (opaque here so only potrf.hip needs the rocBLAS headers)
    // deterministic mode (dsvgp_set_deterministic): caller-owned scratch for split-K slabs / per-workgroup partial sums;
    // null = atomics allowed (run-order dependent rounding of sums)
    void* det_slab = nullptr;
    size_t det_bytes = 0;
    // set by dsvgp_elbo_step_f32 (small problems) while it runs: every output that a launcher would clear before use (split-K
    // targets, OUT_LOWER blocks, the potrf status word, the residual sums) lies in ONE region the step has cleared with a
    // single memset -- the launchers skip their own clears (a dozen ~5 us fill launches per step at M' = 600)
    bool prezeroed = false;
    // hint set by the one-call step around its [Q' | a] solve when a large dense product follows it on the same stream while the
    // side stream's G L_S product is still running (C4): that solve then keeps the register-staged lean kernel (gemm64.hip) -- on the
    // pipelined one it finishes 0.2 ms earlier, starves G L_S, and G L_S's tail then collides with the dense product: the step is 0.1 ms
    // slower (profiles/r05_o_gemm64l_lean.txt)
    bool lean_classic = false;
    // set by dsvgp_elbo_step_f32 around its two kernel backwards (K_ZX-bar, K_ZZ-bar): their last launch -- kernel_bwd_points_kernel,
    // which adds the tile kernels' slabs into d_x1 / d_v1 / d_hyp -- is not queued but noted here, and dsvgp_kernel_bwd_points_flush runs
    // ONE such launch over both slab sets, with the step's scalar tail folded in (round 6: three ~5 us launches fewer at M' = 600)
    // set by the one-call steps around K_ZZ's assembly: the result goes straight into the blocked Cholesky factorisation, which reads the
    // 64 x 64 blocks on and below the block diagonal only -- the assembly kernels that know the switch skip the column tiles that lie
    // entirely to the right of them (round 6: 47 -> ~25 us on the step's critical path at M' = 3000)
    bool fwd_lower_only = false;
    struct PointsJob { const float* slab; int ns; const float* partials; int nparts; float sym; };
    bool defer_points = false;
    int n_deferred = 0;
    PointsJob deferred[2];
};

#define DSVGP_LAUNCH_CHECK()                                  \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return 1000 + (int)e__;        \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// MFMA GEMM (gemm.hip).  C = alpha*op(A)op(B) + beta*Cin on v_mfma_{f32,f64}_16x16x4.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    int M, N, K;
    int M_last, K_last;    // M / K of the LAST batch entry (ragged trtri pairs); 0 = same as M / K
    const void* A; const void* B; const void* Cin; void* C; float* C32; const float* kscale;
    int64_t lda, ldb, ldcin, ldc, ldc32;
    int64_t sA, sB, sC;    // batch strides (elements); Cin/C32 are not batched
    double alpha, beta;
    int flags;
    int batch;
    int splitk;            // >1: epilogue is atomicAdd(alpha*acc) into a caller-initialised C
    int tiles_m, tiles_n, supertile, bn, chunk, bm;   // filled by launch_gemm (bn / bm = output tile width / height, 128 or 64)
    // deterministic mode: split-K slices store their partial tiles to this scratch ([slice][M][N], packed) and a
    // fixed-order pass adds them (launch_splitk_reduce) instead of meeting in atomics; null = atomics
    void* slab; size_t slab_bytes;
    // row-range pieces of a product with a triangular A (the forward solve pipelined under the Cholesky chain, step.hip): the
    // piece's row 0 is row tri_off of the triangle (its K range ends at tri_off + row + 1); wide64: take the 64 x 192 kernel of
    // gemm64.hip whatever the tile count; lds_pad: extra (unused) dynamic LDS per workgroup, which bounds the workgroups a CU
    // takes so that another stream's launches keep finding room.  Only gemm64.hip's wide kernel honours these: launch_gemm
    // refuses (DSVGP_EINVAL) a product with tri_off / wide64 set that it cannot send there.
    int tri_off, wide64, lds_pad;
    int small64;         // (set by launch_gemm) a few-tile fp64 product offered to gemm64.hip's four-buffer pipelined form
    int lean_classic;    // 1: gemm64.hip's lean class stays on the register-staged kernel (see dsvgp_ctx::lean_classic)
};
// how many split-K slices fit the slab (>= 2) or 1 (= do not split); esz = bytes per element
static inline int slab_slices(const GemmArgs& g, int want, size_t esz) {
    if (!g.slab || want <= 1) return want;
    const size_t per = (size_t)g.M * (size_t)g.N * esz;
    const size_t fit = per ? g.slab_bytes / per : 0;
    const int sk = (int)(fit < (size_t)want ? fit : (size_t)want);
    return sk >= 2 ? sk : 1;
}
// C (+ C32) = [C +] sum_s slab[s] in the fixed order s = 0, 1, ... (out_lower: n > m defined as zero) -- gemm.hip
int launch_splitk_reduce(hipStream_t st, int is_double, const void* slab, int nslices, int M, int N, void* C, int64_t ldc,
                         float* C32, int64_t ldc32, int out_lower, int accumulate);
constexpr int DSVGP_GEMM_KEEP_UPPER = 1 << 20;   // internal flag (with OUT_LOWER): do not touch m < n
constexpr int DSVGP_GEMM_C_ZEROED = 1 << 21;     // internal flag: the caller has cleared C / C32 (skip the launcher's own clears)
constexpr int DSVGP_GEMM_UPPER_UNDEF = 1 << 22;  // internal flag (with OUT_LOWER, gemm32.hip): nobody reads the strict upper triangle -- leave it undefined instead of zero-filling it
int launch_gemm(hipStream_t st, int is_double, const GemmArgs& g);
// zero an M x N block (element size esz) with leading dimension ld: linear memset for contiguous rows, a fill kernel of our
// own for padded rows (the runtime's pitched 2-D memset runs below 1 TB/s) -- gemm.hip
hipError_t zero_block(void* C, size_t esz, int64_t ld, int M, int N, hipStream_t st);
void launch_cvt_f64_f32(hipStream_t st, const double* C, int64_t ldc, float* C32, int64_t ldc32, int M, int N);   // gemm64.hip
void launch_widen_f32_f64(hipStream_t st, const float* src, int64_t ld, double* dst, int64_t ldd, int M, int N);      // gemm64.hip
int launch_gemm64(hipStream_t st, const GemmArgs& g);     // gemm64.hip: 1 = taken, 0 = not eligible, > 1 = error
int launch_gemm32(hipStream_t st, const GemmArgs& g);     // gemm32.hip (fp32, 32x32x2 MFMA): same convention

// blocked Cholesky (potrf.hip)
size_t potrf_blocked_workspace_bytes(int n);
// hooks: after the launch that completes block row after_k of L^-1 (rows < 64 (after_k + 1) of Yinv / YinvT are final once it has
// run) the event is recorded on st -- another stream may then start consuming those rows while the chain goes on
struct PotrfHook { int after_k; hipEvent_t ev; };
int launch_potrf_blocked(hipStream_t st, double* A, int n, int64_t lda, int* info, double* ws, double* Yinv, int64_t ldy,
                         double* YinvT, bool info_zeroed = false, const PotrfHook* hooks = nullptr, int nhooks = 0, int pipe_from = 0);
// pipe_from: first block column whose launch may use the one-workgroup-per-CU pipelined kernel (potrf.hip, PIPE): a caller whose
// other stream shares the CUs with the first launches of the chain passes the launch index from which the chain is alone
// pieces of the one-call step (csrc/step.hip) that fold tiny dependent launches into their neighbours
int launch_pack_both(hipStream_t st, const float* Z, const float* V, int M, const float* X, const float* D, int B, int d, int p,
                     const float* rl, const float* rs, const float* rn, float* hyp, float* center, float* PZ, float* sZ, float* vZ,
                     float* PX, float* sX, float* vX);
int kernel_bwd_points_flush(dsvgp_ctx* ctx, const float* P1, const float* vnorm1, int n1, int d, int p, const float* hyp, float* d_x1,
                            float* d_v1, float* d_hyp, const float* scal, const float* kl0, double rows, double num_data, const float* rl,
                            const float* rs, const float* rn, float* drl, float* drs, float* drn, float* dconst, float* loss);
int launch_column_mean_hyp(hipStream_t st, const float* x, int n, int d, float* center, const float* rl, const float* rs,
                           const float* rn, float* hyp);                                       // assemble.hip
int launch_widen_sym_f32_f64(hipStream_t st, const float* src, int64_t lds, double* dst, int64_t ldd, int n);    // elbo.hip: fp64 mirror of an fp32 lower triangle
int launch_mirror_sminus_i_col(hipStream_t st, float* A, int n, int64_t lda, const float* m, const float* hyp, float rows, double* W = nullptr,
                               int64_t ldw = 0);   // elbo.hip
int launch_variational_terms(hipStream_t st, const float* m, const float* LS, int64_t ldls, int Mp, double num_data, int flags,
                             const float* hyp, double global_rows, const float* G, int64_t ldg, float t1_scale, float* kl_out,
                             float* sums, const float* dm_src, float* d_m, float* d_LS, int64_t lddls, int fin_npts, int fin_p,
                             float* fin_scal);                                                 // elbo.hip
// column statistics (mu = A^T m + c, prior diagonal) + residuals / mu-bar / their sums: two launches instead of three
int launch_stats_residual(dsvgp_ctx* ctx, const float* A, int64_t lda, int Mp, int ncols, int p, const float* m,
                          const float* constant, const float* hyp, float* mu, float* var, void* workspace, const float* y,
                          double global_rows, float* mu_bar, float* sums);                     // elbo.hip
// Z-bar, V-bar, d_hyp[0..1] *= 2 vbar and the scalar tail of the step (dsvgp_scale_by_vbar + dsvgp_step_epilogue) in one launch
int launch_scale_epilogue(dsvgp_ctx* ctx, float* x0, int64_t n0, float* x1, int64_t n1, const float* hyp, double rows,
                          const float* scal, const float* kl0, double num_data, const float* rl, const float* rs, const float* rn,
                          float* dh, float* drl, float* drs, float* drn, float* dconst, float* loss);   // elbo.hip

// trtri of the nb x nb diagonal blocks of the lower-triangular L into Dinv (same indexing as L,
// leading dimension ldd); tmp is an n x (nb/2) double scratch.  X64 (may be null): the inverted 64 x 64 diagonal
// blocks the blocked Cholesky left in its workspace ([k][64][64]).
int launch_trtri_blocks(hipStream_t st, const double* L, int64_t ldl, int n, int nb, double* Dinv,
                        int64_t ldd, double* tmp, const double* X64);
