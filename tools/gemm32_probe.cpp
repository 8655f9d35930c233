// Standalone correctness + timing probe of csrc/gemm32.hip (fp32 GEMM on v_mfma_f32_32x32x2_f32) against rocBLAS sgemm
// on the shapes of the ELBO fast path.  Built by tools/gemm32_variants.sh together with gemm32.hip itself (variant -D
// flags), so no libdsvgp_hip.so is involved.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"

#ifdef G32_STAMP
int g32_read_stamps(unsigned long long* out);
#endif
// the two helpers of gemm.hip that launch_gemm32 calls (the probe links gemm32.hip alone)
hipError_t zero_block(void* C, size_t esz, int64_t ld, int M, int N, hipStream_t st) {
    if (ld == N) return hipMemsetAsync(C, 0, (size_t)M * N * esz, st);
    return hipMemset2DAsync(C, (size_t)ld * esz, 0, (size_t)N * esz, (size_t)M, st);
}
int launch_splitk_reduce(hipStream_t, int, const void*, int, int, int, void*, int64_t, float*, int64_t, int, int) { return DSVGP_EINVAL; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)

static void fill(float* p, size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
}

struct Case { const char* name; int flags, M, N, K; int64_t lda, ldb; double flops; };

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    hipStream_t st; CK(hipStreamCreate(&st));
    rocblas_handle h; rocblas_create_handle(&h); rocblas_set_stream(h, st);
    const int TA = DSVGP_GEMM_TRANS_A, TB = DSVGP_GEMM_TRANS_B, OL = DSVGP_GEMM_OUT_LOWER;
    Case cases[] = {
        {"C4 dense  Kzx-bar [3000x3001]x[3001x24576] A k-contig", 0, 3000, 24576, 3001, 3004, 24576, 2.0 * 3000 * 3001 * 24576.0},
        {"C4 dense  transposed A copy   (A m-contig)           ", TA, 3000, 24576, 3001, 3000, 24576, 2.0 * 3000 * 3001 * 24576.0},
        {"C4 Gram   tril([A;mu]A^T) K=24576 (both k-contig)    ", TB | OL, 3001, 3000, 24576, 24576, 24576, 1.0 * 3001 * 3000 * 24576.0},
        {"C4 Gram   from a transposed copy (both m-contig)     ", TA | OL, 3001, 3000, 24576, 3004, 3000, 1.0 * 3001 * 3000 * 24576.0},
        {"C4 Gram   A k-contig, transposed copy as B (m-contig)", OL, 3001, 3000, 24576, 24576, 3000, 1.0 * 3001 * 3000 * 24576.0},
        {"C4/8 dense N=3072                                    ", 0, 3000, 3072, 3001, 3004, 3072, 2.0 * 3000 * 3001 * 3072.0},
        {"C4/8 Gram  K=3072                                    ", TB | OL, 3001, 3000, 3072, 3072, 3072, 1.0 * 3001 * 3000 * 3072.0},
        {"C3 dense  [3300x3301]x[3301x5632]                    ", 0, 3300, 5632, 3301, 3304, 5632, 2.0 * 3300 * 3301 * 5632.0},
        {"C3 Gram   K=5632                                     ", TB | OL, 3301, 3300, 5632, 5632, 5632, 1.0 * 3301 * 3300 * 5632.0},
        {"odd       M=700 N=1100 K=1300 (ragged everything)    ", 0, 700, 1100, 1300, 1300, 1100, 2.0 * 700 * 1100 * 1300.0},
        {"odd TT    M=700 N=1100 K=1300                        ", TA | TB, 700, 1100, 1300, 700, 1300, 2.0 * 700 * 1100 * 1300.0},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& c : cases) {
        const bool ta = c.flags & TA, tb = c.flags & TB, ol = c.flags & OL;
        const size_t na = (size_t)(ta ? c.K : c.M) * c.lda, nb = (size_t)(tb ? c.N : c.K) * c.ldb, nc = (size_t)c.M * c.N;
        float *A, *B, *C, *R;
        CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4)); CK(hipMalloc(&R, nc * 4));
        fill(A, na, 1); fill(B, nb, 2);
        int extra = 0;
        if (c.K % 4) {      // k-contiguous operands: zero the caller-side padding K .. roundup4(K) and promise it (DSVGP_GEMM_K_PADDED)
            const int K4 = (c.K + 3) / 4 * 4;
            if (!ta) CK(hipMemset2D(A + c.K, c.lda * 4, 0, (size_t)(K4 - c.K) * 4, c.M));
            if (tb) CK(hipMemset2D(B + c.K, c.ldb * 4, 0, (size_t)(K4 - c.K) * 4, c.N));
            extra = DSVGP_GEMM_K_PADDED;
        }
        GemmArgs g{};
        g.M = c.M; g.N = c.N; g.K = c.K; g.A = A; g.B = B; g.C = C; g.lda = c.lda; g.ldb = c.ldb; g.ldc = c.N;
        g.alpha = 0.75; g.beta = 0.0; g.flags = c.flags | extra; g.batch = 1; g.splitk = 1;
        float ms_mine = 0.f, ms_lib = 0.f;
        int rc = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            const int n = rep ? reps : 1;
            for (int i = 0; i < n; ++i) rc = launch_gemm32(st, g);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_mine, e0, e1)); ms_mine /= n;
        }
        if (rc != 1) { printf("%s  launch_gemm32 rc=%d (not taken)\n", c.name, rc); continue; }
        // reference: row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T
        const float alpha = 0.75f, beta = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            const int n = rep ? reps : 1;
            for (int i = 0; i < n; ++i)
                rocblas_sgemm(h, tb ? rocblas_operation_transpose : rocblas_operation_none,
                              ta ? rocblas_operation_transpose : rocblas_operation_none, c.N, c.M, c.K, &alpha, B, (int)c.ldb, A,
                              (int)c.lda, &beta, R, c.N);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_lib, e0, e1)); ms_lib /= n;
        }
        std::vector<float> hc(nc), hr(nc);
        CK(hipMemcpy(hc.data(), C, nc * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), R, nc * 4, hipMemcpyDeviceToHost));
        double maxref = 0, maxerr = 0, upper = 0;
        for (int m = 0; m < c.M; ++m)
            for (int n = 0; n < c.N; ++n) {
                const double r = hr[(size_t)m * c.N + n], v = hc[(size_t)m * c.N + n];
                if (ol && n > m) { upper = fmax(upper, fabs(v)); continue; }
                maxref = fmax(maxref, fabs(r)); maxerr = fmax(maxerr, fabs(v - r));
            }
        printf("%s  %8.3f ms %7.2f TF   rocBLAS %8.3f ms %7.2f TF (dense)   rel.err %.2e%s\n", c.name, ms_mine,
               c.flops / ms_mine / 1e9, ms_lib, 2.0 * c.M * c.N * (double)c.K / ms_lib / 1e9, maxerr / maxref,
               ol ? (upper == 0 ? "  upper=0 ok" : "  UPPER NONZERO") : "");
#ifdef G32_STAMP
        {   unsigned long long d[8]; g32_read_stamps(d);
            const double n = (double)d[4], tot = (double)(d[0] + d[1] + d[2] + d[3]);
            printf("    stamps (one mid-grid wave, %.0f stages): per stage %.0f cycles = dma issue %.0f + reads/MFMA %.0f + vmcnt wait %.0f + barrier %.0f\n",
                   n, tot / n, d[0] / n, d[1] / n, d[2] / n, d[3] / n); }
#endif
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(R));
    }
    return 0;
}
