// Standalone timing probe for the MFMA GEMM shapes of the DSVGP step (C4 geometry).
// Build + run on the GPU box:  hipcc -O2 tools/gemm_probe.cpp -Iinclude -L<pkg> -ldsvgp_hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dsvgp.h"
#ifdef GEMM_CLOCK
extern "C" int dsvgp_debug_gemm_clock(unsigned long long* out);
#endif

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)

static void fill(void* p, size_t n, bool dbl) {
    std::vector<double> hd; std::vector<float> hf;
    if (dbl) { hd.resize(n); for (size_t i = 0; i < n; ++i) hd[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0; CK(hipMemcpy(p, hd.data(), n * 8, hipMemcpyHostToDevice)); }
    else     { hf.resize(n); for (size_t i = 0; i < n; ++i) hf[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f; CK(hipMemcpy(p, hf.data(), n * 4, hipMemcpyHostToDevice)); }
}

int main(int argc, char** argv) {
    const int Mp = argc > 1 ? atoi(argv[1]) : 3000, Bp = argc > 2 ? atoi(argv[2]) : 24576;
    dsvgp_ctx* ctx; if (dsvgp_create(&ctx)) return 1;
    hipStream_t st; CK(hipStreamCreate(&st)); dsvgp_set_stream(ctx, st);
    double *L64, *X64, *S64; float *K32, *A32, *W32, *LS32, *vs, *S32;
    CK(hipMalloc(&L64, 8ul * Mp * Mp)); CK(hipMalloc(&X64, 8ul * Mp * Bp)); CK(hipMalloc(&S64, 8ul * Mp * Mp));
    CK(hipMalloc(&K32, 4ul * Mp * Bp)); CK(hipMalloc(&A32, 4ul * Mp * Bp)); CK(hipMalloc(&W32, 4ul * Mp * Bp));
    CK(hipMalloc(&LS32, 4ul * Mp * Mp)); CK(hipMalloc(&vs, 4ul * Bp)); CK(hipMalloc(&S32, 4ul * Mp * Mp));
    fill(L64, (size_t)Mp * Mp, true); fill(K32, (size_t)Mp * Bp, false); fill(A32, (size_t)Mp * Bp, false);
    fill(LS32, (size_t)Mp * Mp, false); fill(vs, Bp, false); fill(X64, (size_t)Mp * Bp, true);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int dbl, flags, M, N, K; const void *A; long lda; const void* B; long ldb; void* C; long ldc; const float* ks; double flops; };
    const double tri = (double)Mp * Mp * Bp, full = 2.0 * Mp * Mp * Bp, cube = (double)Mp * Mp * Mp;
    Case cases[] = {
        {"f64 solve  Dinv(lower) x K_ZX(f32)   ", 1, DSVGP_GEMM_A_LOWER | DSVGP_GEMM_B_IS_FLOAT, Mp, Bp, Mp, L64, Mp, K32, Bp, X64, Bp, nullptr, tri},
        {"f64 solveT Dinv^T(upper) x Abar(f32)  ", 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER | DSVGP_GEMM_B_IS_FLOAT, Mp, Bp, Mp, L64, Mp, K32, Bp, X64, Bp, nullptr, tri},
        {"f64 dense  L[MxK] x X64               ", 1, 0, Mp, Bp, Mp, L64, Mp, X64, Bp, X64, Bp, nullptr, full},
        {"f64 Lbar   tril(Kb64 A64^T) K=B'      ", 1, DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Bp, X64, Bp, X64, Bp, S64, Mp, nullptr, tri},
        {"f64 square L^T Lbar (tri x tri)       ", 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER | DSVGP_GEMM_B_LOWER, Mp, Mp, Mp, L64, Mp, L64, Mp, S64, Mp, nullptr, (double)Mp * Mp * Mp / 1.5},
        {"f32 W = L_S^T A                       ", 0, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER, Mp, Bp, Mp, LS32, Mp, A32, Bp, W32, Bp, nullptr, tri},
        {"f32 U = L_S W                         ", 0, DSVGP_GEMM_A_LOWER, Mp, Bp, Mp, LS32, Mp, A32, Bp, W32, Bp, nullptr, tri},
        {"f32 dense                             ", 0, 0, Mp, Bp, Mp, LS32, Mp, A32, Bp, W32, Bp, nullptr, full},
        {"f32 dLS    tril(A diag(v) W^T) K=B'   ", 0, DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Bp, A32, Bp, K32, Bp, S32, Mp, vs, tri},
        {"f32 Gram   tril(A A^T) k-contig ops   ", 0, DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Bp, A32, Bp, A32, Bp, S32, Mp, nullptr, tri},
        {"f32 Gram   tril(At^T At) transposed    ", 0, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Bp, A32, Mp, A32, Mp, S32, Mp, nullptr, tri},
        {"f32 dense  transposed A (Kb-like)      ", 0, DSVGP_GEMM_TRANS_A, Mp, Bp, Mp, LS32, Mp, A32, Bp, W32, Bp, nullptr, full},
        // M' x M' x M' class (Cholesky backward, Q', L-bar of the ELBO fast path, S, dL_S)
        {"f64 MxM solveT Dinv^T(upper) x G      ", 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER, Mp, Mp, Mp, L64, Mp, S64, Mp, X64, Mp, nullptr, cube},
        {"f64 MxM solveT, f32 rhs (Q')          ", 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER | DSVGP_GEMM_B_IS_FLOAT, Mp, Mp, Mp, L64, Mp, LS32, Mp, X64, Mp, nullptr, cube},
        {"f64 MxM tril(Q G) (L-bar fast, f32 B) ", 1, DSVGP_GEMM_OUT_LOWER | DSVGP_GEMM_B_IS_FLOAT, Mp, Mp, Mp, L64, Mp, LS32, Mp, S64, Mp, nullptr, cube},
        {"f32 MxM S = L_S L_S^T                 ", 0, DSVGP_GEMM_A_LOWER | DSVGP_GEMM_TRANS_B | DSVGP_GEMM_B_UPPER, Mp, Mp, Mp, LS32, Mp, LS32, Mp, S32, Mp, nullptr, cube * 2 / 3},
        {"f32 MxM tril(G tril(L_S))             ", 0, DSVGP_GEMM_B_LOWER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, LS32, Mp, LS32, Mp, S32, Mp, nullptr, cube * 2 / 3},
    };
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {   // rep 0 = warm-up
            CK(hipEventRecord(e0, st));
            const int n = rep ? 3 : 1;
            for (int i = 0; i < n; ++i) {
                int rc = dsvgp_gemm(ctx, c.dbl, c.flags, c.M, c.N, c.K, 1.0, c.A, c.lda, c.B, c.ldb, 0.0, nullptr, 0, c.C, c.ldc, nullptr, 0, c.ks);
                if (rc) { printf("%s rc=%d\n", c.name, rc); break; }
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) {
                printf("%s %8.3f ms  %7.2f TFLOP/s", c.name, ms / n, c.flops / (ms / n) / 1e9);
#ifdef GEMM_CLOCK
                unsigned long long d[2]; dsvgp_debug_gemm_clock(d);
                printf("   in-kernel clock %.2f GHz (mid-grid WG: %.0f us)", (double)d[0] / (double)d[1] * 0.1, (double)d[1] / 100.0);
#endif
                printf("\n");
            }
        }
    }
    return 0;
}
