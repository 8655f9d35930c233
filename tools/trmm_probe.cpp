// tools only: rocBLAS dtrmm / dgemm timings at the M' x M' shapes of the step (is a library triangular product worth routing to?)
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
int main() {
    const int n = 3000, m = 3001;
    rocblas_handle h; rocblas_create_handle(&h);
    double *A, *B, *C;
    hipMalloc(&A, sizeof(double) * n * n); hipMalloc(&B, sizeof(double) * (size_t)n * m); hipMalloc(&C, sizeof(double) * (size_t)n * m);
    std::vector<double> ha((size_t)n * n), hb((size_t)n * m);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (double)((i * 40503u) % 1000) / 1000.0 - 0.5;
    hipMemcpy(A, ha.data(), sizeof(double) * ha.size(), hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice);
    const double one = 1.0, zero = 0.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, double gf, auto fn) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) fn();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-46s %8.3f ms  %6.1f TFLOP/s\n", name, ms, gf / ms);
    };
    const double tri = (double)n * n * m / 1e9;   // triangular product flops (2 n^2 m / 2) in GF
    for (int side = 0; side < 2; ++side)
        for (int up = 0; up < 2; ++up)
            for (int tr = 0; tr < 2; ++tr) {
                char name[96];
                snprintf(name, sizeof name, "dtrmm side=%s uplo=%s trans=%s", side ? "right" : "left", up ? "upper" : "lower", tr ? "T" : "N");
                run(name, tri, [&] {
                    rocblas_dtrmm(h, side ? rocblas_side_right : rocblas_side_left, up ? rocblas_fill_upper : rocblas_fill_lower,
                                  tr ? rocblas_operation_transpose : rocblas_operation_none, rocblas_diagonal_non_unit,
                                  side ? m : n, side ? n : m, &one, A, n, B, side ? m : n, C, side ? m : n);
                });
            }
    run("dgemm NN 3000 x 3001 x 3000 (dense)", 2 * tri, [&] {
        rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, m, n, &one, A, n, B, n, &zero, C, n);
    });
    return 0;
}
