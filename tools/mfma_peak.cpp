// Measured issue rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (no memory traffic): the practical MFMA roof.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.cpp -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = double __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k64(double* out, int iters, unsigned long long* cyc) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, unsigned long long* cyc) {
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K, typename T>
void run(const char* name, K kern, int nacc, int wgs_per_cu, T* out, unsigned long long* cyc, double flop_per_mfma) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma_wave = (double)iters * 4 * nacc;
    const double total = n_mfma_wave * 4 * grid * flop_per_mfma;
    printf("%s  acc/wave %d  waves/SIMD %d: %.1f cycles per MFMA per wave (x waves/SIMD sharing the pipe), %.1f TFLOP/s whole chip\n",
           name, nacc, wgs_per_cu, (double)c / n_mfma_wave, total / (ms * 1e-3) / 1e12);
}

int main() {
    double* o64; float* o32; unsigned long long* cyc;
    hipMalloc(&o64, 8 * 256 * 256 * 8); hipMalloc(&o32, 4 * 256 * 256 * 8); hipMalloc(&cyc, 8);
    run("f64 16x16x4", k64<4>, 4, 1, o64, cyc, 2048.0);
    run("f64 16x16x4", k64<8>, 8, 1, o64, cyc, 2048.0);
    run("f64 16x16x4", k64<4>, 4, 2, o64, cyc, 2048.0);
    run("f64 16x16x4", k64<8>, 8, 4, o64, cyc, 2048.0);
    run("f32 16x16x4", k32<4>, 4, 1, o32, cyc, 2048.0);
    run("f32 16x16x4", k32<8>, 8, 1, o32, cyc, 2048.0);
    run("f32 16x16x4", k32<8>, 8, 2, o32, cyc, 2048.0);
    run("f32 16x16x4", k32<8>, 8, 4, o32, cyc, 2048.0);
    return 0;
}
