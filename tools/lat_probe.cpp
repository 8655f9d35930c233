// tools only (round 6): dependent-chain latencies of the instructions the Cholesky chain is made of, one wave alone on a CU (s_memtime, shader
// cycles): v_fma_f64, v_mul_f64, v_rsq_f64, v_readlane -> VALU, ds_write -> ds_read round trip, v_mfma_f64_16x16x4 -> VALU read, and the 1 / sqrt forms
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/lat_probe.cpp -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = double __attribute__((ext_vector_type(4)));

#define T0() unsigned long long t0_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory")
#define T1(slot) do { unsigned long long t1_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); \
    if (threadIdx.x == 0) out[slot] = t1_ - t0_; } while (0)

constexpr int N = 256;

__global__ void k_fma(unsigned long long* out, double* sink, double x) {
    double a = x + threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_fma(a, 1.0000001, 0.5);
    asm volatile("" :: "v"(a));
    T1(0);
    sink[threadIdx.x] = a;
}
__global__ void k_mul(unsigned long long* out, double* sink, double x) {
    double a = x + threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) a = a * 1.0000001;
    asm volatile("" :: "v"(a));
    T1(1);
    sink[threadIdx.x] = a;
}
__global__ void k_rsq(unsigned long long* out, double* sink, double x) {
    double a = x + threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_amdgcn_rsq(a);
    asm volatile("" :: "v"(a));
    T1(2);
    sink[threadIdx.x] = a;
}
// independent fma stream (issue rate of one wave)
__global__ void k_fma_indep(unsigned long long* out, double* sink, double x) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = x + threadIdx.x + j;
    T0();
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 0.5);
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    asm volatile("" :: "v"(s));
    T1(3);
    sink[threadIdx.x] = s;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
// fma -> readlane -> fma chain
__global__ void k_readlane(unsigned long long* out, double* sink, double x) {
    double a = x + threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = readlane_f64(a, (i * 7) & 63);
        a = __builtin_fma(a, 0.5, s);
    }
    asm volatile("" :: "v"(a));
    T1(4);
    sink[threadIdx.x] = a;
}
// ds_write -> ds_read (other lane) -> fma chain
__global__ void k_lds(unsigned long long* out, double* sink, double x) {
    __shared__ double buf[64];
    double a = x + threadIdx.x;
    const int l = threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        buf[l] = a;
        __builtin_amdgcn_wave_barrier();
        const double s = buf[(l + 17) & 63];
        __builtin_amdgcn_wave_barrier();
        a = __builtin_fma(s, 0.5, 1.0);
    }
    asm volatile("" :: "v"(a));
    T1(5);
    sink[threadIdx.x] = a;
}
// mfma -> VALU use of the result -> mfma operand
__global__ void k_mfma(unsigned long long* out, double* sink, double x) {
    d4 c = d4{0, 0, 0, 0};
    double a = x + threadIdx.x * 1e-3;
    T0();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c, 0, 0, 0);
        a = __builtin_fma(c[1], 1e-3, 0.5);
    }
    asm volatile("" :: "v"(a));
    T1(6);
    sink[threadIdx.x] = a + c[0];
}
// dependent mfma chain through the accumulator
__global__ void k_mfma_acc(unsigned long long* out, double* sink, double x) {
    d4 c = d4{0, 0, 0, 0};
    double a = x + threadIdx.x * 1e-3;
    T0();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c, 0, 0, 0);
    asm volatile("" :: "v"(c));
    T1(7);
    sink[threadIdx.x] = c[0];
}
__device__ __forceinline__ double rsqrt_nr2(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double h = 0.5 * y;
        y = __builtin_fma(h, __builtin_fma(-(d * y), y, 1.0), y);
    }
    return y;
}
__device__ __forceinline__ double rsqrt_c(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double gq = d * y, hy = 0.5 * y;
    const double e = __builtin_fma(-gq, y, 1.0), hg = 0.5 * gq;
    const double y1 = __builtin_fma(hy, e, y), g1 = __builtin_fma(hg, e, gq);
    const double e1 = __builtin_fma(-g1, y1, 1.0), hy1 = 0.5 * y1;
    return __builtin_fma(hy1, e1, y1);
}
__global__ void k_rsqnr(unsigned long long* out, double* sink, double x) {
    double a = x + threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) a = rsqrt_nr2(a) + 1.5;
    asm volatile("" :: "v"(a));
    T1(8);
    sink[threadIdx.x] = a;
}
__global__ void k_rsqc(unsigned long long* out, double* sink, double x) {
    double a = x + threadIdx.x;
    T0();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) a = rsqrt_c(a) + 1.5;
    asm volatile("" :: "v"(a));
    T1(9);
    sink[threadIdx.x] = a;
}
// accuracy of v_rsq_f64 itself and of the two refinements
__global__ void k_acc(double* res) {
    const double d = 0.001 + 0.37 * threadIdx.x + 1e-3 * threadIdx.x * threadIdx.x;
    res[threadIdx.x] = __builtin_amdgcn_rsq(d);
    res[64 + threadIdx.x] = rsqrt_nr2(d);
    res[128 + threadIdx.x] = rsqrt_c(d);
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * y;
    y = __builtin_fma(h, __builtin_fma(-(d * y), y, 1.0), y);
    res[192 + threadIdx.x] = y;
}

int main() {
    unsigned long long* out; double* sink; double* res;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 64 * 8); hipMalloc(&res, 256 * 8);
    hipMemset(out, 0, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        k_fma<<<1, 64>>>(out, sink, 1.0); k_mul<<<1, 64>>>(out, sink, 1.0); k_rsq<<<1, 64>>>(out, sink, 3.0); k_fma_indep<<<1, 64>>>(out, sink, 1.0);
        k_readlane<<<1, 64>>>(out, sink, 1.0); k_lds<<<1, 64>>>(out, sink, 1.0); k_mfma<<<1, 64>>>(out, sink, 1.0); k_mfma_acc<<<1, 64>>>(out, sink, 1.0);
        k_rsqnr<<<1, 64>>>(out, sink, 3.0); k_rsqc<<<1, 64>>>(out, sink, 3.0);
        hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, out, 16 * 8, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f64            %.1f cycles per instruction\n", h[0] / (double)N);
    printf("dependent v_mul_f64            %.1f\n", h[1] / (double)N);
    printf("dependent v_rsq_f64            %.1f\n", h[2] / (double)N);
    printf("independent v_fma_f64 (8-way)  %.1f cycles per instruction (issue)\n", h[3] / (double)N);
    printf("fma -> readlane x2 -> fma      %.1f cycles per round\n", h[4] / (double)N);
    printf("ds_write -> ds_read -> fma     %.1f cycles per round\n", h[5] / (double)N);
    printf("mfma_f64 -> fma(c) -> mfma     %.1f cycles per round\n", h[6] / (double)(N / 4));
    printf("mfma_f64 accumulator chain     %.1f cycles per mfma\n", h[7] / (double)(N / 4));
    printf("rsqrt: rsq + 2 Newton + add    %.1f cycles per round\n", h[8] / (double)(N / 4));
    printf("rsqrt: rsq + coupled + add     %.1f cycles per round\n", h[9] / (double)(N / 4));
    k_acc<<<1, 64>>>(res);
    double r[256];
    hipMemcpy(r, res, 256 * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int t = 0; t < 64; ++t) {
        const long double d = 0.001L + 0.37L * t + 1e-3L * t * t;
        const long double ex = 1.0L / sqrtl(d);
        auto rel = [&](double v) { long double e = (v - ex) / ex; return (double)(e < 0 ? -e : e); };
        e0 = fmax(e0, rel(r[t])); e1 = fmax(e1, rel(r[64 + t])); e2 = fmax(e2, rel(r[128 + t])); e3 = fmax(e3, rel(r[192 + t]));
    }
    printf("max relative error: v_rsq_f64 %.2e   + 1 Newton %.2e   + 2 Newton %.2e   coupled %.2e\n", e0, e3, e1, e2);
    return 0;
}
