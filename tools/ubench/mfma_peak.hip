// mfma_peak.hip -- sustained MFMA rate (register operands only, no memory) and the shader clock under
// that load: fp32 32x32x2 and fp64 16x16x4, 1 or 2 waves per SIMD.  The attainable ceiling for
// k_big_gemm / k_dense_hmc, to set beside the nominal 157.3 / 78.6 TFLOP/s (2.4 GHz) peaks.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_peak tools/ubench/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k32(float* out, int iters, unsigned long long* clk) {
    v16f acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 1.f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ void __launch_bounds__(64) k64(double* out, int iters, unsigned long long* clk) {
    v4d acc[4];
    for (int a = 0; a < 4; ++a) acc[a] = v4d{0, 0, 0, 0};
    const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    double s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename K, typename T>
void run(const char* name, K kern, T*, double flop_per_mfma, int waves, int iters) {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount * 4 * waves;
    T* out; unsigned long long* clk;
    (void)hipMalloc(&out, (size_t)grid * 64 * sizeof(T)); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%s waves/SIMD=%d: %.2f ms, %.1f TFLOP/s, shader clock %.2f GHz\n", name, waves, ms,
           (double)grid * 4.0 * iters * flop_per_mfma / ms / 1e9, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
    (void)hipFree(out); (void)hipFree(clk);
}
int main() {
    for (int w : {1, 2}) run("fp32 mfma 32x32x2 ", k32, (float*)nullptr, 32.0 * 32 * 2 * 2, w, 200000);
    for (int w : {1, 2}) run("fp64 mfma 16x16x4 ", k64, (double*)nullptr, 16.0 * 16 * 4 * 2, w, 200000);
    return 0;
}
