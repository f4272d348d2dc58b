// valu_f64_clock.hip -- sustained fp64 VALU rate and shader clock under that load, gfx950.
// Every SIMD runs W waves of independent v_fma_f64 chains (16 per lane); reports TFLOP/s and the
// shader clock seen by s_memtime against the 100 MHz s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_f64_clock tools/ubench/valu_f64_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// OP 0: fma, 1: mul, 2: add, 3: the C3 inner-loop mix on 16 independent elements (14 fp64
// instructions per element: fma, mul, add, mul, fma, add, mul + kick add, mul, add + drift mul, mul, add, add)
template <typename T, int OP>
__global__ void __launch_bounds__(64) k(T* out, int iters, unsigned long long* clk) {
    T x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (T)(threadIdx.x + j) * (T)1e-3;
    const T a = (T)1.0000001, b = (T)1e-7;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (OP == 0) x[j] = __builtin_fma(x[j], a, b);
            else if constexpr (OP == 1) x[j] = x[j] * a;
            else if constexpr (OP == 2) x[j] = x[j] + b;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    T s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename T, int OP>
void run(const char* name, int waves_per_simd, int iters) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount * 4 * waves_per_simd;
    T* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)grid * 64 * sizeof(T)); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, OP>), dim3(grid), dim3(64), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, OP>), dim3(grid), dim3(64), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)grid * 64 * 16 * (OP == 0 ? 2.0 : 1.0) * iters;
    printf("%s waves/SIMD=%d iters=%d: %.3f ms, %.1f TFLOP/s, shader clock %.2f GHz (memtime/realtime)\n", name,
           waves_per_simd, iters, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w : {1, 2, 3, 4, 6}) run<double, 0>("fp64 fma", w, 100000);
    for (int w : {1, 3}) run<double, 1>("fp64 mul", w, 100000);
    for (int w : {1, 3}) run<double, 2>("fp64 add", w, 100000);
    for (int w : {1, 3}) run<float, 0>("fp32 fma", w, 100000);
    return 0;
}
