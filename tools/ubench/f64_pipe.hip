// Microbenchmark: do fp64 VALU instructions and fp64 MFMA share an execution pipe on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_pipe.hip -o build/f64_pipe && build/f64_pipe
// Each kernel runs one workgroup per CU; waves are MFMA-only, FMA-only (fp64 or int) or mixed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: all waves MFMA; 1: all waves f64 FMA; 2: waves 0-3 MFMA, 4-7 f64 FMA;
                     // 3: waves 0-3 MFMA, 4-7 int mul; 4: one stream, 1 MFMA : 4 f64 FMA
                     // 5: one stream, 1 MFMA : 4 int mad
__global__ void __launch_bounds__(512, 2) k(double* out, int iters, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    double a = threadIdx.x * 1e-3, b = 1.0000001, c0 = 0.1, c1 = 0.2, c2 = 0.3, c3 = 0.4;
    unsigned u0 = threadIdx.x, u1 = 77, u2 = 5, u3 = 9;
    v4f64 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const bool do_mfma = (MODE == 0) || ((MODE == 2 || MODE == 3) && wave < 4) || MODE >= 4;
    const bool do_fma = (MODE == 1) || (MODE == 2 && wave >= 4);
    const bool do_int = (MODE == 3 && wave >= 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE >= 4) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                if (MODE == 4) { c0 = c0 * b + a; c1 = c1 * b + a; c2 = c2 * b + a; c3 = c3 * b + a; }
                else { u0 = u0 * u1 + u2; u1 = u1 * u2 + u3; u2 = u2 * u3 + u0; u3 = u3 * u0 + u1; }
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                if (MODE == 4) { c0 = c0 * b + a; c1 = c1 * b + a; c2 = c2 * b + a; c3 = c3 * b + a; }
                else { u0 = u0 * u1 + u2; u1 = u1 * u2 + u3; u2 = u2 * u3 + u0; u3 = u3 * u0 + u1; }
            }
        }
    } else if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
            }
        }
    } else if (do_fma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { c0 = c0 * b + a; c1 = c1 * b + a; c2 = c2 * b + a; c3 = c3 * b + a; }
        }
    } else if (do_int) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { u0 = u0 * u1 + u2; u1 = u1 * u2 + u3; u2 = u2 * u3 + u0; u3 = u3 * u0 + u1; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + c0 + c1 + c2 + c3 + u0 + u1 + u2 + u3;
}

template <int MODE>
void run(const char* name, double* out, unsigned long long* cyc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[b * 8 + w];
    lo /= 256 * 4; hi /= 256 * 4;
    printf("%-44s %8.3f ms  waves0-3 %10.0f ticks/iter %7.1f | waves4-7 %10.0f ticks/iter %7.1f\n",
           name, ms, lo, lo / iters, hi, hi / iters);
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 20000;
    // per iteration: MFMA waves issue 8 MFMAs (8*64 = 512 pipe cycles); FMA waves issue 32 f64 FMAs
    run<0>("0: 8 waves MFMA (2/SIMD), 8 MFMA/iter", out, cyc, iters);
    run<1>("1: 8 waves f64 FMA, 32 FMA/iter", out, cyc, iters);
    run<2>("2: waves0-3 MFMA | waves4-7 f64 FMA", out, cyc, iters);
    run<3>("3: waves0-3 MFMA | waves4-7 int mad", out, cyc, iters);
    run<4>("4: each wave: 8 MFMA + 32 f64 FMA interleaved", out, cyc, iters);
    run<5>("5: each wave: 8 MFMA + 32 int mad interleaved", out, cyc, iters);
    return 0;
}
