// int_mul_rate.hip -- issue cost of the 32-bit multiplies Philox is made of, gfx950: v_mul_lo_u32 +
// v_mul_hi_u32 (two instructions for the 64-bit product) against ONE v_mad_u64_u32, with v_fma_f64 and
// v_xor_b32 for scale.  W waves per SIMD, 16 independent chains per lane; prints cycles per instruction
// per SIMD (s_memtime).   hipcc --offload-arch=gfx950 -O3 -o build/int_mul_rate tools/ubench/int_mul_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(64) k(unsigned* out, int iters, unsigned long long* clk) {
    unsigned x[16];
    unsigned long long y[16];
    double z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { x[j] = threadIdx.x * 2654435761u + j; y[j] = x[j]; z[j] = x[j] * 1e-9; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (OP == 0) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "v"(0xD2511F53u));
            if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(0xD2511F53u));
            if constexpr (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(y[j]) : "v"((unsigned)y[j]), "v"(0xD2511F53u) : "vcc");
            if constexpr (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[j]) : "v"(0xD2511F53u));
            if constexpr (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(z[j]) : "v"(1.0000001));
            if constexpr (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(0xD2511F53u));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += x[j] + (unsigned)y[j] + (unsigned)z[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP>
void run(const char* name, int w) {
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount * 4 * w, iters = 20000;
    unsigned* out; unsigned long long* clk;
    (void)hipMalloc(&out, (size_t)grid * 64 * 4); (void)hipMalloc(&clk, 16);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(64), 0, 0, out, iters, clk);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(64), 0, 0, out, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    // a wave sees h cycles for its iters*16 instructions; the SIMD issued w times as many in the launch
    printf("%-14s waves/SIMD=%d: wave 0 sees %.2f cycles per own instruction; launch %.3f ms = %.2f ns per "
           "instruction per SIMD\n", name, w, (double)h / ((double)iters * 16), ms, ms * 1e6 / ((double)iters * 16 * w));
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    for (int w : {1, 2, 3, 4}) {
        run<0>("v_mul_hi_u32", w); run<1>("v_mul_lo_u32", w); run<2>("v_mad_u64_u32", w);
        run<3>("v_xor_b32", w); run<5>("v_add_u32", w); run<4>("v_fma_f64", w);
    }
    return 0;
}
