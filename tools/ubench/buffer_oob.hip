// buffer_oob.hip -- does a raw buffer store past num_records get dropped on gfx950, and which
// offsets take part in the range check?  All target addresses stay inside one 8 KiB allocation,
// so a store that is NOT dropped is visible (and harmless).
//   hipcc --offload-arch=gfx950 -O3 -o build/buffer_oob tools/ubench/buffer_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __amdgpu_buffer_rsrc_t mk(void* base, unsigned num_records) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, num_records, 0x00020000);
}
__global__ void k(double* buf, double* loaded) {
    const unsigned lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = mk(buf, 1024);  // 128 doubles are "in range"
    const double one = 1.0, two = 2.0, three = 3.0;
    // (a) voffset out of range, soffset 0            -> target buf[128 + lane]
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, one), r, 1024 + lane * 8, 0, 0);
    // (b) voffset in range, soffset pushes it out     -> target buf[256 + lane]
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, two), r, lane * 8, 2048, 0);
    // (c) voffset huge (0xFFFFF000 + lane*8) with a soffset that wraps it back inside -> buf[384 + lane]
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, three), r, 0xFFFFF000u + lane * 8,
                                          0x1000u + 3072, 0);
    // loads: out-of-range voffset must return 0
    loaded[lane] = __builtin_bit_cast(double, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(r, 1024 + lane * 8, 0, 0));
    loaded[64 + lane] = __builtin_bit_cast(double, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, 2048, 0));
}
int main() {
    double *buf, *loaded, h[1024], hl[128];
    (void)hipMalloc(&buf, 8192); (void)hipMalloc(&loaded, 1024);
    for (int i = 0; i < 1024; ++i) h[i] = -7.0;
    (void)hipMemcpy(buf, h, 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, loaded);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, buf, 8192, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hl, loaded, 1024, hipMemcpyDeviceToHost);
    printf("(a) voffset >= num_records, soffset 0      : buf[128] = %g  (-7 = dropped)\n", h[128]);
    printf("(b) voffset in range, soffset out of range : buf[256] = %g  (-7 = dropped)\n", h[256]);
    printf("(c) voffset huge, soffset wraps it inside  : buf[384] = %g  (-7 = dropped)\n", h[384]);
    printf("load (a): %g   load (b): %g   (0 = out-of-range load returns zero, -7 = it read memory)\n", hl[0], hl[64]);
    return 0;
}
