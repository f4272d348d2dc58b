// pbbi_buf.h -- buffer-descriptor (SRSRC) access to the (D, N) state arrays, gfx950.
//
// Element (d, n) of a state array lives at base + 8*(d*ld + n).  With flat/global addressing
// hipcc materialises one 64-bit VGPR address PER ROW and keeps all of them live from the
// first load to the last store (2*D VGPRs per array: measured 100+ VGPRs of pure addresses
// in the register-resident kernels).  A raw buffer access splits the address into
//     SGPR descriptor (tile base, wave-uniform)  +  ONE 32-bit per-lane byte offset (VGPR)
//     + a wave-uniform row offset (SGPR "soffset", d*ld*8)
// so every row of every array shares a single VGPR.  Offsets are 32-bit: callers guarantee
// D*ld*8 < 2^32 (checked on the host).  The descriptor must be built from wave-uniform values
// only (kernel arguments, blockIdx, readfirstlane results).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int pbbi_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_make(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0,
                                             /*num_records*/ 0xFFFFFFF0u, /*flags*/ 0x00020000);
}

// Bounded descriptor: num_records = the bytes that remain from `base` to the end of a (D, N) array.
// gfx950 range-checks voffset + soffset against it (tools/ubench/buffer_oob.hip): a load past the
// end returns 0 and a store past the end is dropped, so the rows d >= D of a zero-padded chain need
// neither a branch, a clamp nor a sink.  `base` = array + n0 (first chain of the block).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_make_rows(const void* base, int D, int64_t ld,
                                                                int64_t N, int64_t n0, int elem) {
    const int64_t bytes = ((int64_t)(D - 1) * ld + N - n0) * elem;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0,
                                             (unsigned)(bytes > 0 ? bytes : 0), /*flags*/ 0x00020000);
}

template <typename T>
__device__ __forceinline__ T buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes,
                                      uint32_t soff_bytes);
template <>
__device__ __forceinline__ double buf_load<double>(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes,
                                                   uint32_t soff_bytes) {
    return __builtin_bit_cast(double, (pbbi_u32x2)__builtin_amdgcn_raw_buffer_load_b64(
                                          r, voff_bytes, soff_bytes, 0));
}
template <>
__device__ __forceinline__ float buf_load<float>(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes,
                                                 uint32_t soff_bytes) {
    return __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(
                                         r, voff_bytes, soff_bytes, 0));
}

__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes,
                                          uint32_t soff_bytes, double x) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pbbi_u32x2, x), r, voff_bytes,
                                          soff_bytes, 0);
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, uint32_t voff_bytes,
                                          uint32_t soff_bytes, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), r, voff_bytes,
                                          soff_bytes, 0);
}
