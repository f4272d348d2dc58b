// pbbi_rng.h -- device side of the RNG contract in include/pbbi.h.
//
// Counter-based Philox-4x32-10 (bit-identical to oracle/pbbi_oracle.c::philox4x32_10) followed
// by a SINGLE-PRECISION Box-Muller transform on the hardware transcendental unit (v_log_f32,
// v_sqrt_f32, v_sin_f32, v_cos_f32): one 128-bit block yields FOUR standard normals for
// ~90 instructions.  Why not fp64 log/sincospi: on gfx950 an f64 MFMA stream leaves room for
// only ~3 other vector instructions per MFMA (tools/ubench/f64_pipe.hip), so every VALU
// instruction in the fused HMC kernel is paid for in matrix-pipe time; the fp64 library
// transform costs ~400 instructions per PAIR of normals and was 75 % of the kernel's
// vector work.  The draws are exact N(0,1) variates on a 2^-24-relative grid with tails to
// 6.7 sigma -- statistically indistinguishable for momentum refreshment -- and are
// reproducible bit for bit on the device; the host mirror in the oracle agrees to ~1e-6.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PhiloxOut {
    uint32_t x0, x1, x2, x3;
};

__device__ __forceinline__ PhiloxOut philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                   uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one v_mad_u64_u32 per 64-bit product: v_mul_hi_u32 + v_mul_lo_u32 cost as much EACH
        // (~2.2 ns per instruction and SIMD, like an fp64 FMA; tools/ubench/int_mul_rate.hip)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return PhiloxOut{c0, c1, c2, c3};
}

__device__ __forceinline__ PhiloxOut rng_block(uint64_t seed, uint32_t stream, uint64_t iter,
                                               uint64_t chain, uint32_t blk) {
    return philox4x32_10((uint32_t)chain, blk, (uint32_t)iter,
                         (stream & 0xFFu) | ((uint32_t)(chain >> 32) << 8), (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

__device__ __forceinline__ double u53(uint32_t lo, uint32_t hi) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53;
}

// (a, b) -> (r cos(2 pi u2), r sin(2 pi u2)), r = sqrt(-2 ln u1):
//   u1 = a*2^-32 + 2^-33 in (0, 1]  (float: the tail keeps its full 32-bit resolution)
//   u2 = (b >> 8) * 2^-24 in [0, 1)
__device__ __forceinline__ void box_muller_f32(uint32_t a, uint32_t b, float& zc, float& zs) {
    const float u1 = fmaf((float)a, 0x1.0p-32f, 0x1.0p-33f);
    const float u2 = (float)(b >> 8) * 0x1.0p-24f;
    // -2 ln(u1) = (-2 ln 2) * log2(u1);  v_log_f32 is log2, v_sin/v_cos take revolutions
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    zc = r * __builtin_amdgcn_cosf(u2);
    zs = r * __builtin_amdgcn_sinf(u2);
}

#ifdef PBBI_DRAW_F64
// DIAGNOSTIC build only (tools/build_variant.sh draw64 kernels_dense -- -DPBBI_DRAW_F64): what a
// double-precision draw would cost -- 53-bit uniforms (two Philox blocks per four normals) through
// the fp64 library log / sqrt / sincospi.  Not part of the RNG contract, not mirrored by the oracle:
// only its THROUGHPUT is read.
__device__ __forceinline__ void box_muller_f64(const PhiloxOut& x, double& zc, double& zs) {
    const double u1 = ((double)((((uint64_t)x.x1 << 32) | x.x0) >> 11) + 0.5) * 0x1.0p-53;
    const double u2 = (double)((((uint64_t)x.x3 << 32) | x.x2) >> 11) * 0x1.0p-53;
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    zc = r * cs;
    zs = r * sn;
}
template <typename Z>
__device__ __forceinline__ void rng_normal4(uint64_t seed, uint32_t stream, uint64_t iter,
                                            uint64_t chain, uint32_t blk, Z (&z)[4]) {
    double a, b, c, d;
    box_muller_f64(rng_block(seed, stream, iter, chain, blk), a, b);
    box_muller_f64(rng_block(seed, stream, iter, chain, blk | 0x80000000u), c, d);
    z[0] = (Z)a; z[1] = (Z)b; z[2] = (Z)c; z[3] = (Z)d;
}
#else
// The four standard normals of one block: z[slot], slot = (dim >> 2) & 3 of the dims
// {16*(blk>>2) + (blk&3) + 4*slot}.
__device__ __forceinline__ void rng_normal4(uint64_t seed, uint32_t stream, uint64_t iter,
                                            uint64_t chain, uint32_t blk, float (&z)[4]) {
    const PhiloxOut x = rng_block(seed, stream, iter, chain, blk);
    box_muller_f32(x.x0, x.x1, z[0], z[1]);
    box_muller_f32(x.x2, x.x3, z[2], z[3]);
}
#endif

__device__ __forceinline__ uint32_t rng_block_of_dim(int dim) {
    return (uint32_t)(((dim >> 4) << 2) | (dim & 3));
}

__device__ __forceinline__ double rng_normal(uint64_t seed, uint32_t stream, uint64_t iter,
                                             uint64_t chain, int dim) {
    float z[4];
    rng_normal4(seed, stream, iter, chain, rng_block_of_dim(dim), z);
    const int slot = (dim >> 2) & 3;
    return (double)(slot == 0 ? z[0] : slot == 1 ? z[1] : slot == 2 ? z[2] : z[3]);
}

// PBBI_PER_CHAIN_STEPS: the chain's own number of leapfrog steps, 1 + floor(u * L) capped at L
__device__ __forceinline__ int rng_steps(uint64_t seed, uint64_t iter, uint64_t chain, int L) {
    const PhiloxOut x = rng_block(seed, /*PBBI_STREAM_STEPS*/ 3u, iter, chain, 0xFFFFFFFFu);
    const int s = 1 + (int)(u53(x.x0, x.x1) * (double)L);
    return s > L ? L : s;
}

__device__ __forceinline__ double rng_uniform(uint64_t seed, uint64_t iter, uint64_t chain) {
    const PhiloxOut x = rng_block(seed, /*PBBI_STREAM_UNIFORM*/ 2u, iter, chain, 0xFFFFFFFFu);
    return u53(x.x0, x.x1);
}
