// pbbi_rng.h -- device side of the RNG contract in include/pbbi.h (Philox-4x32-10).
// The integer part is bit-identical to oracle/pbbi_oracle.c::philox4x32_10.
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
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
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

// Both Box-Muller branches of one block: zc for the dim with bit 2 clear, zs for bit 2 set.
__device__ __forceinline__ void rng_normal_pair(uint64_t seed, uint32_t stream, uint64_t iter,
                                                uint64_t chain, uint32_t blk, double& zc,
                                                double& zs) {
    const PhiloxOut x = rng_block(seed, stream, iter, chain, blk);
    const double u1 = (double)(((((uint64_t)x.x1 << 32) | x.x0) >> 11) + 1) * 0x1.0p-53;
    const double u2 = u53(x.x2, x.x3);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    zc = r * c;
    zs = r * s;
}

__device__ __forceinline__ double rng_normal(uint64_t seed, uint32_t stream, uint64_t iter,
                                             uint64_t chain, int dim) {
    double zc, zs;
    rng_normal_pair(seed, stream, iter, chain, (uint32_t)(((dim >> 3) << 2) | (dim & 3)), zc, zs);
    return ((dim >> 2) & 1) ? zs : zc;
}

__device__ __forceinline__ double rng_uniform(uint64_t seed, uint64_t iter, uint64_t chain) {
    const PhiloxOut x = rng_block(seed, /*PBBI_STREAM_UNIFORM*/ 2u, iter, chain, 0xFFFFFFFFu);
    return u53(x.x0, x.x1);
}
