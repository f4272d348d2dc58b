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
// Flag PBBI_DRAW_F64 selects the double-precision draw further down instead (bit-identical to the oracle).
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

// The four standard normals of one block: z[slot], slot = (dim >> 2) & 3 of the dims
// {16*(blk>>2) + (blk&3) + 4*slot}.
__device__ __forceinline__ void rng_normal4(uint64_t seed, uint32_t stream, uint64_t iter,
                                            uint64_t chain, uint32_t blk, float (&z)[4]) {
    const PhiloxOut x = rng_block(seed, stream, iter, chain, blk);
    box_muller_f32(x.x0, x.x1, z[0], z[1]);
    box_muller_f32(x.x2, x.x3, z[2], z[3]);
}

// ---- PBBI_DRAW_F64: the double-precision draw of the RNG contract (include/pbbi.h) -----------------
// Built from +, -, *, /, sqrt and fma in a fixed order -- no transcendental unit, no library call whose
// rounding is the vendor's business -- so that oracle/pbbi_oracle.c, which restates the same steps in C,
// produces the SAME BITS on the host.  Two Philox blocks per four normals (blk for slots 0, 1;
// blk | 0x80000000 for slots 2, 3); u1 carries 52 random bits (tails to 8.57 sigma), the angle 53.
__device__ __forceinline__ double draw_log_unit(double u) {  // ln u, u a normal double in (0, 1]
    constexpr double ln2_hi = 0x1.62e42fee00000p-1, ln2_lo = 0x1.a39ef35793c76p-33,
                     Lg1 = 0x1.5555555555593p-1, Lg2 = 0x1.999999997fa04p-2, Lg3 = 0x1.2492494229359p-2,
                     Lg4 = 0x1.c71c51d8e78afp-3, Lg5 = 0x1.7466496cb03dep-3, Lg6 = 0x1.39a09d078c69fp-3,
                     Lg7 = 0x1.2f112df3e5244p-3;  // fdlibm e_log.c
    const uint64_t bits = (uint64_t)__double_as_longlong(u);
    int k = (int)(bits >> 52) - 1023;
    const uint64_t mant = bits & 0xFFFFFFFFFFFFFull;
    const bool up = mant > 0x6A09E667F3BCCull;  // m > sqrt(2): halve it
    k += up ? 1 : 0;
    const double m = __longlong_as_double((long long)(mant | ((up ? 1022ull : 1023ull) << 52)));
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

__device__ __forceinline__ void draw_sincos_quarter(double y, double& sn, double& cs) {  // of (pi/2) y
    // Taylor coefficients (pi/2)^j / j! rounded from 60-digit decimals (tools/gen_draw_coeffs.py)
    constexpr double S[9] = {0x1.921fb54442d18p+0, -0x1.4abbce625be53p-1, 0x1.466bc6775aae2p-4,
                             -0x1.32d2cce62bd86p-8, 0x1.50783487ee782p-13, -0x1.e3074fde8871fp-19,
                             0x1.e8f434d018d63p-25, -0x1.6fadb9f155744p-31, 0x1.aaec32af93359p-38};
    constexpr double C[10] = {0x1.0000000000000p+0, -0x1.3bd3cc9be45dep+0, 0x1.03c1f081b5ac4p-2,
                              -0x1.55d3c7e3cbffap-6, 0x1.e1f506891babbp-11, -0x1.a6d1f2a204a8cp-16,
                              0x1.f9d38a3763cc3p-22, -0x1.b6e24f44b128fp-28, 0x1.20c62c2f2d7f5p-34,
                              -0x1.2a0c591af8314p-41};
    const double z = y * y;
    double s = S[8], c = C[9];
#pragma unroll
    for (int k = 7; k >= 0; --k) s = fma(s, z, S[k]);
#pragma unroll
    for (int k = 8; k >= 0; --k) c = fma(c, z, C[k]);
    sn = y * s;
    cs = c;
}

__device__ __forceinline__ void box_muller_f64(const PhiloxOut& x, double& zc, double& zs) {
    const uint64_t w1 = ((uint64_t)x.x1 << 32) | x.x0, w2 = ((uint64_t)x.x3 << 32) | x.x2;
    const double u1 = ((double)(w1 >> 12) + 0.5) * 0x1.0p-52;
    const uint64_t k2 = w2 >> 11;
    const uint64_t n = (k2 + (1ull << 50)) >> 51;   // quadrant: angle = (pi/2)(n + y)
    const double y = (double)((long long)k2 - (long long)(n << 51)) * 0x1.0p-51;
    double sn, cs;
    draw_sincos_quarter(y, sn, cs);
    const double r = __builtin_sqrt(-2.0 * draw_log_unit(u1));
    const uint32_t q = (uint32_t)n & 3u;
    const double c = (q & 1u) ? sn : cs, s = (q & 1u) ? cs : sn;   // |cos|, |sin| of the rotated angle
    zc = r * ((q == 1u || q == 2u) ? -c : c);
    zs = r * ((q >= 2u) ? -s : s);
}

// The four standard normals of block blk as doubles, by either draw (f64: wave-uniform).
__device__ __forceinline__ void rng_normal4d(uint64_t seed, uint32_t stream, uint64_t iter, uint64_t chain,
                                             uint32_t blk, bool f64, double (&z)[4]) {
    if (f64) {
        box_muller_f64(rng_block(seed, stream, iter, chain, blk), z[0], z[1]);
        box_muller_f64(rng_block(seed, stream, iter, chain, blk | 0x80000000u), z[2], z[3]);
    } else {
        float zf[4];
        rng_normal4(seed, stream, iter, chain, blk, zf);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = (double)zf[i];
    }
}

__device__ __forceinline__ uint32_t rng_block_of_dim(int dim) {
    return (uint32_t)(((dim >> 4) << 2) | (dim & 3));
}

__device__ __forceinline__ double rng_normal(uint64_t seed, uint32_t stream, uint64_t iter,
                                             uint64_t chain, int dim, bool f64 = false) {
    const int slot = (dim >> 2) & 3;
    if (f64) {
        double a, b;
        box_muller_f64(rng_block(seed, stream, iter, chain,
                                 rng_block_of_dim(dim) | (slot >= 2 ? 0x80000000u : 0u)), a, b);
        return (slot & 1) ? b : a;
    }
    float z[4];
    rng_normal4(seed, stream, iter, chain, rng_block_of_dim(dim), z);
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
