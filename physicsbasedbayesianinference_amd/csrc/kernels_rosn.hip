// kernels_rosn.hip -- Rosenbrock potential, 16 < D <= 256, Leapfrog in the PBBI_KDK_FMA form: a chain's
// dimensions are cut into 16-dim PARTS held by different WAVES of one workgroup (the layout of
// kernels_sepn.hip), gfx950.
//
// The nearest-neighbour coupling crosses a part boundary twice per gradient evaluation: part w needs
// q_{16w+16} (first element of part w+1) for its t_15, and t_{16w-1} of part w-1 for the term carried
// into its g_0.  Every part publishes its first and last position through LDS (double-buffered),
// ONE __syncthreads per evaluation, and recomputes the neighbour's t from the two boundary values.
// The two Hamiltonians ride on the first and the last gradient evaluation (same t values), and the
// parts' shares of oldH - newH are summed through LDS like in kernels_sepn.hip.
//
// Per element-step: t, c1*q, the (a - q) term, two fmas into v_j, one into v_{j+1}, one drift fma
// = 7 fp64 instructions (kernels_lane2.hip's kdk_kick), state q and the half-step velocity only.
// Algebraically src/integrator.py:105-120; q, p within 1e-12 of the oracle, accept masks equal
// (tests/test_gpu_parity.py::test_rosenbrock_multiwave_kdk).  Taken only under PBBI_KDK_FMA, and for
// D > 32 (the two-lane kernel of kernels_lane2.hip is faster at D <= 32).
#include <cstdlib>

#include "pbbi_buf.h"
#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

constexpr int DL = 16;    // dims per wave
constexpr int MAXG = 16;  // waves per workgroup (D <= 256)

struct RosPrm {
    const double* q_in;
    const double* p_in;
    const double* u_in;
    const double* mass;
    double* q_out;
    double* p_out;
    double* ratio_out;
    uint8_t* reject_out;
    int64_t N, ldn_in, ldn_out;
    double h, a, b, inv_s, kT, c1, c2, c3;  // c1 = (-4b)/s, c2 = 2/s, c3 = (2b)/s
    int L, D, flags, rng;
    uint64_t seed, iter, chain0;
};

// GMAX: upper bound on the waves per workgroup of this instantiation (4: D <= 64, 256-thread blocks
// may use 168 VGPRs at three waves per SIMD; 16: D <= 256, 128 VGPRs)
template <bool UNIT, bool FULL, int GMAX>
__global__ void __launch_bounds__(64 * GMAX, (GMAX <= 4 ? 3 : 1)) k_rosn_hmc(RosPrm prm) {
    extern __shared__ double smem_ros[];  // edge[2][G][2][64] then dH[G][64]
    const int c = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform
    const int G = (int)(blockDim.x >> 6);
    double* edge = smem_ros;
    double* dH = smem_ros + 4 * G * 64;
    const int64_t n0 = (int64_t)blockIdx.x * 64;  // block-uniform
    const int64_t left = prm.N - n0;
    const bool valid = c < left;
    const int cc = valid ? c : (int)left - 1;
    const int D = prm.D;
    const int d0 = DL * part;
    const double m = UNIT ? 1.0 : prm.mass[n0 + cc];
    const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);
    const uint32_t rin = 8u * (uint32_t)prm.ldn_in, rout = 8u * (uint32_t)prm.ldn_out;
    const uint32_t vo = 8u * (uint32_t)cc;
    // descriptors bounded to the array (pbbi_buf.h::buf_make_rows): rows past D read 0 / drop stores
    const __amdgpu_buffer_rsrc_t bq = buf_make_rows(prm.q_in + n0 + (int64_t)d0 * prm.ldn_in, D - d0, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bp = buf_make_rows(prm.p_in + n0 + (int64_t)d0 * prm.ldn_in, D - d0, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bqo = buf_make_rows(prm.q_out + n0 + (int64_t)d0 * prm.ldn_out, D - d0, prm.ldn_out, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bpo = buf_make_rows(prm.p_out + n0 + (int64_t)d0 * prm.ldn_out, D - d0, prm.ldn_out, prm.N, n0, 8);
    auto exists = [&](int j) { return FULL || d0 + j < D; };       // wave-uniform
    auto has_next = [&](int j) {                                    // dim d0+j has a right neighbour
        if constexpr (FULL) return j + 1 < DL ? true : part + 1 < G;
        return d0 + j + 1 < D;
    };
    auto ld = [&](__amdgpu_buffer_rsrc_t r, int j) { return buf_load<double>(r, vo, (uint32_t)j * rin); };
    auto st = [&](__amdgpu_buffer_rsrc_t r, int j, double x) { buf_store(r, vo, (uint32_t)j * rout, x); };

    double q[DL], v[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        const double x = ld(bq, j);
        q[j] = exists(j) ? x : 0.0;
    }
    const double pstd = prm.rng ? sqrt(m * prm.kT) : 1.0;  // src/ensemble.py:88
    auto draw = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // this part's group of 16 dims: blocks (part<<2)|r
            double z[4];
            rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, prm.iter, chain, (uint32_t)((part << 2) | r), (prm.flags & PBBI_DRAW_F64) != 0, z);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) v[r + 4 * sl] = exists(r + 4 * sl) ? z[sl] * pstd : 0.0;
        }
    };
    auto load_p = [&]() {
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double x = ld(bp, j);
            v[j] = exists(j) ? x : 0.0;
        }
    };
    if (prm.rng) draw(); else load_p();
    auto pp_part = [&]() {
        double pp = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) pp = fma(v[j], v[j], pp);
        return pp;
    };
    const double pp_old = pp_part();

    // one gradient evaluation: v_j += kk * (-g_j); returns this part's share of sum b t^2 + sum (a-q)^2
    const double nc1 = -prm.c1, nc2 = -prm.c2, c2a = prm.c2 * prm.a, nc3 = -prm.c3;
    int buf = 0;
    auto kick = [&](double kk) {
        buf ^= 1;
        double* e = edge + (size_t)buf * 2 * G * 64;
        e[(part * 2 + 0) * 64 + c] = q[0];
        e[(part * 2 + 1) * 64 + c] = q[DL - 1];
        __syncthreads();
        const double q_ext = (part + 1 < G) ? e[((part + 1) * 2 + 0) * 64 + c] : 0.0;
        const double q_prev = (part > 0) ? e[((part - 1) * 2 + 1) * 64 + c] : 0.0;
        const double kn3 = kk * nc3;
        // term carried in from the previous part: c3 * t_{d0-1}, t_{d0-1} = q_{d0} - q_{d0-1}^2
        const double t_prev = fma(-q_prev, q_prev, q[0]);
        const double v0 = fma(kn3, t_prev, v[0]);
        v[0] = (part > 0 && exists(0)) ? v0 : v[0];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            const double t = fma(-q[j], q[j], qn);
            const double r = prm.a - q[j];
            const double nfirst = fma(nc1 * q[j], t, fma(nc2, q[j], c2a));
            const bool hn = has_next(j);
            const double vj = fma(nfirst, kk, v[j]);
            v[j] = hn ? vj : v[j];
            if (j + 1 < DL) {
                const double vn = fma(kn3, t, v[(j + 1) & (DL - 1)]);
                v[(j + 1) & (DL - 1)] = hn ? vn : v[(j + 1) & (DL - 1)];
            }
            const double n1 = fma(prm.b * t, t, s1), n2 = fma(r, r, s2);
            s1 = hn ? n1 : s1;
            s2 = hn ? n2 : s2;
        }
        return s1 + s2;
    };

    // ---- Leapfrog, kick-drift-kick: vh = v + a0 h/2;  L x { q += vh h; vh += a(q) h }, last kick half
    const double h = prm.h, hm = UNIT ? h : h / m, hhm = 0.5 * hm;
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] / m;
    }
    double u_old = 0.0, u_new = 0.0;
    if (prm.L > 0) {
        u_old = kick(hhm);
        for (int s = 0; s < prm.L; ++s) {
#pragma unroll
            for (int j = 0; j < DL; ++j) q[j] = fma(v[j], h, q[j]);
            u_new = kick((s + 1 < prm.L) ? hm : hhm);
        }
    }
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] * m;  // p = v*m
    }
    // this part's share of oldH - newH, then the sum over the parts (the same value in every wave)
    const double dpart = prm.L > 0 ? 0.5 * (pp_old - pp_part()) / m + (u_old - u_new) * prm.inv_s : 0.0;
    dH[part * 64 + c] = dpart;
    __syncthreads();
    double dsum = 0.0;
    for (int g = 0; g < G; ++g) dsum += dH[g * 64 + c];
    const double ratio = exp(dsum * pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const double u = prm.rng ? rng_uniform(prm.seed, prm.iter, chain) : prm.u_in[n0 + cc];
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    if (reject) {
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] = ld(bq, j);  // :175
        if (prm.p_out) {
            if (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) {  // :176  p <- oldQ
#pragma unroll
                for (int j = 0; j < DL; ++j) v[j] = q[j];
            } else if (prm.rng) {
                draw();
            } else {
                load_p();
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < DL; ++j) st(bqo, j, q[j]);
        if (prm.p_out) {
#pragma unroll
            for (int j = 0; j < DL; ++j) st(bpo, j, v[j]);
        }
        if (part == 0) {
            if (prm.ratio_out) prm.ratio_out[n0 + c] = ratio;
            if (prm.reject_out) prm.reject_out[n0 + c] = reject ? 1 : 0;
        }
    }
}

}  // namespace

// true if this path takes the call: Rosenbrock, fp64, Leapfrog, PBBI_KDK_FMA, 32 < D <= 256
// (PBBI_ROSN_MIN_D lowers the bound for A/B runs against the two-lane kernel)
bool rosn_applies(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    static const int min_d = getenv("PBBI_ROSN_MIN_D") ? atoi(getenv("PBBI_ROSN_MIN_D")) : 33;
    return pot->kind == KIND_ROSENBROCK && pot->dtype == PBBI_F64 && a.method == PBBI_LEAPFROG &&
           (a.flags & PBBI_KDK_FMA) != 0 && pot->D >= min_d && pot->D > 16 && pot->D <= DL * MAXG &&
           (int64_t)DL * (a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) < ((int64_t)1 << 28);
}

int rosn_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (a.N == 0) return PBBI_OK;
    const double inv_s = 1.0 / pot->s;
    RosPrm prm{(const double*)a.q_in, (const double*)a.p_in, (const double*)a.u_in,
               (const double*)a.mass, (double*)a.q_out, (double*)a.p_out, (double*)a.ratio_out,
               a.reject_out, a.N, a.ldn_in, a.ldn_out, a.h, pot->a, pot->b, inv_s, a.kT,
               (-4.0 * pot->b) * inv_s, 2.0 * inv_s, (2.0 * pot->b) * inv_s, a.L, pot->D, a.flags,
               a.rng, a.seed, a.iter, a.chain0};
    const int G = (pot->D + DL - 1) / DL;
    const dim3 grid((unsigned)((a.N + 63) / 64)), block(64 * G);
    const size_t lds = (size_t)(4 * G + G) * 64 * sizeof(double);
    const bool full = (pot->D % DL == 0);
#define ROSN_LAUNCH(U_, F_)                                                                          \
    {                                                                                                \
        if (G <= 4) hipLaunchKernelGGL((k_rosn_hmc<U_, F_, 4>), grid, block, lds, a.stream, prm);    \
        else hipLaunchKernelGGL((k_rosn_hmc<U_, F_, MAXG>), grid, block, lds, a.stream, prm);        \
    }
    if (a.mass) {
        if (full) ROSN_LAUNCH(false, true) else ROSN_LAUNCH(false, false)
    } else {
        if (full) ROSN_LAUNCH(true, true) else ROSN_LAUNCH(true, false)
    }
#undef ROSN_LAUNCH
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}
