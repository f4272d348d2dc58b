// kernels_dense_dev.h -- device side of the register-resident dense-Gaussian kernels (gfx950): shared by
// kernels_dense.hip (P resident in LDS, D <= 128) and kernels_dstream.hip (P streamed through an LDS ring,
// 128 < D <= 256).  See kernels_dense.hip for the design notes.
#pragma once
#include <cstdlib>
#include <vector>

#include "pbbi_buf.h"
#include <type_traits>

#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int BLOCK = 256;          // 4 waves, one per SIMD
constexpr int CHAINS_PER_WAVE = 16;
constexpr int CHAINS_PER_WG = 64;

#ifdef PBBI_STAMPS
// Diagnostic build only (tools/build_stamps.sh): s_memtime stamps of the phases of
// k_dense_hmc, written to a buffer of their own (tools/stamp_probe.py).  Never quote
// the run time of such a build; read the SHARES.
static unsigned long long* g_stamp_buf = nullptr;
extern "C" void pbbi_debug_set_stamp_buffer(void* p) { g_stamp_buf = (unsigned long long*)p; }
// PBBI_STAMPS=1: every phase (s_memtime, shader cycles).  PBBI_STAMPS=2: only entry/exit, taken
// with s_memrealtime (100 MHz, one clock for the whole chip) for the workgroup timeline.
#if PBBI_STAMPS == 2
#define STAMP(i)                                                                      \
    do {                                                                              \
        if (((i) == 0 || (i) == 41) && prm.stamps && lane == 0) {                     \
            unsigned long long t_;                                                    \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            prm.stamps[((size_t)blockIdx.x * 8 + wave) * 64 + (i)] = t_;              \
        }                                                                             \
    } while (0)
#else
#define STAMP(i)                                                                      \
    do {                                                                              \
        if (prm.stamps && lane == 0) {                                                \
            __builtin_amdgcn_sched_barrier(0);                                        \
            unsigned long long t_;                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            prm.stamps[((size_t)blockIdx.x * 8 + wave) * 64 + (i)] = t_;              \
            __builtin_amdgcn_sched_barrier(0);                                        \
        }                                                                             \
    } while (0)
#endif
#else
#define STAMP(i) do { } while (0)
#endif

struct DensePrm {
#ifdef PBBI_STAMPS
    unsigned long long* stamps;
#endif
    const double* frag;   // DP*DP, A-fragment order
    const double* mu;     // DP, zero padded
    const double* q_in;
    const double* p_in;
    const double* u_in;
    const double* mass;
    double* q_out;
    double* p_out;
    double* v_out;
    double* ratio_out;
    uint8_t* reject_out;
    int64_t N, ldn_in, ldn_out;
    double h, cst, kT;
    int L, D, flags, rng, mode;  // mode 0: HMC iteration, 1: integrate only (in place)
    uint64_t seed, iter, chain0;
    const int32_t* steps_in;  // PBBI_PER_CHAIN_STEPS, uploaded mode (nullptr: L)
    int32_t* steps_out;
    // gradient carried across the iterations of a run (CARRY, see k_dense_hmc)
    double* carry_g;            // [2][D][N]
    uint8_t* carry_sel;         // [N]: which of the two slabs holds g at the chain's current position
    uint32_t carry_slab_bytes;  // D*N*8
    // several iterations of a run in one launch (FUSE, see k_dense_hmc; IterArgs::fuse_*)
    // fuse_first: iteration 0 of this launch is the FIRST of the run -- it forms g(q_0) itself and stores
    // it as slab 0 of the carried gradient (what a CARRY = 1 launch of its own used to do)
    int fuse_S, fuse_wrap2, fuse_first;
    int64_t fuse_slab0, fuse_slab;
    double* fuse_q_base;
};

template <int NT>
__device__ __forceinline__ void stage_lds(const double* __restrict__ gfrag,
                                          const double* __restrict__ gmu, v2f64* frag2,
                                          double* mu) {
    constexpr int DP = 16 * NT;
    const v2f64* src = reinterpret_cast<const v2f64*>(gfrag);
    for (int i = threadIdx.x; i < DP * DP / 2; i += BLOCK) frag2[i] = src[i];
    for (int i = threadIdx.x; i < DP; i += BLOCK) mu[i] = gmu[i];
    __syncthreads();
}

// acc[t][r] (dim 16t+4r+g) = sum_j P[dim][j] * (q_j - mu_j) for the wave's 16 chains.
// fragL = frag2 + lane, muG = mu + g.  Software-pipelined by hand: the A fragments and mu of
// K-step s+1 are fetched from LDS before the NT MFMAs of K-step s issue; the sched_barrier
// pins that shape (left alone, the scheduler hoists every ds_read of the unrolled loop to
// the top and spills hundreds of VGPRs).
template <int NT>
__device__ __forceinline__ void matvec(const v2f64* __restrict__ fragL,
                                       const double* __restrict__ muG,
                                       const double (&q)[4 * NT], v4f64 (&acc)[NT]) {
    constexpr int KS = 4 * NT;
    constexpr int H = NT / 2;
    v2f64 A[H], An[H];
    double m0 = muG[0], mn = 0.0;
#pragma unroll
    for (int t2 = 0; t2 < H; ++t2) A[t2] = fragL[t2 * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int t2 = 0; t2 < H; ++t2) An[t2] = fragL[((s + 1) * H + t2) * 64];
            mn = muG[4 * (s + 1)];
        }
        const double x = q[s] - m0;
#pragma unroll
        for (int t2 = 0; t2 < H; ++t2) {
            acc[2 * t2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].x, x, acc[2 * t2], 0, 0, 0);
            acc[2 * t2 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].y, x, acc[2 * t2 + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t2 = 0; t2 < H; ++t2) A[t2] = An[t2];
        m0 = mn;
    }
}

// sum over the 4 lanes (g = 0..3) that hold one chain; identical bits in all four.
__device__ __forceinline__ double chain_sum(double x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

template <int NT>
__device__ __forceinline__ double dot_x_acc(const double* __restrict__ muG,
                                             const double (&q)[4 * NT], const v4f64 (&acc)[NT]) {
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * t + r;
            sum += (q[s] - muG[4 * s]) * acc[t][r];
        }
    return sum;
}

// State element s of this lane lives at row 4s+g, column (tile base + cc).  Buffer addressing
// (pbbi_buf.h): descriptor = tile base (wave-uniform), voff = 8*(g*ld + cc) in ONE VGPR for
// every row of the array, row offset s*stride4 (stride4 = 8*4*ld) in an SGPR.  The host checks
// DP*ld < 2^29 so that all byte offsets fit 32 bits.  All validity branches are on kernel
// arguments only (uniform): rows 4s..4s+3 are all inside D, partly inside (only when
// D % 4 != 0) or all padding.  FULL (D == DP, no padded rows) compiles every branch away.
template <bool FULL>
__device__ __forceinline__ double load_elem(__amdgpu_buffer_rsrc_t base, uint32_t voff,
                                            uint32_t stride4, uint32_t ld, int s, int g, int D) {
    if constexpr (FULL) return buf_load<double>(base, voff, (uint32_t)s * stride4);
    double val = 0.0;
    if (4 * s + 3 < D) {
        val = buf_load<double>(base, voff, (uint32_t)s * stride4);
    } else if (4 * s < D) {  // partial row group: out-of-range lanes read row 4s and discard
        const bool ok = 4 * s + g < D;
        const double t = buf_load<double>(base, voff - (ok ? 0u : (uint32_t)g * ld),
                                          (uint32_t)s * stride4);
        val = ok ? t : 0.0;
    }
    return val;
}

template <bool FULL>
__device__ __forceinline__ void store_elem(__amdgpu_buffer_rsrc_t base, uint32_t voff,
                                           uint32_t stride4, int s, int g, int D, double val) {
    if constexpr (FULL) {
        buf_store(base, voff, (uint32_t)s * stride4, val);
        return;
    }
    if (4 * s + 3 < D) {
        buf_store(base, voff, (uint32_t)s * stride4, val);
    } else if (4 * s < D) {
        if (4 * s + g < D) buf_store(base, voff, (uint32_t)s * stride4, val);
    }
}

// k_dense_hmc's accessors: no guards at all.  For D < DP the state arrays are addressed through BOUNDED
// descriptors (pbbi_buf.h::buf_make_rows: num_records ends at the array's last element), so a load of a row
// d >= D returns 0 and a store to it is dropped by the hardware's range check -- the row guards of
// load_elem / store_elem above cost the fused kernel > 100 spilled registers (phi copies of the state arrays
// around every branch), which is why padded D used to run unfused.
__device__ __forceinline__ double load_row(__amdgpu_buffer_rsrc_t base, uint32_t voff, uint32_t stride4, int s) {
    return buf_load<double>(base, voff, (uint32_t)s * stride4);
}
__device__ __forceinline__ void store_row(__amdgpu_buffer_rsrc_t base, uint32_t voff, uint32_t stride4, int s,
                                          double val) {
    buf_store(base, voff, (uint32_t)s * stride4, val);
}
// descriptor of a (D, N)-shaped array (leading stride ld) seen from chain n0: unbounded when D == DP
template <bool FULL>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_of(const double* arr, int64_t n0, int D, int64_t ld,
                                                          int64_t N) {
    if constexpr (FULL) return buf_make(arr + n0);
    return buf_make_rows(arr + n0, D, ld, N, n0, 8);
}

template <int NT, int METHOD, bool FULL>
__global__ void __launch_bounds__(BLOCK, 1) k_dense_traj(DensePrm prm) {
    constexpr int DP = 16 * NT;
    constexpr int KS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f64* frag2 = reinterpret_cast<v2f64*>(smem);
    double* mu = reinterpret_cast<double*>(smem + (size_t)DP * DP * sizeof(double));
    stage_lds<NT>(prm.frag, prm.mu, frag2, mu);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int c = lane & 15;
    const v2f64* fragL = frag2 + lane;
    const double* muG = mu + g;
    const int D = prm.D;
    const bool unit = (prm.mass == nullptr);
    const double h = prm.h, h2 = prm.h * prm.h;
    const int64_t n_wg_tiles = (prm.N + CHAINS_PER_WG - 1) / CHAINS_PER_WG;

    for (int64_t wt = blockIdx.x; wt < n_wg_tiles; wt += gridDim.x) {
        const int64_t n0 = (wt * 4 + wave) * CHAINS_PER_WAVE;  // wave-uniform
        if (n0 >= prm.N) continue;
        const int64_t left = prm.N - n0;
        const bool valid = c < left;
        const int cc = c < left ? c : (int)left - 1;  // ragged tail: compute on a clamped chain
        // byte offsets for the buffer accesses (pbbi_buf.h)
        const uint32_t ld_in = 8u * (uint32_t)prm.ldn_in, ld_out = 8u * (uint32_t)prm.ldn_out;
        const uint32_t vin = (uint32_t)g * ld_in + 8u * (uint32_t)cc, s4in = 4u * ld_in;
        const uint32_t vout = (uint32_t)g * ld_out + 8u * (uint32_t)cc, s4out = 4u * ld_out;
        const __amdgpu_buffer_rsrc_t qin = buf_make(prm.q_in + n0);
        const __amdgpu_buffer_rsrc_t pin = buf_make(prm.p_in + n0);
        const __amdgpu_buffer_rsrc_t qout = buf_make(prm.q_out + n0);
        const __amdgpu_buffer_rsrc_t pout = buf_make(prm.p_out + n0);
        const bool have_pout = (prm.p_out != nullptr);
        const double m = unit ? 1.0 : prm.mass[n0 + cc];
        const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);

        double q[KS], v[KS], a[KS];
        v4f64 acc[NT];
        // ---- momentum first (v holds p until the division by mass below); q is fetched
        //      afterwards so that it is not live across the register-hungry RNG code.
        double u = 0.0;
        if (prm.rng) {
            const double pstd = sqrt(m * prm.kT);  // src/ensemble.py:88
#pragma unroll
            for (int k = 0; k < KS / 4; ++k) {  // block k: rows 16k + 4*slot + g
                double z[4];
                rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, prm.iter, chain, (uint32_t)((k << 2) | g),
                             (prm.flags & PBBI_DRAW_F64) != 0, z);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl)
                    v[4 * k + sl] = (FULL || 16 * k + 4 * sl + g < D) ? z[sl] * pstd : 0.0;
            }
            u = rng_uniform(prm.seed, prm.iter, chain);
            if (have_pout && !(prm.flags & PBBI_COMPAT_P_FROM_OLDQ) && valid) {
                // non-compat: a rejected chain reports its drawn momentum; park the draw now
                // (accepted chains overwrite it below) rather than regenerate it later.
#pragma unroll
                for (int s = 0; s < KS; ++s) store_elem<FULL>(pout, vout, s4out, s, g, D, v[s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) v[s] = load_elem<FULL>(pin, vin, s4in, ld_in, s, g, D);
            if (prm.mode == 0) u = prm.u_in[n0 + cc];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) q[s] = load_elem<FULL>(qin, vin, s4in, ld_in, s, g, D);
        double pp_old = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pp_old += v[s] * v[s];
        if (!unit) {
#pragma unroll
            for (int s = 0; s < KS; ++s) v[s] = v[s] / m;  // v = p/m  (:106,:143)
        }

        // ---- first gradient evaluation; doubles as U(q_old)
        matvec<NT>(fragL, muG, q, acc);
        double xg_old = dot_x_acc<NT>(muG, q, acc);
        double xg_new = xg_old;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[4 * t + r] = unit ? -acc[t][r] : -acc[t][r] / m;

        if constexpr (METHOD == PBBI_LEAPFROG) {
            for (int j = 0; j < prm.L; ++j) {
#pragma unroll
                for (int s = 0; s < KS; ++s) q[s] += (v[s] * h + (0.5 * a[s]) * h2);  // :112-115
                matvec<NT>(fragL, muG, q, acc);                                       // :116
                if (j == prm.L - 1) xg_new = dot_x_acc<NT>(muG, q, acc);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = 4 * t + r;
                        const double an = unit ? -acc[t][r] : -acc[t][r] / m;
                        v[s] += (0.5 * (a[s] + an)) * h;  // :117
                        a[s] = an;
                    }
            }
        } else {  // Stormer-Verlet, src/integrator.py:142-163 (a[] becomes qPast)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double q0 = q[s];
                q[s] = (q0 + v[s] * h) + (0.5 * a[s]) * h2;  // :145-150
                a[s] = q0;                                   // qPast
            }
            for (int j = 0; j < prm.L; ++j) {
                matvec<NT>(fragL, muG, q, acc);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = 4 * t + r;
                        const double an = unit ? -acc[t][r] : -acc[t][r] / m;
                        const double cur = q[s];
                        q[s] = (2 * cur - a[s]) + an * h2;  // :155-159
                        a[s] = cur;
                    }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) v[s] = (q[s] - a[s]) / h;  // :162
            if (prm.mode == 0) {  // U(q_new) needs its own mat-vec here
                matvec<NT>(fragL, muG, q, acc);
                xg_new = dot_x_acc<NT>(muG, q, acc);
            }
        }

        if (prm.mode == 1) {  // integrate(): in place q, p; optional Integrator.v
            const __amdgpu_buffer_rsrc_t vout_p = buf_make(prm.v_out + n0);
            if (valid) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    store_elem<FULL>(qout, vout, s4out, s, g, D, q[s]);
                    store_elem<FULL>(pout, vout, s4out, s, g, D, unit ? v[s] : v[s] * m);
                }
                if (prm.v_out) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) store_elem<FULL>(vout_p, vout, s4out, s, g, D, v[s]);
                }
            }
            continue;
        }

        // ---- energies, ratio, decision (src/HMC.py:109-115,166-173)
        double pp_new = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            v[s] = unit ? v[s] : v[s] * m;  // p = v*m  (:119); v now holds p
            pp_new += v[s] * v[s];
        }
        pp_old = chain_sum(pp_old);
        pp_new = chain_sum(pp_new);
        xg_old = chain_sum(xg_old);
        xg_new = chain_sum(xg_new);
        const double oldH = 0.5 * pp_old / m + (0.5 * xg_old + prm.cst);
        const double newH = 0.5 * pp_new / m + (0.5 * xg_new + prm.cst);
        const double ratio = exp((oldH - newH) * pbbi_accept_beta(prm.flags, prm.kT));
        const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
        const bool compat = (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) != 0;
        bool store_p = have_pout;
        if (reject) {  // rare: fetch the old point again instead of keeping 64 more VGPRs live
#pragma unroll
            for (int s = 0; s < KS; ++s) q[s] = load_elem<FULL>(qin, vin, s4in, ld_in, s, g, D);  // :175
            if (compat) {  // :176  p <- oldQ
#pragma unroll
                for (int s = 0; s < KS; ++s) v[s] = q[s];
            } else if (prm.rng) {
                store_p = false;  // the parked draw stays
            } else if (have_pout) {
#pragma unroll
                for (int s = 0; s < KS; ++s) v[s] = load_elem<FULL>(pin, vin, s4in, ld_in, s, g, D);
            }
        }
        if (valid) {
#pragma unroll
            for (int s = 0; s < KS; ++s) store_elem<FULL>(qout, vout, s4out, s, g, D, q[s]);  // :178
            if (store_p) {
#pragma unroll
                for (int s = 0; s < KS; ++s) store_elem<FULL>(pout, vout, s4out, s, g, D, v[s]);  // :179
            }
        }
        if (valid && g == 0) {
            if (prm.ratio_out) prm.ratio_out[n0 + c] = ratio;
            if (prm.reject_out) prm.reject_out[n0 + c] = reject ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_dense_hmc: the production leapfrog path (L >= 1).  Two waves per SIMD.
//
// The f64 MFMA pipe is per SIMD; one wave alone cannot keep it busy while it also runs the
// leapfrog bookkeeping, the momentum draw and its loads/stores (k_dense_traj above: matrix
// pipe 50 % busy).  Two waves per SIMD let the hardware overlap one wave's VALU / memory
// phases with the other wave's MFMAs.  That needs a workgroup of 512 threads (8 waves share
// the one 129 KiB LDS image of P) and a wave that fits in 256 registers -- which the
// reference's velocity-Verlet form (q, v, a + accumulator = 256 registers of state alone)
// does not.  Three measures make it fit, each verified against the ISA / on the GPU:
//   1. Kick-drift-kick form: state is q and the half-step velocity vh only,
//        vh_0     = v_0 + (0.5*a_0)*h          q_{j+1} = q_j + vh_j*h
//        vh_{j+1} = vh_j + a_{j+1}*h           v_L     = vh_{L-1} + (0.5*a_L)*h
//      algebraically identical to src/integrator.py:112-117 (q += v*h + 0.5*a*h**2;
//      v += 0.5*(a+a')*h), different by rounding only (a few ulp per step; the dense path is
//      tolerance-checked in any case because of the MFMA summation order).  Masses enter as
//      v = p*(1/m), a = -(g*(1/m)) (exact for the reference's default unit masses).
//   2. The 128 output rows of each mat-vec are produced in two PASSES of 64 rows, so the live
//      accumulator is 32 registers, not 64; same MFMA count, x_s is simply formed twice.
//   3. One 128-chain tile per workgroup, no persistent loop (a grid-stride variant measured the
//      same throughput and spills twice as much), no run-time branches around the
//      register arrays, buffer (SRSRC) addressing: in a loop hipcc hoists every loop-invariant
//      (~60 fp64 polynomial constants of log/sincospi/exp, all row offsets) and keeps them
//      live for the whole kernel (~140 VGPRs); branches around array updates double the
//      arrays through phi copies; flat addressing costs one 64-bit VGPR address per row.
// The momentum is drawn straight into the velocity registers: one Philox block and two
// single-precision Box-Muller transforms per four rows (pbbi_rng.h), nothing goes through memory.
// ------------------------------------------------------------------------------------------
constexpr int BLOCK2 = 512;
constexpr int CHAINS_PER_WG2 = 128;

// One PASS of the mat-vec: row tiles [PASS*NTP, (PASS+1)*NTP) of G = P X over all K-steps.
// DRIFT (first pass of a step only): q[s] += vh[s]*h just before element s feeds its K-step.
// ONE set of A fragments, each pair reloaded for the next K-step right after the two MFMAs
// that consumed it; the next x is formed in the shadow of the current K-step's MFMAs.
// ZMEAN: mu == 0, x_s is q_s itself (no LDS read, no subtraction).  The drift is one fma.
// KSKIP (D < DP): the K-steps past the last column of P multiply zeros -- the pass ends at ks_act = ceil(D / 4)
// (wave-uniform; the exit points that can occur for this tile size are the only ones compiled in).
template <int NT>
struct KSkip {  // smallest ks_act a kernel of NT row tiles is launched with: D > the next smaller tile size
    static constexpr int MIN = NT == 2 ? 1 : NT == 4 ? 9 : NT == 6 ? 17 : NT == 8 ? 25 : NT == 12 ? 33 : NT == 16 ? 49 : 1;
};
template <int NT, int NTP, int PASS, bool DRIFT, bool ZMEAN, bool KSKIP = false>
__device__ __forceinline__ void matvec_pass(const v2f64* __restrict__ fragL,
                                            const double* __restrict__ muG, double (&q)[4 * NT],
                                            const double (&vh)[4 * NT], v4f64 (&acc)[NTP],
                                            double h, int ks_act = 4 * NT) {
    constexpr int KS = 4 * NT;
    constexpr int H = NT / 2;    // fragment pairs per K-step in the LDS image
    constexpr int HP = NTP / 2;  // pairs this pass consumes
    constexpr int T0 = PASS * HP;
    v2f64 A[HP];
#pragma unroll
    for (int t2 = 0; t2 < HP; ++t2) A[t2] = fragL[(T0 + t2) * 64];
#pragma unroll
    for (int t = 0; t < NTP; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    if constexpr (DRIFT) q[0] = fma(vh[0], h, q[0]);
    double x = ZMEAN ? q[0] : q[0] - muG[0];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if constexpr (KSKIP) {
            if (s >= KSkip<NT>::MIN && s >= ks_act) break;
        }
#pragma unroll
        for (int t2 = 0; t2 < HP; ++t2) {
            acc[2 * t2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].x, x, acc[2 * t2], 0, 0, 0);
            acc[2 * t2 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].y, x, acc[2 * t2 + 1], 0, 0, 0);
            if (s + 1 < KS) A[t2] = fragL[((s + 1) * H + T0 + t2) * 64];
        }
        double xn = 0.0;
        if (s + 1 < KS) {
            if constexpr (DRIFT) q[s + 1] = fma(vh[s + 1], h, q[s + 1]);
            xn = ZMEAN ? q[s + 1] : q[s + 1] - muG[4 * (s + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
        x = xn;
    }
}

// ------------------------------------------------------------------------------------------
// STREAM (kernels_dstream.hip, 128 < D <= 256): P does not fit the LDS (512 KiB at D = 256), the chain state does
// still fit the registers of ONE wave per SIMD (q and vh: 256 of 512).  The wave keeps its 16 chains for the
// whole launch exactly as above and P is STREAMED past it: the A fragments are laid out in the order the two row
// passes of a mat-vec consume them, cut into chunks of KC = 4 K-steps, and travel L2 -> LDS by LDS-DMA
// (buffer_load ... lds: no registers, no wave time) into a ring of RING chunks, RING - 1 of them in flight
// ahead of the MFMAs.  Every mat-vec of the kernel consumes the same NCH chunks in the same order, so the
// stream never stops: the tail of one mat-vec prefetches the head of the next (P is the same).  The ring
// length divides the chunks of a pass, which makes every slot index a compile-time constant.
//   acquire(j), just before the first fragment of chunk j is read:
//     s_waitcnt vmcnt(...)   this wave's share of chunk j has landed (its DMA instructions retire in order)
//     barrier                everybody's has -- and everybody has finished reading chunk j - 1
//     DMA chunk j + RING - 1 into the slot of chunk j - 1
// All four waves execute every acquire (no early exit: a wave past the end of the ensemble recomputes the
// last tile with its stores masked).  Loads and stores of the state that sit between the DMA instructions
// only make the vmcnt wait stricter than necessary (gfx9 retires vector memory operations in order).
#ifndef PBBI_DSTREAM_ABLATE
#define PBBI_DSTREAM_ABLATE 0   // timing experiments of tools/build_variant_dstream.sh; never in a shipped build
#endif
template <int NT>
struct StreamCfg {
    static constexpr int NTP = NT / 2, HP = NTP / 2, KS = 4 * NT;
    static constexpr int KC = 4;                       // K-steps per chunk
    static constexpr int CHUNK = KC * HP * 1024;       // bytes: KC * HP fragment pairs of 64 lanes x 16 B
    static constexpr int CPP = KS / KC;                // chunks per row pass (= NT)
    static constexpr int NCH = 2 * CPP;                // chunks per mat-vec
    static constexpr int RING = NT / 2;                // slots (8 x 16 KiB at DP = 256, 6 x 12 KiB at DP = 192)
    static constexpr int DMA = KC * HP / 4;            // DMA instructions (1 KiB each) per wave and chunk
    static constexpr int LDS_RING = RING * CHUNK;
    static_assert(NT % 4 == 0 && CPP % RING == 0 && (KC * HP) % 4 == 0, "static slots, whole DMA shares");
};

struct PRing {
    __amdgpu_buffer_rsrc_t src;  // the chunk-ordered fragment image (kernels_dstream.hip: stream_build)
    char* lds;                   // ring base + wave * 1024 (wave-uniform)
    uint32_t voff;               // 16 * lane
    uint32_t woff;               // 1024 * wave
};

// one of the DMA instructions of chunk ci (a wave's share of a chunk: m = 0 .. DMA - 1)
template <int NT>
__device__ __forceinline__ void ring_issue_part(const PRing& r, int ci /* chunk of the mat-vec, compile-time */, int m) {
    using C = StreamCfg<NT>;
    typedef __attribute__((address_space(3))) void lds_void;
    const int slot = ci % C::RING;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.src, (lds_void*)(r.lds + slot * C::CHUNK + m * 4096), 16, r.voff,
                                             (uint32_t)(ci * C::CHUNK + m * 4096) + r.woff, 0, 0);
}
template <int NT>
__device__ __forceinline__ void ring_issue(const PRing& r, int ci) {
#pragma unroll
    for (int m = 0; m < StreamCfg<NT>::DMA; ++m) ring_issue_part<NT>(r, ci, m);
}

// INLOOP: the acquire inside a pass sits in the middle of a chunk's last K-step, BEFORE that K-step's DMA
// instruction (if it has one) -- one DMA instruction fewer is younger than the chunk waited for.
template <int NT, bool INLOOP>
__device__ __forceinline__ void ring_acquire(const PRing& r) {
    using C = StreamCfg<NT>;
    constexpr int YOUNGER = INLOOP ? C::DMA * (C::RING - 3) + (C::DMA < C::KC - 1 ? C::DMA : C::KC - 1)
                                   : C::DMA * (C::RING - 2);
    // (not __syncthreads(): its release fence drains vmcnt to 0 -- the whole prefetch -- before the barrier.  The
    //  LDS reads of chunk j - 1 are complete (lgkmcnt 0), the DMA data of chunk j is in LDS once vmcnt says so.)
#if !(PBBI_DSTREAM_ABLATE & 2)  // (bit 1: no wait, no barrier -- timing experiment, wrong results)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(YOUNGER) : "memory");
#endif
}

// The streamed form of matvec_pass: same MFMA order (bit-identical sums), fragments from the ring.  One wave per
// SIMD has nobody to hide its latencies, so each K-step is: half of its MFMAs | the requests of K-step s + 1 (A
// fragments and mu into a second register set, and ONE of the DMA instructions that refill the slot the previous
// chunk left) | the other half | x of K-step s + 1 from what has arrived meanwhile.  Chunk j + RING - 1 is
// requested during the K-steps of chunk j, after acquire(j) has shown that everybody is done with chunk j - 1.
template <int NT, int PASS, bool DRIFT, bool ZMEAN, bool KSKIP = false>
__device__ __forceinline__ void matvec_pass_stream(const PRing& r, const v2f64* __restrict__ ringL,
                                                   const double* __restrict__ muG, double (&q)[4 * NT],
                                                   const double (&vh)[4 * NT], v4f64 (&acc)[NT / 2], double h,
                                                   int ks_act = 4 * NT) {
    using C = StreamCfg<NT>;
    constexpr int KS = C::KS, HP = C::HP, NTP = C::NTP, KC = C::KC;
    constexpr int C0 = PASS * C::CPP;  // first chunk of this pass
    constexpr int H1 = (HP + 1) / 2;   // fragment pairs issued before the next K-step's requests
    // fragment pair t2 of K-step s of this pass: chunk C0 + s / KC, slot (that) % RING
    auto frag_at = [&](int s, int t2) -> v2f64 {
        const int slot = (C0 + s / KC) % C::RING;
        return ringL[((slot * KC + s % KC) * HP + t2) * 64];
    };
    ring_acquire<NT, false>(r);
    v2f64 A[HP], An[HP];
#pragma unroll
    for (int t2 = 0; t2 < HP; ++t2) A[t2] = frag_at(0, t2);
#pragma unroll
    for (int t = 0; t < NTP; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    if constexpr (DRIFT) q[0] = fma(vh[0], h, q[0]);
    double x = ZMEAN ? q[0] : q[0] - muG[0];
    [[maybe_unused]] int s_exit = KS;  // KSKIP: the K-step at which the pass ran out of columns
    // Two gaps per K-step.  (One slot of other work behind every MFMA instead measured 5 % slower: a DMA
    // instruction between two MFMAs costs more of the wave's issue time than one next to VALU work.)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if constexpr (KSKIP) {
            // the columns past D are zeros: no MFMAs, but the ring keeps turning (every chunk is acquired and
            // its successor requested, in the K-steps' own order, by every wave)
            if (s >= KSkip<NT>::MIN && s >= ks_act) {
                s_exit = s;
                break;
            }
        }
#pragma unroll
        for (int t2 = 0; t2 < H1; ++t2) {
            acc[2 * t2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].x, x, acc[2 * t2], 0, 0, 0);
            acc[2 * t2 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].y, x, acc[2 * t2 + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        double mun = 0.0;
        if (s + 1 < KS) {
            if ((s + 1) % KC == 0) ring_acquire<NT, true>(r);
#pragma unroll
            for (int t2 = 0; t2 < HP; ++t2) An[t2] = frag_at(s + 1, t2);
            if constexpr (!ZMEAN) mun = muG[4 * (s + 1)];
            if constexpr (DRIFT) q[s + 1] = fma(vh[s + 1], h, q[s + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t2 = H1; t2 < HP; ++t2) {
            acc[2 * t2] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].x, x, acc[2 * t2], 0, 0, 0);
            acc[2 * t2 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t2].y, x, acc[2 * t2 + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the DMA instruction costs its wave 60+ cycles of issue: cheapest in this gap, which has no LDS reads)
#if !(PBBI_DSTREAM_ABLATE & 1)  // (bit 0: the ring is never refilled -- timing experiment, wrong results)
        if (s % KC < C::DMA) ring_issue_part<NT>(r, (C0 + s / KC + C::RING - 1) % C::NCH, s % KC);
#endif
        if (s + 1 < KS) x = ZMEAN ? q[s + 1] : q[s + 1] - mun;
#pragma unroll
        for (int t2 = 0; t2 < HP; ++t2) A[t2] = An[t2];
    }
    if constexpr (KSKIP) {  // the skipped K-steps' share of the ring protocol (a rolled loop: chunk indices at run time)
        for (int s2 = s_exit; s2 < KS; ++s2) {
            if (s2 + 1 < KS && (s2 + 1) % KC == 0) ring_acquire<NT, true>(r);
            if (s2 % KC < C::DMA) ring_issue_part<NT>(r, (C0 + s2 / KC + C::RING - 1) % C::NCH, s2 % KC);
        }
    }
}

// sum over this pass's rows of x * g   (s = 4*(PASS*NTP + t) + r)
template <int NT, int NTP, int PASS, bool ZMEAN>
__device__ __forceinline__ double dot_pass(const double* __restrict__ muG,
                                           const double (&q)[4 * NT], const v4f64 (&acc)[NTP]) {
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * (PASS * NTP + t) + r;
            sum = fma(ZMEAN ? q[s] : q[s] - muG[4 * s], acc[t][r], sum);
        }
    return sum;
}

// vh[s] += a_s * hk with a_s = -g_s/m: one fma per element, ck = hk/m
template <int NT, int NTP, int PASS>
__device__ __forceinline__ void kick_pass(double (&vh)[4 * NT], const v4f64 (&acc)[NTP], double ck) {
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * (PASS * NTP + t) + r;
            vh[s] = fma(-acc[t][r], ck, vh[s]);
        }
}

// MODE 0: one HMC iteration (src/HMC.py:154-179).  MODE 1: integrate() in place.
// METHOD: Leapfrog, or Stormer-Verlet (src/integrator.py:142-163), which in the same state is
//   d = q_n - q_{n-1} = vh*h:  vh_1 = v_0 + (0.5*a_0)*h;  q_{n+1} = q_n + vh*h;  vh += a_n*h
// i.e. kick-drift-kick WITHOUT the final half kick and with L+1 drifts; its returned velocity
// (q_{L+1} - q_L)/h is vh itself.  U(q_new) then needs one more mat-vec at q_{L+1}.
// DYN (PBBI_PER_CHAIN_STEPS, Leapfrog, MODE 0): chain c takes L_c <= L steps.  The tile runs the
// largest L_c of the wave's 16 chains; a chain that has finished is frozen by per-lane coefficients
// (drift step 0, kick 0) instead of a branch around the register arrays, its last kick is its own
// half kick, and x.g for H_new is taken at the wave's last step, where every frozen chain still
// holds its final position.
// CARRY (MODE 0, inside pbbi_hmc_run; Stormer-Verlet in the fused form only): the last mat-vec of an iteration is g(q_new), and the
// next iteration starts from q_new (accepted) or from the point this one started from (rejected) -- in
// both cases a gradient that already exists.  It is kept in HBM, two slabs per chain and one byte that
// says which is current: an iteration reads g from the current slab instead of forming it (L instead of
// L + 1 mat-vecs), writes g(q_new) to the other one and flips the byte when it accepts.  CARRY = 1 is the
// first iteration of a run (forms g(q_0), stores it as slab 0), CARRY = 2 every later one.  The values
// are the ones the mat-vec would produce again -- same instructions on the same q -- so a run's samples
// do not change by a bit.
// DRAW: which momentum draw is compiled in -- 0 single precision only, 1 PBBI_DRAW_F64 only, 2 both behind a
// run-time (wave-uniform) branch.  The fused launch exists as 0 and 1 (the branch costs it 18 spilled
// registers), everything else as 2.
// STREAM (kernels_dstream.hip): 128 < D <= 256, one wave per SIMD with the 512-register budget, P streamed through
// an LDS ring (StreamCfg above); everything else -- layout, passes, carried gradient, fused iterations -- as is.
template <int NT, bool FULL, int MODE, bool ZMEAN, int METHOD = PBBI_LEAPFROG, bool DYN = false, int CARRY = 0,
          bool FUSE = false, int DRAW = 2, bool STREAM = false>
__global__ void __launch_bounds__(STREAM ? BLOCK : BLOCK2, STREAM ? 1 : 2) k_dense_hmc(DensePrm prm) {
    static_assert(CARRY == 0 || (MODE == 0 && !DYN), "carry: HMC iterations of fixed trajectory length");
    static_assert(!FUSE || CARRY == 2, "a fused launch reads the carried gradient (its first iteration may form it)");
    static_assert(!STREAM || NT % 4 == 0, "streamed P: two row passes");
    constexpr int DP = 16 * NT;
    constexpr int KS = 4 * NT;
    constexpr int NPASS = (NT % 4 == 0) ? 2 : 1;  // row passes per mat-vec (NT = 6, DP = 96: one pass of six row tiles)
    constexpr int NTP = NT / NPASS;         // row tiles per pass (even)
    constexpr int WPB = STREAM ? 4 : 8;     // waves per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    STAMP(0);
    v2f64* frag2 = reinterpret_cast<v2f64*>(smem);
    double* mu = reinterpret_cast<double*>(smem + (STREAM ? (size_t)StreamCfg<STREAM ? NT : 4>::LDS_RING
                                                          : (size_t)DP * DP * sizeof(double)));
    [[maybe_unused]] PRing ring{};
    if constexpr (STREAM) {  // the head of the stream: RING - 1 chunks in flight before anything else happens
        using C = StreamCfg<NT>;
        ring.src = buf_make(prm.frag);
        ring.lds = smem + wave * 1024;
        ring.voff = 16u * (uint32_t)lane;
        ring.woff = 1024u * (uint32_t)wave;
#pragma unroll
        for (int ci = 0; ci < C::RING - 1; ++ci) ring_issue<NT>(ring, ci);
        for (int i = threadIdx.x; i < DP; i += BLOCK) mu[i] = prm.mu[i];
        __syncthreads();
    } else
    {   // stage P: all 16-byte loads of a thread in flight at once (a rolled loop waits for each)
        const v2f64* src = reinterpret_cast<const v2f64*>(prm.frag);
        constexpr int NCH = DP * DP / 2 / BLOCK2;  // 16 at D = 128
        if constexpr (NCH >= 1 && (DP * DP / 2) % BLOCK2 == 0) {
            v2f64 tmp[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) tmp[j] = src[threadIdx.x + j * BLOCK2];
#pragma unroll
            for (int j = 0; j < NCH; ++j) frag2[threadIdx.x + j * BLOCK2] = tmp[j];
        } else {
            for (int i = threadIdx.x; i < DP * DP / 2; i += BLOCK2) frag2[i] = src[i];
        }
        for (int i = threadIdx.x; i < DP; i += BLOCK2) mu[i] = prm.mu[i];
        __syncthreads();
    }
    // Static priority for one of the two waves that share a SIMD (waves w and w+4).
#ifndef PBBI_DENSE_FUSE_PRIO
#define PBBI_DENSE_FUSE_PRIO 1
#endif
    if (!STREAM && (!FUSE || PBBI_DENSE_FUSE_PRIO) && wave < 4) __builtin_amdgcn_s_setprio(1);
    STAMP(1);
    const int g = lane >> 4;
    const int c = lane & 15;
    const v2f64* fragL = frag2 + lane;
    const double* muG = mu + g;
    const int D = prm.D;
    const double h = prm.h;
    [[maybe_unused]] const int ks_act = (D + 3) >> 2;  // K-steps that touch a column of P (D < DP: the rest are zeros)
    // one row pass of a mat-vec: P from its LDS image, or from the ring it streams through
#define MATVEC(PASS_, DRIFT_, Q_, V_, ACC_, H_)                                                        \
    do {                                                                                              \
        if constexpr (STREAM) matvec_pass_stream<NT, PASS_, DRIFT_, ZMEAN, !FULL>(ring, fragL, muG, Q_, V_, ACC_, H_, ks_act); \
        else matvec_pass<NT, NTP, PASS_, DRIFT_, ZMEAN, !FULL>(fragL, muG, Q_, V_, ACC_, H_, ks_act); \
    } while (0)

    int64_t n0 = ((int64_t)blockIdx.x * WPB + wave) * CHAINS_PER_WAVE;  // wave-uniform
    bool ghost = false;  // STREAM: every wave takes part in the ring; one past the end recomputes the last tile, stores masked
    if (n0 >= prm.N) {
        if constexpr (!STREAM) return;
        ghost = true;
        n0 = (prm.N - 1) / CHAINS_PER_WAVE * CHAINS_PER_WAVE;
    }
    const int64_t left = prm.N - n0;
    const bool valid = !ghost && c < left;
    const int cc = c < left ? c : (int)left - 1;  // ragged tail: compute on a clamped chain
    // byte offsets for the buffer accesses (pbbi_buf.h)
    const uint32_t ld_in = 8u * (uint32_t)prm.ldn_in, ld_out = 8u * (uint32_t)prm.ldn_out;
    const uint32_t vin = (uint32_t)g * ld_in + 8u * (uint32_t)cc, s4in = 4u * ld_in;
    const uint32_t vout = (uint32_t)g * ld_out + 8u * (uint32_t)cc, s4out = 4u * ld_out;
    const __amdgpu_buffer_rsrc_t qin = rows_of<FULL>(prm.q_in, n0, D, prm.ldn_in, prm.N);
    const __amdgpu_buffer_rsrc_t qout = rows_of<FULL>(prm.q_out, n0, D, prm.ldn_out, prm.N);
    const __amdgpu_buffer_rsrc_t pout = rows_of<FULL>(prm.p_out, n0, D, prm.ldn_out, prm.N);
    const double m = prm.mass ? prm.mass[n0 + cc] : 1.0;
    const double minv = prm.mass ? 1.0 / m : 1.0;
    const bool rng = (MODE == 0) && prm.rng;
    // carried gradient: slabs [2][D][N] with the leading stride N; this lane's offset into the current one
    [[maybe_unused]] const uint32_t ld_g = 8u * (uint32_t)prm.N, s4g = 4u * ld_g;
    [[maybe_unused]] uint32_t vg_cur = 0, vg_new = 0;
    [[maybe_unused]] uint32_t csel = 0;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t gbuf = buf_make(CARRY ? prm.carry_g + n0 : nullptr);
    if constexpr (CARRY != 0) {
        if constexpr (CARRY == 2) csel = prm.carry_sel[n0 + cc];
    }
    double q[KS];
    if constexpr (FUSE) {  // the chain's position stays in these registers for the whole launch
#pragma unroll
        for (int s = 0; s < KS; ++s) q[s] = load_row(qin, vin, s4in, s);
    }
    // FUSE: prm.fuse_S consecutive iterations of the run in this launch (the host passes ldn_in == ldn_out:
    // every iteration after a run's first reads the previous position slab).  Iteration kf draws with
    // counter iter + kf, re-reads a rejected chain's position from slab (slab0 + kf - 1), writes position
    // slab (slab0 + kf) of fuse_q_base (modulo 2 for a burn-in's two scratch slabs) and momentum / ratio /
    // decision rows kf.  Waves that share a SIMD drift apart, so one's draw and stores run under the other's
    // MFMAs; P is staged once per launch and q is never re-read while the chain keeps accepting.
    const int nfuse = FUSE ? prm.fuse_S : 1;
    // KEEPG (streamed P, fused run, DP = 192): with one wave per SIMD nobody covers the wait for the carried
    // gradient (8 of an iteration's 150 us at DP = 256, tools/stamp_probe_dstream.py).  Where the registers allow,
    // the gradient at the chain's position therefore STAYS IN REGISTERS from one iteration to the next (gk: row
    // pass 0, acc: row pass 1, xg_keep: this lane's part of x.g); only a rejected chain goes back to the slab of
    // the position it started from.  The slabs are written as before -- they are what a later rejection (and
    // the next launch) reads.  At DP = 256 the 128 extra live registers spill ~300 (a reload costs what the
    // gradient load did): measured 1 % slower there, 2-10 % faster at DP = 192.
#ifndef PBBI_DSTREAM_KEEPG
#define PBBI_DSTREAM_KEEPG 1
#endif
#ifndef PBBI_DSTREAM_KEEPG_MAXNT
#define PBBI_DSTREAM_KEEPG_MAXNT 12
#endif
    constexpr bool KEEPG = STREAM && FUSE && PBBI_DSTREAM_KEEPG && NT <= PBBI_DSTREAM_KEEPG_MAXNT;
    v4f64 acc[NTP];
    [[maybe_unused]] v4f64 gk[KEEPG ? NTP : 1];
    [[maybe_unused]] double xg_keep = 0.0;
    if constexpr (FUSE) {
        // The launch starts the run (fuse_first): no carried gradient exists yet.  g(q_0) is formed here, once,
        // and stored as slab 0 -- what a CARRY = 1 launch did -- so that the loop below has ONE shape for every
        // iteration (a branch inside it cost 33 spilled registers): iteration 0 reads back what this wave just
        // wrote (same wave, same addresses: program order).
        if (prm.fuse_first) {
            const uint32_t vg0 = (uint32_t)g * ld_g + 8u * (uint32_t)cc;
            v4f64 (&acc0)[NTP] = acc;
            MATVEC(0, false, q, q, acc0, h);
            if (valid) {
#pragma unroll
                for (int t = 0; t < NTP; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_row(gbuf, vg0, s4g, 4 * t + r, acc0[t][r]);
            }
            if constexpr (KEEPG) {
                xg_keep = dot_pass<NT, NTP, 0, ZMEAN>(muG, q, acc0);
#pragma unroll
                for (int t = 0; t < NTP; ++t) gk[t] = acc0[t];
            }
            if constexpr (NPASS == 2) {
                MATVEC(1, false, q, q, acc0, h);
                if (valid) {
#pragma unroll
                    for (int t = 0; t < NTP; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            store_row(gbuf, vg0, s4g, 4 * (NTP + t) + r, acc0[t][r]);
                }
                if constexpr (KEEPG) xg_keep += dot_pass<NT, NTP, 1, ZMEAN>(muG, q, acc0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (KEEPG) {  // the launch continues a run: once per launch, from the chain's current slab
            const uint32_t vg0 = (uint32_t)g * ld_g + 8u * (uint32_t)cc + (csel ? prm.carry_slab_bytes : 0u);
#pragma unroll
            for (int t = 0; t < NTP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gk[t][r] = load_row(gbuf, vg0, s4g, 4 * t + r);
                    acc[t][r] = load_row(gbuf, vg0, s4g, 4 * (NTP + t) + r);
                }
            xg_keep = dot_pass<NT, NTP, 0, ZMEAN>(muG, q, gk);
            xg_keep += dot_pass<NT, NTP, 1, ZMEAN>(muG, q, acc);
        }
    }
#pragma nounroll
    for (int kf = 0; kf < nfuse; ++kf) {
    STAMP(42);
    const uint64_t iter_k = prm.iter + (uint64_t)kf;
    double* const ratio_k = (FUSE && prm.ratio_out) ? prm.ratio_out + (int64_t)kf * prm.N : prm.ratio_out;
    uint8_t* const reject_k = (FUSE && prm.reject_out) ? prm.reject_out + (int64_t)kf * prm.N : prm.reject_out;
    __amdgpu_buffer_rsrc_t qin_k = qin, qout_k = qout, pout_k = pout;
    if constexpr (FUSE) {
        const int64_t s_out = prm.fuse_wrap2 ? ((prm.fuse_slab0 + kf) & 1) : prm.fuse_slab0 + kf;
        const int64_t s_prev = prm.fuse_wrap2 ? ((prm.fuse_slab0 + kf - 1) & 1) : prm.fuse_slab0 + kf - 1;
        if (kf > 0) qin_k = rows_of<FULL>(prm.fuse_q_base + s_prev * prm.fuse_slab, n0, D, prm.ldn_out, prm.N);
        qout_k = rows_of<FULL>(prm.fuse_q_base + s_out * prm.fuse_slab, n0, D, prm.ldn_out, prm.N);
        pout_k = rows_of<FULL>(prm.p_out + (prm.p_out ? (int64_t)kf * prm.fuse_slab : 0), n0, D, prm.ldn_out, prm.N);
    }
    if constexpr (CARRY != 0) {
        const uint32_t vg0 = (uint32_t)g * ld_g + 8u * (uint32_t)cc;
        vg_cur = vg0 + (csel ? prm.carry_slab_bytes : 0u);
        vg_new = vg0 + (csel ? 0u : prm.carry_slab_bytes);
    }

    // ---- momentum: one Philox block per 4 rows (RNG mode) or the uploaded p_in, then q
    double vh[KS];
    // rows of pass PASS of the carried gradient <-> acc (element s = 4*(PASS*NTP + t) + r, like q[s])
    auto carry_load = [&](auto pass_c) {
        constexpr int PASS = decltype(pass_c)::value;
#pragma unroll
        for (int t = 0; t < NTP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[t][r] = load_row(gbuf, vg_cur, s4g, 4 * (PASS * NTP + t) + r);
    };
    auto carry_store = [&](auto pass_c, uint32_t voff) {
        constexpr int PASS = decltype(pass_c)::value;
        if (valid) {
#pragma unroll
            for (int t = 0; t < NTP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    store_row(gbuf, voff, s4g, 4 * (PASS * NTP + t) + r, acc[t][r]);
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // fused run: the first pass of the carried gradient is requested before the draw (acc is idle until
    // the draw is done, and a thousand vector instructions cover the round trip)
    if constexpr (CARRY == 2 && FUSE && !KEEPG) carry_load(P0{});
    double pp = 0.0;
    if constexpr (KEEPG) {
        // Draw, kinetic energy, v = p/m and the opening half kick in ONE pass over the rows: element s of the
        // kept gradient is consumed the moment element s of the momentum exists, so its registers are free for
        // the next Philox block's temporaries.  Same operations on the same values in the same order as the
        // separate loops below.
        const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);
        const double pstd = sqrt(m * prm.kT);  // src/ensemble.py:88
        const double ck0k = 0.5 * (h * minv);
        const bool park = prm.p_out && !(prm.flags & PBBI_COMPAT_P_FROM_OLDQ) && valid;
        const bool draw64 = DRAW == 2 ? (prm.flags & PBBI_DRAW_F64) != 0 : DRAW == 1;  // wave-uniform
#pragma unroll
        for (int k = 0; k < KS / 4; ++k) {
            double z[4];
            rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, iter_k, chain, (uint32_t)((k << 2) | g), draw64, z);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int s = 4 * k + sl;
                double pv = (FULL || 16 * k + 4 * sl + g < D) ? z[sl] * pstd : 0.0;
                if (park) store_row(pout_k, vout, s4out, s, pv);
                pp += pv * pv;
                pv *= minv;
                vh[s] = fma(-(k < NTP ? gk[k % NTP][sl] : acc[k % NTP][sl]), ck0k, pv);
            }
            __builtin_amdgcn_sched_barrier(0);  // block by block: interleaved blocks keep every block's temporaries live
        }
        STAMP(2);
    } else {
    if (rng) {
        const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);
        const double pstd = sqrt(m * prm.kT);  // src/ensemble.py:88
        const bool draw64 = DRAW == 2 ? (prm.flags & PBBI_DRAW_F64) != 0 : DRAW == 1;  // wave-uniform
#pragma unroll
        for (int k = 0; k < KS / 4; ++k) {  // block k: rows 16k + 4*slot + g, slot = 0..3
            double z[4];
            rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, iter_k, chain, (uint32_t)((k << 2) | g), draw64, z);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                vh[4 * k + sl] = (FULL || 16 * k + 4 * sl + g < D) ? z[sl] * pstd : 0.0;
        }
        if (prm.p_out && !(prm.flags & PBBI_COMPAT_P_FROM_OLDQ) && valid) {
            // non-compat: a rejected chain reports its drawn momentum; park the draw in the slab
            // now (accepted chains overwrite it below) rather than regenerate it later.
#pragma unroll
            for (int s = 0; s < KS; ++s) store_row(pout_k, vout, s4out, s, vh[s]);
        }
    } else {
        const __amdgpu_buffer_rsrc_t pin = rows_of<FULL>(prm.p_in, n0, D, prm.ldn_in, prm.N);
#pragma unroll
        for (int s = 0; s < KS; ++s) vh[s] = load_row(pin, vin, s4in, s);
    }
    STAMP(2);
    if constexpr (!FUSE) {
#pragma unroll
        for (int s = 0; s < KS; ++s) q[s] = load_row(qin, vin, s4in, s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        pp += vh[s] * vh[s];
        vh[s] *= minv;  // v = p/m  (:106), as p*(1/m)
    }
    __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(3);

    // ---- g(q_0) in row passes: U(q_old) and the first half kick
    const double ck = h * minv, ckh = 0.5 * ck;  // kick coefficients h/m and h/(2m)
    int Ln = prm.L;  // this chain's steps
    if constexpr (DYN) {  // drawn or uploaded, in [0, L] (include/pbbi.h); 0: the chain does not move at all
        if (rng) Ln = (prm.flags & PBBI_PER_CHAIN_STEPS)
                          ? rng_steps(prm.seed, iter_k, prm.chain0 + (uint64_t)(n0 + cc), prm.L) : prm.L;
        else if (prm.steps_in) Ln = prm.steps_in[n0 + cc];
        Ln = Ln < 0 ? 0 : (Ln > prm.L ? prm.L : Ln);
    }
    const double ck0 = (DYN && Ln == 0) ? 0.0 : ckh;  // a chain with no step gets no opening half kick either
    double xg = 0.0;
    // FOLD1 (streamed kernels that form g(q_0) themselves, fixed lengths): the opening mat-vec is trip -1 of the step
    // loop below -- drift step 0 (q + vh * 0 = q exactly), half kick, x.g taken for H_old -- rather than a copy of
    // the pass code in front of it: the 512-register kernels pay for every copy of that code in spilled registers.
    constexpr bool FOLD1 = STREAM && CARRY != 2;
    if constexpr (KEEPG) {  // x.g at the chain's position was kept; the half kick went with the draw
        xg = xg_keep;
    } else if constexpr (!FOLD1) {
    if constexpr (CARRY == 2) {
        if constexpr (!FUSE) carry_load(P0{});
    } else {
        MATVEC(0, false, q, vh, acc, h);
    }
    if constexpr (CARRY == 1) carry_store(P0{}, vg_cur);
    xg = dot_pass<NT, NTP, 0, ZMEAN>(muG, q, acc);
    kick_pass<NT, NTP, 0>(vh, acc, ck0);
    if constexpr (NPASS == 2) {
        if constexpr (CARRY == 2) carry_load(P1{});
        else MATVEC(1, false, q, vh, acc, h);
        if constexpr (CARRY == 1) carry_store(P1{}, vg_cur);
        xg += dot_pass<NT, NTP, 1, ZMEAN>(muG, q, acc);
        kick_pass<NT, NTP, 1>(vh, acc, ck0);
    }
    }
    // H(q_old, p_old) now, so that only one double stays live across the trajectory
    double oldH = 0.5 * chain_sum(pp) / m + (0.5 * chain_sum(xg) + prm.cst);   // (FOLD1: set in trip -1 below)
    STAMP(4);
    [[maybe_unused]] const double xg_old_keep = xg;
    if constexpr (!DYN) xg = 0.0;  // (DYN keeps x.g(q_0): a wave none of whose chains steps ends where it started)
    if constexpr (DYN) {
        // Per-chain lengths.  Chain c takes Ln <= L steps (drawn or uploaded) and, with PBBI_UTURN_STOP, stops
        // at the first step j where (q_j - q_0) . v_j < 0.  The 16-chain tile keeps stepping while any of its
        // chains is live; a chain that has finished is frozen by per-lane coefficients (drift step 0, kick 0),
        // never by a branch around the register arrays, and every executed step recomputes g at its (fixed)
        // position, so x.g of the last executed step is x.g at every chain's final position.
        //   * a chain whose last step is known in advance (Ln) gets its closing HALF kick there;
        //   * a U-turn is known only after the step's full kick: the chain then owes half a kick BACK,
        //     -(h/2m)(-g_j), paid at the next executed step (where g_j is formed again) -- if necessary an
        //     extra step that moves nothing.  v_j for the test is vh_{j-1/2} - (h/2m) g_j, formed pass by pass
        //     before the kick; q_0 is re-read from the iteration's input (the start point does not fit the
        //     registers).
        const bool uturn = (prm.flags & PBBI_UTURN_STOP) != 0;
        bool alive = true;
        double pend = 0.0;  // kick coefficient owed (0 or -h/2m)
        int taken = Ln;
        auto uturn_dot = [&](auto pass_c) {  // sum over this pass's rows of (q - q_0) * v_j
            constexpr int PASS = decltype(pass_c)::value;
            double q0r[4 * NTP];
#pragma unroll
            for (int t = 0; t < 4 * NTP; ++t) q0r[t] = load_row(qin_k, vin, s4in, 4 * PASS * NTP + t);
            double sum = 0.0;
#pragma unroll
            for (int t = 0; t < NTP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * (PASS * NTP + t) + r;
                    sum = fma(q[s] - q0r[4 * t + r], fma(-acc[t][r], ckh, vh[s]), sum);
                }
            return sum;
        };
        using U0 = std::integral_constant<int, 0>;
        using U1 = std::integral_constant<int, 1>;
        // (FOLD1: trip -1 is the opening mat-vec -- nobody drifts, every chain takes its opening half kick ck0, x.g is
        //  taken for H_old, no U-turn test, no vote: all four waves make that trip)
        for (int j = FOLD1 ? -1 : 0; j <= prm.L; ++j) {  // at most L steps and one step that only pays a kick back
            const bool first = FOLD1 && j < 0;
            const bool active = !first && alive && j < Ln;
            bool more = first || __builtin_amdgcn_ballot_w64(active || pend != 0.0) != 0;  // wave-uniform
            if (STREAM && !first) {
                // the ring keeps the workgroup's four waves in ONE mat-vec sequence: they step while any of them has a
                // live chain (a wave whose chains are done runs frozen steps).  One flag per wave and step parity,
                // one bare barrier per step.
                int* vote = reinterpret_cast<int*>(mu + DP) + 4 * (j & 1);
                if (lane == 0) vote[wave] = more ? 1 : 0;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                more = (vote[0] | vote[1] | vote[2] | vote[3]) != 0;
            }
            if (!more) break;
            const double cj = first ? ck0 : (active ? (j == Ln - 1 ? ckh : ck) : pend);
            const double hq = active ? h : 0.0;
            if (!first) pend = 0.0;
            double dot = 0.0;
            MATVEC(0, true, q, vh, acc, hq);  // drift + g(q_{j+1})
            if (MODE == 0) xg = dot_pass<NT, NTP, 0, ZMEAN>(muG, q, acc);
            if (uturn && !first) dot = uturn_dot(U0{});
            kick_pass<NT, NTP, 0>(vh, acc, cj);
            if constexpr (NPASS == 2) {
                MATVEC(1, false, q, vh, acc, h);
                if (MODE == 0) xg += dot_pass<NT, NTP, 1, ZMEAN>(muG, q, acc);
                if (uturn && !first) dot += uturn_dot(U1{});
                kick_pass<NT, NTP, 1>(vh, acc, cj);
            }
            if (first) {   // H(q_old, p_old); xg stays x.g(q_0) for a wave none of whose chains steps
                oldH = 0.5 * chain_sum(pp) / m + (0.5 * chain_sum(xg) + prm.cst);
                continue;
            }
            if (uturn) {
                dot = chain_sum(dot);
                if (active && j != Ln - 1 && dot < 0.0) {  // (at its known last step the chain ends anyway)
                    alive = false;
                    taken = j + 1;
                    pend = -ckh;
                }
            }
            if (active && j == Ln - 1) alive = false;
        }
        Ln = taken;
    } else {
    // Stormer-Verlet's position step L + 1 (:155-159 on the last pass of the reference's loop) is, for an HMC
    // iteration, one more trip through the same body: drift + the mat-vec that yields U(q_new) -- which is also
    // g(q_new), carried like Leapfrog's last -- and a kick with coefficient 0.  (A block of its own after the loop
    // had the same instructions again and cost the 512-register kernels 200 spilled registers.)
    constexpr bool SV_EXTRA = METHOD == PBBI_STORMER_VERLET && MODE == 0;
    const int nsteps = SV_EXTRA ? prm.L + 1 : prm.L;
    for (int j = FOLD1 ? -1 : 0; j < nsteps; ++j) {
        const bool first = FOLD1 && j < 0;
        const bool last = (j == nsteps - 1) && (METHOD == PBBI_LEAPFROG || SV_EXTRA);
        // Leapfrog: the last kick is a half kick; Stormer-Verlet: every kick is a full one, the extra trip has none
        const double cj = first ? ck0 : (METHOD == PBBI_LEAPFROG ? (last ? ckh : ck) : (last ? 0.0 : ck));
        const double hj = first ? 0.0 : h;
        STAMP(5 + 2 * j);
        MATVEC(0, true, q, vh, acc, hj);  // drift + g(q_{j+1})
        if ((first || last) && MODE == 0) xg = dot_pass<NT, NTP, 0, ZMEAN>(muG, q, acc);
        if constexpr (CARRY == 1)
            if (first) carry_store(P0{}, vg_cur);
        if constexpr (CARRY != 0)
            if (last) carry_store(P0{}, vg_new);  // g(q_new), for the next iteration if this one accepts
        kick_pass<NT, NTP, 0>(vh, acc, cj);
        if constexpr (KEEPG) {
            if (last) {
#pragma unroll
                for (int t = 0; t < NTP; ++t) gk[t] = acc[t];
            }
        }
        STAMP(6 + 2 * j);
        if constexpr (NPASS == 2) {
            MATVEC(1, false, q, vh, acc, h);
            if ((first || last) && MODE == 0) xg += dot_pass<NT, NTP, 1, ZMEAN>(muG, q, acc);
            if constexpr (CARRY == 1)
                if (first) carry_store(P1{}, vg_cur);
            if constexpr (CARRY != 0)
                if (last) carry_store(P1{}, vg_new);
            kick_pass<NT, NTP, 1>(vh, acc, cj);
        }
        if constexpr (FOLD1) {
            if (first) {
                oldH = 0.5 * chain_sum(pp) / m + (0.5 * chain_sum(xg) + prm.cst);
                xg = 0.0;
            }
        }
    }
    }
    if constexpr (METHOD == PBBI_STORMER_VERLET && MODE == 1) {  // integrate(): position step L + 1, no evaluation
#pragma unroll
        for (int s = 0; s < KS; ++s) q[s] = fma(vh[s], h, q[s]);
    }
    // vh = final velocity, xg = x . g at the final position

    if constexpr (MODE == 1) {  // integrate(): in place q, p; optional Integrator.v
        const __amdgpu_buffer_rsrc_t vout_p = rows_of<FULL>(prm.v_out, n0, D, prm.ldn_out, prm.N);
        if (valid) {
            if (prm.v_out) {
#pragma unroll
                for (int s = 0; s < KS; ++s) store_row(vout_p, vout, s4out, s, vh[s]);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                store_row(qout, vout, s4out, s, q[s]);
                store_row(pout, vout, s4out, s, vh[s] * m);  // p = v*m (:119)
            }
        }
    } else {
        // ---- energies, ratio, decision (src/HMC.py:109-115,166-173)
        pp = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            vh[s] *= m;  // p = v*m  (:119); vh now holds p
            pp += vh[s] * vh[s];
        }
        const double newH = 0.5 * chain_sum(pp) / m + (0.5 * chain_sum(xg) + prm.cst);
        STAMP(40);
        const double ratio = exp((oldH - newH) * pbbi_accept_beta(prm.flags, prm.kT));
        const double u = rng ? rng_uniform(prm.seed, iter_k, prm.chain0 + (uint64_t)(n0 + cc))
                             : prm.u_in[n0 + cc];
        const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
        const bool compat = (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) != 0;
        bool store_p = (prm.p_out != nullptr);
        if (reject) {  // rare: fetch the old point again instead of keeping it in registers
#pragma unroll
            for (int s = 0; s < KS; ++s) q[s] = load_row(qin_k, vin, s4in, s);  // :175
            if constexpr (KEEPG) {  // ... and its gradient, from the slab of the position the chain started from
#pragma unroll
                for (int t = 0; t < NTP; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gk[t][r] = load_row(gbuf, vg_cur, s4g, 4 * t + r);
                        acc[t][r] = load_row(gbuf, vg_cur, s4g, 4 * (NTP + t) + r);
                    }
            }
            if (compat) {  // :176  p <- oldQ
#pragma unroll
                for (int s = 0; s < KS; ++s) vh[s] = q[s];
            } else if (rng) {
                store_p = false;  // the draw parked in the slab stays
            } else if (store_p) {
                const __amdgpu_buffer_rsrc_t pin = rows_of<FULL>(prm.p_in, n0, D, prm.ldn_in, prm.N);
#pragma unroll
                for (int s = 0; s < KS; ++s) vh[s] = load_row(pin, vin, s4in, s);
            }
        }
        if (valid) {
#pragma unroll
            for (int s = 0; s < KS; ++s) store_row(qout_k, vout, s4out, s, q[s]);  // :178
            if (store_p) {
#pragma unroll
                for (int s = 0; s < KS; ++s) store_row(pout_k, vout, s4out, s, vh[s]);  // :179
            }
        }
        if constexpr (CARRY != 0) csel ^= reject ? 0u : 1u;  // accepted: the other slab is current now
        if constexpr (KEEPG) xg_keep = reject ? xg_old_keep : xg;
        if (valid && g == 0) {
            if (ratio_k) ratio_k[n0 + c] = ratio;
            if (reject_k) reject_k[n0 + c] = reject ? 1 : 0;
            if constexpr (DYN) {
                if (prm.steps_out) prm.steps_out[n0 + c] = Ln;
            }
            if constexpr (CARRY != 0 && !FUSE) prm.carry_sel[n0 + c] = (uint8_t)csel;
        }
        STAMP(41);
    }
    }  // kf
    if constexpr (FUSE) {
        if (valid && g == 0) prm.carry_sel[n0 + c] = (uint8_t)csel;
    }
    // the ring's last prefetches (the head of a mat-vec that never comes) must land before the LDS is given back
    if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef MATVEC
}

}  // namespace
