// kernels_lane2.hip -- Rosenbrock potential, 16 < D <= 32, Leapfrog: TWO lanes per chain.
//
// BASELINE config 3 (Rosenbrock d = 32, 262 144 chains) is HBM-bound on paper (103 B and ~870
// flop per step*chain), but with one chain per lane its q, v, a (192 VGPRs at D = 32) leave room
// for only ONE wave per SIMD, so every wave's load, compute and store phases coincide and
// nothing overlaps (measured: 22-28 % of the HBM roofline, time = memory time + VALU time).
// Here lanes l and l^32 share a chain and hold 16 dims each (96 VGPRs of state in the reference
// order, 64 in the kick-drift-kick form): two / four waves per SIMD fit, one wave's HBM phase hides
// under the others' arithmetic, and pbbi_hmc_run keeps the chain in these registers for several
// iterations per launch (k_ros2_hmc below).
//
//   lane l: chain c = l & 31 of the wave's 32 chains, half = l >> 5, dims i = 16*half + j.
//   Loads/stores: each half-wave touches 32 consecutive chains of one row = 256 contiguous bytes.
//   The only coupling across the halves is the nearest-neighbour term at i = 15/16: one
//   v_permlane-style exchange of q_16 and one of the carried gradient term per evaluation.
//
// Arithmetic and summation ORDER are exactly the oracle's (and the chain-per-lane kernel's):
// the sequential energy sums are continued across the halves (half 1 starts from half 0's
// prefix), so q, p are bit-exact against the oracle, as for every chain-per-lane kernel.
#include <cstdlib>

#include "pbbi_buf.h"
#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

constexpr int BLOCK = 64;            // one wave = 32 chains per workgroup: C3's 8192 one-wave workgroups are
constexpr int CHAINS_PER_BLOCK = 32; // exactly two rounds of the chip's 4096 slots (four waves per SIMD)
constexpr int DL = 16;               // dims per lane

struct Ros2Prm {
#ifdef PBBI_STAMPS_ROS2
    unsigned long long* stamps;
#endif
    const double* q_in;
    const double* p_in;
    const double* u_in;
    const double* mass;
    double* q_out;
    double* p_out;
    double* ratio_out;
    uint8_t* reject_out;
    int64_t N, ldn_in, ldn_out;
    double h, a, b, inv_s, cst, kT, c1, c2, c3;
    int L, D, flags, rng;
    uint64_t seed, iter, chain0;
};

__device__ __forceinline__ double xchg(double x) { return __shfl_xor(x, 32, 64); }

#ifdef PBBI_STAMPS_ROS2
// Diagnostic build only (tools/build_stamps_ros2.sh, tools/ros2_timeline.py): per-tile
// s_memrealtime stamps (100 MHz, one clock for the chip) and the hardware wave slot, written to a
// buffer of their own.  Read the TIMELINE of such a build, never its run time.
static unsigned long long* g_ros2_stamps = nullptr;
extern "C" void pbbi_debug_set_ros2_stamp_buffer(void* p) { g_ros2_stamps = (unsigned long long*)p; }
#define RSTAMP(tile, i)                                                                        \
    do {                                                                                       \
        if (prm.stamps && (threadIdx.x & 63) == 0) {                                           \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            unsigned long long t_;                                                             \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            prm.stamps[(size_t)(tile) * 8 + (i)] = t_;                                         \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    } while (0)
#define RSTAMP_HWID(tile)                                                                      \
    do {                                                                                       \
        if (prm.stamps && (threadIdx.x & 63) == 0) {                                           \
            unsigned hw_, xcc_;                                                                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                  \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                \
            prm.stamps[(size_t)(tile) * 8 + 7] = hw_;                                          \
            prm.stamps[(size_t)(tile) * 8 + 6] = xcc_;                                         \
        }                                                                                      \
    } while (0)
#else
#define RSTAMP(tile, i) do { } while (0)
#define RSTAMP_HWID(tile) do { } while (0)
#endif

// FULL: D == 32.  Then every dim exists and only the chain's last dim (half 1, j = 15) lacks a
// right neighbour, so the per-lane predicates below fold to compile-time constants except for
// one select -- without it they are divergent branches around every element.
template <bool FULL>
struct Ros2 {
    double a, b, inv_s, cst, c1, c2, c3;  // c1 = (-4b)/s, c2 = 2/s, c3 = (2b)/s (see kernels_lane.hip)
    int D, half;
    // does global dim 16*half + j have a right neighbour inside D?
    __device__ __forceinline__ bool has_next(int j) const {
        if constexpr (FULL) return j + 1 < DL ? true : half == 0;
        return 16 * half + j + 1 < D;
    }
    __device__ __forceinline__ bool exists(int j) const { return FULL ? true : 16 * half + j < D; }

    // visit(j, -g_j) for this lane's 16 dims.  The oracle's g_i = (0 + c3*t_{i-1}) + first_i is
    // produced NEGATED (what getAccel needs): every term is negated exactly -- -c1, -c3 and the
    // swapped sign inside the fma -- so -g is bit-identical to negating the oracle's g, without
    // one sign-flip instruction per element.
    template <typename F>
    __device__ __forceinline__ void neg_grad_each(const double (&q)[DL], F&& visit) const {
        const double q_ext = xchg(q[0]);  // half 0 receives q_16
        const double nc1 = -c1, nc3 = -c3;
        // -second_15 of half 0 feeds -g_16 (local j = 0 of half 1)
        const double t15 = fma(-q[15], q[15], q_ext);
        const double nsec15 = has_next(15) ? nc3 * t15 : 0.0;
        const double carry_ext = xchg(nsec15);
        double carry = half ? carry_ext : 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            const double t = fma(-q[j], q[j], qn);
            const double nfirst = fma(nc1 * q[j], t, c2 * (a - q[j]));
            const bool hn = has_next(j);
            const double ngj = hn ? carry + nfirst : carry;  // select, not a branch
            carry = hn ? nc3 * t : 0.0;
            visit(j, ngj);
        }
    }

    // PBBI_KDK_FMA: v_j += kk * (-g_j) for this lane's 16 dims, with every product folded into a
    // fused multiply-add and the neighbour term c3*t_j added straight into v_{j+1}: 6 fp64
    // instructions per element including the kick (t, c1 q, the (a - q) term, two fmas into v_j,
    // one into v_{j+1}).  Same -g as neg_grad_each up to rounding.
    __device__ __forceinline__ void kdk_kick(const double (&q)[DL], double (&v)[DL], double kk) const {
        const double q_ext = xchg(q[0]);  // half 0 receives q_16
        const double nc1 = -c1, nc2 = -c2, c2a = c2 * a, kn3 = kk * (-c3);
        const double t15 = fma(-q[15], q[15], q_ext);
        const double t15x = xchg(has_next(15) ? t15 : 0.0);  // half 1 receives half 0's t_15
        v[0] = half ? fma(kn3, t15x, v[0]) : v[0];
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            const double t = fma(-q[j], q[j], qn);
            const double nfirst = fma(nc1 * q[j], t, fma(nc2, q[j], c2a));
            const bool hn = has_next(j);
            const double vj = fma(nfirst, kk, v[j]);
            v[j] = hn ? vj : v[j];
            if (j + 1 < DL) {
                const double vn = fma(kn3, t, v[(j + 1) & (DL - 1)]);
                v[(j + 1) & (DL - 1)] = hn ? vn : v[(j + 1) & (DL - 1)];
            }
        }
    }

    // PBBI_KDK_FMA: U(q) from the two halves' partial sums (one exchange; both halves get the
    // same value) instead of the order-preserving two-pass form below.
    __device__ __forceinline__ double U_pair(const double (&q)[DL]) const {
        const double q_ext = xchg(q[0]);
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            const double t = fma(-q[j], q[j], qn);
            const double r = a - q[j];
            const double n1 = fma(b * t, t, s1), n2 = fma(r, r, s2);
            const bool hn = has_next(j);
            s1 = hn ? n1 : s1;
            s2 = hn ? n2 : s2;
        }
        const double s = s1 + s2;
        return (s + xchg(s)) * inv_s + cst;
    }

    // U(q), valid in the half-1 lanes: the two sequential sums continue half 0's prefixes.  The terms
    // (t_j, b t_j, a - q_j) are formed once; only the two dependent summation chains run twice (from 0
    // for half 0's prefix, then continued from it for half 1).
    __device__ __forceinline__ double U(const double (&q)[DL]) const {
        const double q_ext = xchg(q[0]);
        double t[DL], bt[DL], r[DL];
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            t[j] = fma(-q[j], q[j], qn);
            bt[j] = b * t[j];
            r[j] = a - q[j];
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {  // pass 0: from 0 (half 0's prefix); pass 1: continued
            double r1 = pass ? xchg(s1) : 0.0, r2 = pass ? xchg(s2) : 0.0;
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                const double n1 = fma(bt[j], t[j], r1);
                r1 = has_next(j) ? n1 : r1;
            }
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                const double n2 = fma(r[j], r[j], r2);
                r2 = has_next(j) ? n2 : r2;
            }
            if (pass == 0 || half) { s1 = r1; s2 = r2; }
        }
        return (s1 + s2) * inv_s + cst;
    }
};

// sum_d p_d^2 in oracle order, valid in the half-1 lanes (squares formed once, summed twice)
__device__ __forceinline__ double pp_seq(const double (&p)[DL], int half) {
    double sq[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) sq[j] = p[j] * p[j];
    double s = 0.0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        double r = pass ? xchg(s) : 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) r += sq[j];
        if (pass == 0 || half) s = r;
    }
    return s;
}

// KDK (PBBI_KDK_FMA): kick-drift-kick with fused multiply-adds -- q and the half-step velocity
// only, 7 instead of 14 fp64 instructions per element-step and 64 instead of 96 state registers.
// sum_d p_d^2 from the halves' partial sums (PBBI_KDK_FMA), the same value in both halves
__device__ __forceinline__ double pp_pair(const double (&p)[DL]) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < DL; ++j) s = fma(p[j], p[j], s);
    return s + xchg(s);
}

// One HMC iteration of the wave's 32 chains once their positions are in q[]: momentum (drawn or
// uploaded), H_old, the trajectory, H_new, the decision and the stores (src/HMC.py:154-179).
// n0 = first chain of the tile (wave-uniform), c = chain within the tile, half = which 16 dims.
// DYN (PBBI_PER_CHAIN_STEPS / PBBI_UTURN_STOP, reference-order form only): the chain's own step count
// and the stop at (q_j - q_0) . v_j < 0.  Both lanes of a chain take the same decision (the dot product
// is summed in dimension order across the halves, like the energies: bit-exact with the oracle's
// leapfrog_chain_dyn), so they stay paired for the q_16 / carry exchanges inside the masked loop.
// DRAW: 0 single-precision momentum draw only, 1 PBBI_DRAW_F64 only, 2 both behind a wave-uniform branch (the
// kick-drift-kick kernels exist as 0 and 1: the branch costs them registers they do not have at 4 waves / SIMD)
template <bool UNIT, bool FULL, bool KDK, bool DYN = false, int DRAW = 2>
__device__ __forceinline__ void ros2_tile(const Ros2Prm& prm, const Ros2<FULL>& pot, int64_t n0, int c,
                                          int half, bool valid, int cc, double (&q)[DL], double& U_carry,
                                          bool have_U, double (&a)[DL], bool& a_valid,
                                          const int32_t* steps_in = nullptr, int32_t* steps_out = nullptr) {
    static_assert(!(KDK && DYN), "per-chain lengths run in the reference-order form");
    const int D = prm.D;
    const double m = UNIT ? 1.0 : prm.mass[n0 + cc];
    const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);
    // buffer addressing: per-lane byte offset = column + this half's first row; row j via soffset
    const uint32_t rin = 8u * (uint32_t)prm.ldn_in, rout = 8u * (uint32_t)prm.ldn_out;
    const uint32_t vin = 8u * (uint32_t)cc + (uint32_t)(16 * half) * rin;
    const uint32_t vout = 8u * (uint32_t)cc + (uint32_t)(16 * half) * rout;
    // descriptors bounded to the array (pbbi_buf.h::buf_make_rows): half 1's rows past D read 0 and
    // drop their stores in hardware, no per-lane predication around the memory instructions
    const __amdgpu_buffer_rsrc_t bq = buf_make_rows(prm.q_in + n0, D, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bp = buf_make_rows(prm.p_in + n0, D, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bqo = buf_make_rows(prm.q_out + n0, D, prm.ldn_out, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bpo = buf_make_rows(prm.p_out + n0, D, prm.ldn_out, prm.N, n0, 8);
    auto exists = [&](int j) { return pot.exists(j); };

    double v[DL];  // v holds p, then the velocity, then p again (a: the caller's, see a_valid below)
    [[maybe_unused]] int steps = 0;  // DYN: leapfrog steps this chain took
    const double pstd = prm.rng ? sqrt(m * prm.kT) : 1.0;  // src/ensemble.py:88
    auto draw = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // this half's group of 16 dims: blocks (half<<2)|r
            double z[4];
            rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, prm.iter, chain, (uint32_t)((half << 2) | r),
                         DRAW == 2 ? (prm.flags & PBBI_DRAW_F64) != 0 : DRAW == 1, z);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) v[r + 4 * sl] = exists(r + 4 * sl) ? z[sl] * pstd : 0.0;
        }
    };
    auto load_p = [&]() {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = buf_load<double>(bp, vin, (uint32_t)j * rin);
    };
    if (prm.rng) draw(); else load_p();

    // H(q_old, p_old), src/HMC.py:109-111; correct in the half-1 lanes, then shared with half 0.
    // U(q_old): inside a fused run the position is the one the previous iteration ended on, whose
    // potential energy that iteration already formed (U of its proposal if it accepted, its own U(q_old)
    // if it rejected): the same value, carried in a register instead of evaluated again.
    const double U_old = have_U ? U_carry : (KDK ? pot.U_pair(q) : pot.U(q));
    double oldH;
    if constexpr (KDK) {
        oldH = 0.5 * pp_pair(v) / m + U_old;
    } else {
        oldH = 0.5 * pp_seq(v, half) / m + U_old;
        const double o = xchg(oldH);
        if (!half) oldH = o;
    }

    RSTAMP(n0 / CHAINS_PER_BLOCK, 2);
    // ---- Leapfrog.integrate, src/integrator.py:105-120 (operation order kept)
    // hh2, hh: bit-identical cheaper forms of (0.5*a)*h**2 and (0.5*(a+a'))*h (kernels_lane.hip)
    const double h = prm.h, hh2 = 0.5 * (prm.h * prm.h), hh = 0.5 * prm.h;
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] / m;
    }
    if constexpr (KDK) {
        // vh_{1/2} = v + a0 (h/2);  L x { q += vh h;  vh += a(q) h }, the last kick a half kick
        const double hm = UNIT ? h : h / m, hhm = UNIT ? hh : hh / m;
        if (prm.L > 0) {
            pot.kdk_kick(q, v, hhm);
            for (int s = 0; s < prm.L; ++s) {
#pragma unroll
                for (int j = 0; j < DL; ++j) q[j] = fma(v[j], h, q[j]);
                pot.kdk_kick(q, v, (s + 1 < prm.L) ? hm : hhm);
            }
        }
    } else if constexpr (DYN) {
        int Ln = prm.L;
        if (prm.flags & PBBI_PER_CHAIN_STEPS) {
            if (prm.rng) Ln = prm.L > 0 ? rng_steps(prm.seed, prm.iter, chain, prm.L) : 0;
            else if (steps_in) Ln = steps_in[n0 + cc];
            Ln = Ln < 0 ? 0 : (Ln > prm.L ? prm.L : Ln);
        }
        const bool uturn = (prm.flags & PBBI_UTURN_STOP) != 0;
        double q0[DL];
#pragma unroll
        for (int j = 0; j < DL; ++j) q0[j] = q[j];
        pot.neg_grad_each(q, [&](int j, double ng) { a[j] = UNIT ? ng : ng / m; });
        bool active = Ln > 0;
        while (__builtin_amdgcn_ballot_w64(active) != 0) {  // until the wave's last chain has stopped
            if (active) {
#pragma unroll
                for (int j = 0; j < DL; ++j) q[j] += (v[j] * h + a[j] * hh2);
                pot.neg_grad_each(q, [&](int j, double ng) {
                    const double an = UNIT ? ng : ng / m;
                    v[j] += (a[j] + an) * hh;
                    a[j] = an;
                });
                ++steps;
                // (q - q0) . v summed over the 32 dimensions in order: half 1 continues half 0's sum
                double dot = 0.0;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    double r = pass ? xchg(dot) : 0.0;
#pragma unroll
                    for (int j = 0; j < DL; ++j) r += (q[j] - q0[j]) * v[j];
                    if (pass == 0 || half) dot = r;
                }
                const double other = xchg(dot);
                if (!half) dot = other;  // both lanes hold the chain's sum
                active = steps < Ln && !(uturn && dot < 0.0);
            }
        }
    } else {
        // a = -grad U(q)/m at the starting point.  In a fused run it is what the previous iteration's last
        // step left in a[] -- unless a chain of this wave rejected and went back: then (a_valid false) the
        // wave evaluates it again, which gives the accepted chains the values they already hold.
        if (!a_valid) pot.neg_grad_each(q, [&](int j, double ng) { a[j] = UNIT ? ng : ng / m; });
        for (int s = 0; s < prm.L; ++s) {
#pragma unroll
            for (int j = 0; j < DL; ++j) q[j] += (v[j] * h + a[j] * hh2);
            pot.neg_grad_each(q, [&](int j, double ng) {
                const double an = UNIT ? ng : ng / m;
                v[j] += (a[j] + an) * hh;
                a[j] = an;
            });
        }
    }
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] * m;  // p = v*m
    }

    const double U_new = KDK ? pot.U_pair(q) : pot.U(q);
    double newH;
    if constexpr (KDK) {
        newH = 0.5 * pp_pair(v) / m + U_new;
    } else {
        newH = 0.5 * pp_seq(v, half) / m + U_new;
        const double o = xchg(newH);
        if (!half) newH = o;
    }
    RSTAMP(n0 / CHAINS_PER_BLOCK, 3);
    const double ratio = exp((oldH - newH) * pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const double u = prm.rng ? rng_uniform(prm.seed, prm.iter, chain) : prm.u_in[n0 + cc];
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    U_carry = reject ? U_old : U_new;  // U of the position the chain holds now (both lanes of a chain decide alike)
    if constexpr (!KDK && !DYN) a_valid = (__builtin_amdgcn_ballot_w64(reject) == 0);
    if (reject) {
#pragma unroll
        for (int j = 0; j < DL; ++j)
            q[j] = buf_load<double>(bq, vin, (uint32_t)j * rin);  // :175
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
        for (int j = 0; j < DL; ++j)
            buf_store(bqo, vout, (uint32_t)j * rout, q[j]);
        if (prm.p_out) {
#pragma unroll
            for (int j = 0; j < DL; ++j)
                buf_store(bpo, vout, (uint32_t)j * rout, v[j]);
        }
        if (half == 0) {
            if (prm.ratio_out) prm.ratio_out[n0 + c] = ratio;
            if (prm.reject_out) prm.reject_out[n0 + c] = reject ? 1 : 0;
            if constexpr (DYN) {
                if (steps_out) steps_out[n0 + c] = steps;
            }
        }
    }
    RSTAMP(n0 / CHAINS_PER_BLOCK, 4);
}

// waves per SIMD: the kick-drift-kick form (64 state registers) runs four (three with per-chain
// masses), the reference-order form (96) two, all without scratch (tools/kernel_resources.py)
#ifndef PBBI_ROS2_WAVES_KDK
#define PBBI_ROS2_WAVES_KDK 4
#endif
#ifndef PBBI_ROS2_WAVES_EXACT
#define PBBI_ROS2_WAVES_EXACT 2
#endif

// Where the iterations of a fused run put their results (pbbi_hmc_run, IterArgs::fuse_*): iteration k
// of the launch writes position slab (slab0 + k), modulo 2 for a burn-in's two scratch slabs, and
// momentum slab k; ratio / reject rows k.
struct Ros2Run {
    int S;            // iterations in this launch (1: plain pbbi_hmc_iter semantics)
    int wrap2;        // position slabs alternate between slab 0 and 1 of q_base (burn-in)
    int64_t slab0;    // index of the first iteration's position slab
    int64_t slab;     // elements per slab (D * N)
    double* q_base;   // slab 0 of the position slabs
    const int32_t* steps_in;  // DYN instantiation only: PBBI_PER_CHAIN_STEPS with uploaded draws (nullptr: L)
    int32_t* steps_out;       // DYN instantiation only: the steps each chain took
};

// One 32-chain tile per one-wave workgroup; the chain stays in registers for run.S consecutive HMC
// iterations (config C3's launch: the same 262 144 chains, iteration after iteration).  An iteration
// launched on its own spends a quarter of its time outside the vector pipes' steady state
// (tools/ros2_timeline.py: 4 us until the first trajectory starts because every wave of the first
// round loads and draws at once, 8 us of emptying chip at the end, slot turnover in between); fused,
// a wave pays the load once per launch, only samples leave the chip, and waves that share a SIMD
// drift out of step within a few iterations, so that one wave's draw (32-bit multiplies, xors) and
// stores run beside another's trajectory (fp64).
template <bool UNIT, bool FULL, bool KDK, bool DYN = false, int DRAW = 2>
__global__ void __launch_bounds__(BLOCK, KDK ? (UNIT ? PBBI_ROS2_WAVES_KDK : PBBI_ROS2_WAVES_KDK - 1)
                                              : PBBI_ROS2_WAVES_EXACT)
    k_ros2_hmc(Ros2Prm prm, Ros2Run run) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, c = lane & 31;
    const int64_t n0 = (int64_t)blockIdx.x * CHAINS_PER_BLOCK;  // block-uniform
    if (n0 >= prm.N) return;
    RSTAMP(blockIdx.x, 0);
    RSTAMP_HWID(blockIdx.x);
    const int64_t left = prm.N - n0;
    const bool valid = c < left;
    const int cc = valid ? c : (int)left - 1;
    const Ros2<FULL> pot{prm.a, prm.b, prm.inv_s, prm.cst, prm.c1, prm.c2, prm.c3, prm.D, half};
    const uint32_t rin = 8u * (uint32_t)prm.ldn_in;
    const uint32_t vin = 8u * (uint32_t)cc + (uint32_t)(16 * half) * rin;
    const __amdgpu_buffer_rsrc_t bq = buf_make_rows(prm.q_in + n0, prm.D, prm.ldn_in, prm.N, n0, 8);
    double q[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) q[j] = buf_load<double>(bq, vin, (uint32_t)j * rin);
    double U_carry = 0.0;  // U at the position in q[], from the second iteration of the launch on
    [[maybe_unused]] double a_carry[DL];  // reference-order form: -grad U / m there, while a_valid
    bool a_valid = false;
#pragma nounroll
    for (int k = 0; k < run.S; ++k) {
        Ros2Prm it = prm;  // this iteration's view: where a rejected chain re-reads its position, where the results go
        const int64_t s_out = run.wrap2 ? ((run.slab0 + k) & 1) : run.slab0 + k;
        const int64_t s_prev = run.wrap2 ? ((run.slab0 + k - 1) & 1) : run.slab0 + k - 1;
        if (k > 0) {
            it.q_in = run.q_base + s_prev * run.slab;
            it.ldn_in = prm.ldn_out;
        }
        it.q_out = run.q_base + s_out * run.slab;
        if (prm.p_out) it.p_out = prm.p_out + (int64_t)k * run.slab;
        if (prm.ratio_out) it.ratio_out = prm.ratio_out + (int64_t)k * prm.N;
        if (prm.reject_out) it.reject_out = prm.reject_out + (int64_t)k * prm.N;
        it.iter = prm.iter + (uint64_t)k;
        if constexpr (DYN) {  // (at the register limit: it evaluates U(q_old) and a(q_old) every iteration rather than spill)
            double unused = 0.0, a_dyn[DL];
            bool no = false;
            ros2_tile<UNIT, FULL, KDK, true, DRAW>(it, pot, n0, c, half, valid, cc, q, unused, false, a_dyn, no, run.steps_in,
                                             run.steps_out);
        } else {
            ros2_tile<UNIT, FULL, KDK, false, DRAW>(it, pot, n0, c, half, valid, cc, q, U_carry, k > 0, a_carry, a_valid);
        }
    }
#ifdef PBBI_STAMPS_ROS2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSTAMP(blockIdx.x, 5);
#endif
}

}  // namespace

// true if this path takes the call (Rosenbrock, Leapfrog, fp64, 16 < D <= 32)
bool lane2_applies(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    return pot->kind == KIND_ROSENBROCK && pot->dtype == PBBI_F64 && a.method == PBBI_LEAPFROG &&
           pot->D > 16 && pot->D <= 32;
}

int lane2_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (a.N == 0) return PBBI_OK;
    Ros2Prm prm{
#ifdef PBBI_STAMPS_ROS2
                g_ros2_stamps,
#endif
                (const double*)a.q_in, (const double*)a.p_in, (const double*)a.u_in,
                (const double*)a.mass, (double*)a.q_out, (double*)a.p_out, (double*)a.ratio_out,
                a.reject_out, a.N, a.ldn_in, a.ldn_out, a.h, pot->a, pot->b, 1.0 / pot->s, pot->cst,
                a.kT, (-4.0 * pot->b) * (1.0 / pot->s), 2.0 * (1.0 / pot->s),
                (2.0 * pot->b) * (1.0 / pot->s), a.L, pot->D, a.flags, a.rng, a.seed, a.iter,
                a.chain0};
    // a single iteration is a run of one whose only "slab" is q_out
    Ros2Run run{1, 0, 0, (int64_t)pot->D * a.N, (double*)a.q_out, a.steps_in, a.steps_out};
    if (a.fuse_S > 1)
        run = Ros2Run{a.fuse_S, a.fuse_wrap2, a.fuse_slab0, (int64_t)pot->D * a.N, (double*)a.fuse_q_base, nullptr, nullptr};
    const dim3 grid((unsigned)((a.N + CHAINS_PER_BLOCK - 1) / CHAINS_PER_BLOCK)), block(BLOCK);
    const bool full = (pot->D == 32);
    const bool dyn = pbbi_dyn(a);                                  // per-chain lengths: reference-order form
    const bool kdk = !dyn && (a.flags & PBBI_KDK_FMA) != 0;
#define ROS2_LAUNCH(U_, F_)                                                                          \
    {                                                                                                \
        if (dyn)                                                                                     \
            hipLaunchKernelGGL((k_ros2_hmc<U_, F_, false, true>), grid, block, 0, a.stream, prm, run); \
        else if (kdk && (a.flags & PBBI_DRAW_F64))                                                   \
            hipLaunchKernelGGL((k_ros2_hmc<U_, F_, true, false, 1>), grid, block, 0, a.stream, prm, run); \
        else if (kdk)                                                                                \
            hipLaunchKernelGGL((k_ros2_hmc<U_, F_, true, false, 0>), grid, block, 0, a.stream, prm, run); \
        else                                                                                         \
            hipLaunchKernelGGL((k_ros2_hmc<U_, F_, false>), grid, block, 0, a.stream, prm, run);     \
    }
    if (a.mass) {
        if (full) ROS2_LAUNCH(false, true) else ROS2_LAUNCH(false, false)
    } else {
        if (full) ROS2_LAUNCH(true, true) else ROS2_LAUNCH(true, false)
    }
#undef ROS2_LAUNCH
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}
