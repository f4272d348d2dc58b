// kernels_lane.hip -- chain-per-lane kernels for potentials whose gradient is elementwise
// or nearest-neighbour (harmonic, diagonal Gaussian, Rosenbrock), gfx950.
//
// One ensemble member ("chain") per lane; its q, v, a live in VGPRs for all L steps, so an
// HMC iteration touches HBM once on the way in and once on the way out.  The (D, N)
// chain-fastest layout makes every load/store a 512-byte contiguous segment per wave.
// Operation order follows the reference exactly (no FMA contraction: built with
// -ffp-contract=off), so in parity mode these kernels are bit-identical to the oracle
// for q and p (only exp() of the ratio may differ in the last ulp):
//   Leapfrog.integrate        src/integrator.py:105-120
//   StormerVerlet.integrate   src/integrator.py:142-163
//   H = 0.5*dot(p,p)/m + U    src/HMC.py:100-102,109-115
//   accept/reject + stores    src/HMC.py:164-179
#include <cstdlib>
#include <type_traits>

#include "pbbi_buf.h"
#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

constexpr int BLOCK = 256;

// ------------------------------------------------------------------ potentials
template <typename T, int DMAX, bool FULL>
struct SeparablePot {  // harmonic (src/potential.py:27) and diagonal Gaussian
    const T* __restrict__ mean;
    const T* __restrict__ prec;
    T cst;
    int D;
    int harmonic;
    // mean / prec are stored zero-padded to a multiple of 64 entries (pbbi_api.hip::upload_padded):
    // a padded dim (d >= D) has precision 0, mean 0, and its q, v, a are zero and stay zero, so its
    // terms add exact zeros and no per-element guard is needed.
    __device__ __forceinline__ T pr(int d) const { return prec[d]; }
    __device__ __forceinline__ T mu(int d) const { return mean[d]; }
    __device__ __forceinline__ T U(const T (&q)[DMAX]) const {
        T acc = T(0);
        if (harmonic) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) acc += pr(d) * (q[d] * q[d]);
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const T x = q[d] - mu(d);
                acc += (pr(d) * x) * x;
            }
        }
        return T(0.5) * acc + cst;
    }
    __device__ __forceinline__ void grad(const T (&q)[DMAX], T (&g)[DMAX]) const {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) g[d] = pr(d) * (q[d] - mu(d));
    }
    // visit(d, g_d) for d = 0..DMAX-1 in order, one element at a time (no g[] array live)
    template <typename F>
    __device__ __forceinline__ void grad_each(const T (&q)[DMAX], F&& visit) const {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) visit(d, pr(d) * (q[d] - mu(d)));
    }
};

template <typename T, int DMAX, bool FULL>
struct RosenbrockPot {  // U = (sum b t^2 + sum (a-q_i)^2) * (1/s),  t_i = fma(-q_i, q_i, q_{i+1})
    // The build's own potential (not in the reference): constants pre-combined and fused
    // multiply-adds are part of its DEFINITION (oracle/pbbi_oracle.c::pot_grad restates it with
    // the same fma calls), because this kernel is bound by its fp64 instruction count.
    T a, b, inv_s, cst, c1, c2, c3;  // c1 = (-4b)/s, c2 = 2/s, c3 = (2b)/s
    int D;
    __device__ __forceinline__ T U(const T (&q)[DMAX]) const {
        T s1 = T(0), s2 = T(0);
#pragma unroll
        for (int i = 0; i + 1 < DMAX; ++i)
            if (FULL || i + 1 < D) {
                const T t = fma(-q[i], q[i], q[i + 1]);
                s1 = fma(b * t, t, s1);
            }
#pragma unroll
        for (int i = 0; i + 1 < DMAX; ++i)
            if (FULL || i + 1 < D) {
                const T r = a - q[i];
                s2 = fma(r, r, s2);
            }
        return (s1 + s2) * inv_s + cst;
    }
    __device__ __forceinline__ void grad(const T (&q)[DMAX], T (&g)[DMAX]) const {
        grad_each(q, [&](int d, T gd) { g[d] = gd; });
    }
    // g_i = (0 + c3*t_{i-1}) + fma(c1*q_i, t_i, -(c2*(a - q_i))), element by element
    template <typename F>
    __device__ __forceinline__ void grad_each(const T (&q)[DMAX], F&& visit) const {
        T carry = T(0);
#pragma unroll
        for (int i = 0; i < DMAX; ++i) {
            T gi = carry;
            carry = T(0);
            if (i + 1 < DMAX && (FULL || i + 1 < D)) {
                const T t = fma(-q[i], q[i], q[i + 1]);
                gi += fma(c1 * q[i], t, -(c2 * (a - q[i])));
                carry = c3 * t;
            }
            visit(i, (FULL || i < D) ? gi : T(0));
        }
    }
};

// Rows d >= D of a zero-padded chain: the state arrays are addressed through descriptors bounded
// to the array (pbbi_buf.h::buf_make_rows), so the hardware returns 0 for those loads and drops
// those stores -- no branch around a memory instruction (a scalar branch per row makes hipcc put an
// s_waitcnt between consecutive memory instructions).
template <typename T, bool FULL>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const T* arr, int D, int64_t ld, int64_t N,
                                                            int64_t n0) {
    if constexpr (FULL) return buf_make(arr + n0);
    else return buf_make_rows(arr + n0, D, ld, N, n0, (int)sizeof(T));
}
template <typename T, bool FULL>
__device__ __forceinline__ T load_row(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t row, int d, int) {
    return buf_load<T>(r, voff, (uint32_t)d * row);
}
template <typename T, bool FULL>
__device__ __forceinline__ void store_row(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t row, int d,
                                          int, T x) {
    buf_store(r, voff, (uint32_t)d * row, x);
}

// ---------------------------------------------------------------- integrators
template <typename T, typename Pot, int DMAX, bool UNIT>
__device__ __forceinline__ void accel(const Pot& pot, const T (&q)[DMAX], T m, T (&a)[DMAX]) {
    T g[DMAX];
    pot.grad(q, g);
    if constexpr (UNIT) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) a[d] = -g[d];
    } else {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) a[d] = -g[d] / m;  // src/integrator.py:73
    }
}

// UNIT: every mass is exactly 1 (mass == NULL): v = p, a = -g, p = v, no fp64 divisions.
template <typename T, typename Pot, int DMAX, int METHOD, bool UNIT>
__device__ __forceinline__ void integrate_chain(const Pot& pot, T (&q)[DMAX], T (&p)[DMAX],
                                                T (&v)[DMAX], T m, T h, int L) {
    const T h2 = h * h;
    const T half = T(0.5);
    // (0.5*a)*h**2 == a*(0.5*h**2) and (0.5*(a+a'))*h == (a+a')*(0.5*h) bit for bit (scaling by
    // 0.5 is exact), so the Leapfrog lines below cost 4 + 3 instead of 5 + 4 fp64 instructions
    // and still reproduce src/integrator.py:112-117 exactly.
    const T hh2 = half * h2, hh = half * h;
    T a[DMAX];
    if constexpr (UNIT) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = p[d];
    } else {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = p[d] / m;
    }
    if constexpr (METHOD == PBBI_LEAPFROG) {
        // a = getAccel(q) (:108); the gradient is consumed element by element so that only
        // q, v, a are live (3*DMAX registers instead of 5*DMAX with separate g[] / a'[] arrays)
        pot.grad_each(q, [&](int d, T g) { a[d] = UNIT ? -g : -g / m; });
        for (int j = 0; j < L; ++j) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) q[d] += (v[d] * h + a[d] * hh2);  // :112-115
            pot.grad_each(q, [&](int d, T g) {                                       // :116-118
                const T an = UNIT ? -g : -g / m;
                v[d] += (a[d] + an) * hh;
                a[d] = an;
            });
        }
    } else {
        T qpast[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) qpast[d] = q[d];
        accel<T, Pot, DMAX, UNIT>(pot, q, m, a);
#pragma unroll
        for (int d = 0; d < DMAX; ++d) q[d] = (q[d] + v[d] * h) + (half * a[d]) * h2;
        for (int j = 0; j < L; ++j) {
            accel<T, Pot, DMAX, UNIT>(pot, q, m, a);
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const T cur = q[d];
                q[d] = (T(2) * cur - qpast[d]) + a[d] * h2;
                qpast[d] = cur;
            }
        }
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = (q[d] - qpast[d]) / h;
    }
    if constexpr (UNIT) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) p[d] = v[d];
    } else {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) p[d] = v[d] * m;
    }
}

template <typename T, typename Pot, int DMAX>
__device__ __forceinline__ T hamiltonian(const Pot& pot, const T (&q)[DMAX], const T (&p)[DMAX],
                                         T m) {
    T pp = T(0);
#pragma unroll
    for (int d = 0; d < DMAX; ++d) pp += p[d] * p[d];
    return T(0.5) * pp / m + pot.U(q);
}

// 0.5 p.p / m: hamiltonian's first term (the two are added in the same order there and in k_lane_hmc)
template <typename T, int DMAX>
__device__ __forceinline__ T kinetic(const T (&p)[DMAX], T m) {
    T pp = T(0);
#pragma unroll
    for (int d = 0; d < DMAX; ++d) pp += p[d] * p[d];
    return T(0.5) * pp / m;
}

// momentum draw of one chain: one Philox block per four dims (RNG contract, include/pbbi.h)
template <typename T, int DMAX>
__device__ __forceinline__ void draw_momentum(T (&p)[DMAX], int D, uint64_t seed, uint64_t iter,
                                              uint64_t chain, double pstd, bool f64) {
#pragma unroll
    for (int d = 0; d < DMAX; ++d) p[d] = T(0);
#pragma unroll
    for (int G = 0; G < (DMAX + 15) / 16; ++G)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * G + r < DMAX) {
                double z[4];
                rng_normal4d(seed, PBBI_STREAM_MOMENTUM, iter, chain, (uint32_t)((G << 2) | r), f64, z);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const int d = 16 * G + r + 4 * sl;
                    if (d < DMAX) p[d] = (d < D) ? (T)(z[sl] * pstd) : T(0);  // D uniform
                }
            }
        }
}

// -------------------------------------------------------------------- kernels
template <typename T>
struct HmcPrm {
    const T* q_in;
    const T* p_in;
    const T* u_in;
    const T* mass;
    T* q_out;
    T* p_out;
    T* ratio_out;
    uint8_t* reject_out;
    int64_t N, ldn_in, ldn_out;
    T h;
    int L, D, flags, rng;
    uint64_t seed, iter, chain0;
    double kT;
    // fused run (IterArgs::fuse_*; k_lane_hmc, D <= 16): iteration k of the launch writes position slab
    // (fuse_slab0 + k) of fuse_q_base (modulo 2 when fuse_wrap2), momentum slab k, ratio / reject rows k
    int fuse_S, fuse_wrap2;
    int64_t fuse_slab0, fuse_slab;
    T* fuse_q_base;
};

// fuse_S > 1 (small chains, D <= 16: the launch, not the arithmetic, is what an iteration costs): the lane
// keeps its chain for fuse_S consecutive iterations of the run, and the potential energy of the position an
// iteration starts from is the one the previous iteration evaluated (carried, the same value).
template <typename T, typename Pot, int DMAX, int METHOD, bool FULL, bool UNIT>
__global__ void __launch_bounds__(BLOCK) k_lane_hmc(HmcPrm<T> prm, Pot pot) {
    const int64_t n0 = (int64_t)blockIdx.x * BLOCK;  // block-uniform base chain
    const int64_t n = n0 + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint64_t chain = prm.chain0 + (uint64_t)n;
    // buffer addressing (pbbi_buf.h): one per-lane byte offset for every row of every array
    const uint32_t voff = (uint32_t)threadIdx.x * (uint32_t)sizeof(T);
    const uint32_t rin = (uint32_t)prm.ldn_in * (uint32_t)sizeof(T);
    const uint32_t rout = (uint32_t)prm.ldn_out * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t bq = rows_rsrc<T, FULL>(prm.q_in, D, prm.ldn_in, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bp = rows_rsrc<T, FULL>(prm.p_in, D, prm.ldn_in, prm.N, n0);
    const double pstd = prm.rng ? sqrt((double)m * prm.kT) : 1.0;  // src/ensemble.py:88

    T q[DMAX], p[DMAX], v[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) q[d] = load_row<T, FULL>(bq, voff, rin, d, D);
    T U_cur = T(0);
    const int nfuse = (DMAX <= 16 && prm.fuse_S > 1) ? prm.fuse_S : 1;
#pragma nounroll
    for (int kf = 0; kf < nfuse; ++kf) {
        const uint64_t iter_k = prm.iter + (uint64_t)kf;
        // this iteration's view: where a rejected chain re-reads its position, where the results go
        const int64_t s_out = prm.fuse_wrap2 ? ((prm.fuse_slab0 + kf) & 1) : prm.fuse_slab0 + kf;
        const int64_t s_prev = prm.fuse_wrap2 ? ((prm.fuse_slab0 + kf - 1) & 1) : prm.fuse_slab0 + kf - 1;
        const T* q_in_k = kf > 0 ? prm.fuse_q_base + s_prev * prm.fuse_slab : prm.q_in;
        const int64_t ld_in_k = kf > 0 ? prm.ldn_out : prm.ldn_in;
        T* q_out_k = nfuse > 1 ? prm.fuse_q_base + s_out * prm.fuse_slab : prm.q_out;
        T* p_out_k = (prm.p_out && nfuse > 1) ? prm.p_out + (int64_t)kf * prm.fuse_slab : prm.p_out;
        const uint32_t rin_k = (uint32_t)ld_in_k * (uint32_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t bq_k = rows_rsrc<T, FULL>(q_in_k, D, ld_in_k, prm.N, n0);
        const __amdgpu_buffer_rsrc_t bqo = rows_rsrc<T, FULL>(q_out_k, D, prm.ldn_out, prm.N, n0);
        const __amdgpu_buffer_rsrc_t bpo = rows_rsrc<T, FULL>(p_out_k, D, prm.ldn_out, prm.N, n0);
        T u;
        if (prm.rng) {
            draw_momentum<T, DMAX>(p, D, prm.seed, iter_k, chain, pstd, (prm.flags & PBBI_DRAW_F64) != 0);
            u = (T)rng_uniform(prm.seed, iter_k, chain);
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; ++d)
                p[d] = load_row<T, FULL>(bp, voff, rin, d, D);
            u = prm.u_in[n];
        }
        const T U_old = kf > 0 ? U_cur : pot.U(q);
        const T oldH = kinetic<T, DMAX>(p, m) + U_old;
        integrate_chain<T, Pot, DMAX, METHOD, UNIT>(pot, q, p, v, m, prm.h, prm.L);
        const T U_new = pot.U(q);
        const T newH = kinetic<T, DMAX>(p, m) + U_new;  // p -> -p leaves dot(p,p) unchanged
        const T ratio = exp((oldH - newH) * (T)pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
        // mask = u > min(1, ratio); NaN ratio compares False => accepted (src/HMC.py:168-173)
        const bool reject = (ratio == ratio) && (u > (ratio < T(1) ? ratio : T(1)));
        U_cur = reject ? U_old : U_new;
        if (reject) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d)
                q[d] = load_row<T, FULL>(bq_k, voff, rin_k, d, D);  // :175
            if (prm.p_out) {
                if (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) {  // :176  p <- oldQ
#pragma unroll
                    for (int d = 0; d < DMAX; ++d) p[d] = q[d];
                } else if (prm.rng) {
                    draw_momentum<T, DMAX>(p, D, prm.seed, iter_k, chain, pstd, (prm.flags & PBBI_DRAW_F64) != 0);
                } else {
#pragma unroll
                    for (int d = 0; d < DMAX; ++d)
                        p[d] = load_row<T, FULL>(bp, voff, rin, d, D);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            store_row<T, FULL>(bqo, voff, rout, d, D, q[d]);
        if (prm.p_out) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d)
                store_row<T, FULL>(bpo, voff, rout, d, D, p[d]);
        }
        if (prm.ratio_out) prm.ratio_out[(int64_t)kf * prm.N + n] = ratio;
        if (prm.reject_out) prm.reject_out[(int64_t)kf * prm.N + n] = reject ? 1 : 0;
    }
}

// ---- per-chain trajectory lengths (PBBI_PER_CHAIN_STEPS / PBBI_UTURN_STOP, include/pbbi.h) -------
// Leapfrog.integrate (src/integrator.py:105-120) with the step count of every chain its own: lane n
// takes steps while j < L_n and, under PBBI_UTURN_STOP, until (q_j - q_0) . v_j < 0 (p = v*m has v's
// sign).  The velocity-Verlet form has the synchronised velocity after every step, so the test costs
// one dot product; q_0 rides in p[] (dead between v = p/m and p = v*m).  Finished lanes are masked
// out by the branch; the loop ends for the wave when no lane is active.  Same operation order as
// integrate_chain -> bit-exact with the oracle's leapfrog_chain_dyn, step counts included.
template <typename T, typename Pot, int DMAX, bool UNIT>
__device__ __forceinline__ int integrate_chain_dyn(const Pot& pot, T (&q)[DMAX], T (&p)[DMAX], T (&v)[DMAX],
                                                   T m, T h, int Ln, bool uturn) {
    const T hh2 = T(0.5) * (h * h), hh = T(0.5) * h;
    T a[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        v[d] = UNIT ? p[d] : p[d] / m;
        p[d] = q[d];  // q_0
    }
    pot.grad_each(q, [&](int d, T g) { a[d] = UNIT ? -g : -g / m; });
    int steps = 0;
    bool active = Ln > 0;
    while (__builtin_amdgcn_ballot_w64(active) != 0) {  // wave-uniform: until the wave's last chain stops
        if (active) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) q[d] += (v[d] * h + a[d] * hh2);
            pot.grad_each(q, [&](int d, T g) {
                const T an = UNIT ? -g : -g / m;
                v[d] += (a[d] + an) * hh;
                a[d] = an;
            });
            ++steps;
            T dot = T(0);
#pragma unroll
            for (int d = 0; d < DMAX; ++d) dot += (q[d] - p[d]) * v[d];
            active = steps < Ln && !(uturn && dot < T(0));
        }
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d) p[d] = UNIT ? v[d] : v[d] * m;
    return steps;
}

template <typename T>
struct DynPrm {
    HmcPrm<T> b;
    const int32_t* steps_in;
    int32_t* steps_out;
};

template <typename T, typename Pot, int DMAX, bool FULL, bool UNIT>
__global__ void __launch_bounds__(BLOCK) k_lane_dyn_hmc(DynPrm<T> dp, Pot pot) {
    const HmcPrm<T>& prm = dp.b;
    const int64_t n0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t n = n0 + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint64_t chain = prm.chain0 + (uint64_t)n;
    const uint32_t voff = (uint32_t)threadIdx.x * (uint32_t)sizeof(T);
    const uint32_t rin = (uint32_t)prm.ldn_in * (uint32_t)sizeof(T);
    const uint32_t rout = (uint32_t)prm.ldn_out * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t bq = rows_rsrc<T, FULL>(prm.q_in, D, prm.ldn_in, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bp = rows_rsrc<T, FULL>(prm.p_in, D, prm.ldn_in, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bqo = rows_rsrc<T, FULL>(prm.q_out, D, prm.ldn_out, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bpo = rows_rsrc<T, FULL>(prm.p_out, D, prm.ldn_out, prm.N, n0);
    const double pstd = prm.rng ? sqrt((double)m * prm.kT) : 1.0;  // src/ensemble.py:88

    T q[DMAX], p[DMAX], v[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) q[d] = load_row<T, FULL>(bq, voff, rin, d, D);
    T u;
    int Ln = prm.L;
    if (prm.rng) {
        draw_momentum<T, DMAX>(p, D, prm.seed, prm.iter, chain, pstd, (prm.flags & PBBI_DRAW_F64) != 0);
        u = (T)rng_uniform(prm.seed, prm.iter, chain);
        if (prm.flags & PBBI_PER_CHAIN_STEPS) Ln = prm.L > 0 ? rng_steps(prm.seed, prm.iter, chain, prm.L) : 0;
    } else {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) p[d] = load_row<T, FULL>(bp, voff, rin, d, D);
        u = prm.u_in[n];
        if ((prm.flags & PBBI_PER_CHAIN_STEPS) && dp.steps_in) {
            Ln = dp.steps_in[n];
            Ln = Ln < 0 ? 0 : (Ln > prm.L ? prm.L : Ln);
        }
    }
    const T oldH = hamiltonian<T, Pot, DMAX>(pot, q, p, m);
    const int steps = integrate_chain_dyn<T, Pot, DMAX, UNIT>(pot, q, p, v, m, prm.h, Ln,
                                                              (prm.flags & PBBI_UTURN_STOP) != 0);
    const T newH = hamiltonian<T, Pot, DMAX>(pot, q, p, m);
    const T ratio = exp((oldH - newH) * (T)pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const bool reject = (ratio == ratio) && (u > (ratio < T(1) ? ratio : T(1)));
    if (reject) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) q[d] = load_row<T, FULL>(bq, voff, rin, d, D);  // :175
        if (prm.p_out) {
            if (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) {  // :176  p <- oldQ
#pragma unroll
                for (int d = 0; d < DMAX; ++d) p[d] = q[d];
            } else if (prm.rng) {
                draw_momentum<T, DMAX>(p, D, prm.seed, prm.iter, chain, pstd, (prm.flags & PBBI_DRAW_F64) != 0);
            } else {
#pragma unroll
                for (int d = 0; d < DMAX; ++d) p[d] = load_row<T, FULL>(bp, voff, rin, d, D);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d) store_row<T, FULL>(bqo, voff, rout, d, D, q[d]);
    if (prm.p_out) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) store_row<T, FULL>(bpo, voff, rout, d, D, p[d]);
    }
    if (prm.ratio_out) prm.ratio_out[n] = ratio;
    if (prm.reject_out) prm.reject_out[n] = reject ? 1 : 0;
    if (dp.steps_out) dp.steps_out[n] = steps;
}

// One GIST (self-tuning no-U-turn) iteration of a chain, fused: forward U-turn search from (q, p), the length
// draw, the proposal (L steps from the re-read q and the re-drawn p), the backward search from (q', -p'), the
// accept test with tau_f / tau_b -- include/pbbi.h pbbi_hmc_run_gist, oracle_hmc_iter_gist; the same arithmetic as
// the composed form (three k_lane_dyn_hmc launches + the small kernels of pbbi_api.hip), so the same bits.  The
// proposal is parked in q_out / p_out before the backward search (q_out must not alias q_in); a rejected chain
// overwrites it with its old position.  In-kernel draws only.  steps_out: (3, N) = tau_f, L, tau_b.
template <typename T, typename Pot, int DMAX, bool FULL, bool UNIT>
__global__ void __launch_bounds__(BLOCK) k_lane_gist_hmc(DynPrm<T> dp, Pot pot) {
    const HmcPrm<T>& prm = dp.b;
    const int64_t n0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t n = n0 + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint64_t chain = prm.chain0 + (uint64_t)n;
    const uint32_t voff = (uint32_t)threadIdx.x * (uint32_t)sizeof(T);
    const uint32_t rin = (uint32_t)prm.ldn_in * (uint32_t)sizeof(T);
    const uint32_t rout = (uint32_t)prm.ldn_out * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t bq = rows_rsrc<T, FULL>(prm.q_in, D, prm.ldn_in, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bqo = rows_rsrc<T, FULL>(prm.q_out, D, prm.ldn_out, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bpo = rows_rsrc<T, FULL>(prm.p_out, D, prm.ldn_out, prm.N, n0);
    const double pstd = sqrt((double)m * prm.kT);  // src/ensemble.py:88
    const bool f64 = (prm.flags & PBBI_DRAW_F64) != 0;
    T q[DMAX], p[DMAX], v[DMAX];
    auto load_q = [&]() {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) q[d] = load_row<T, FULL>(bq, voff, rin, d, D);
    };
    load_q();
    draw_momentum<T, DMAX>(p, D, prm.seed, prm.iter, chain, pstd, f64);
    const T oldH = hamiltonian<T, Pot, DMAX>(pot, q, p, m);
    const int tau_f = integrate_chain_dyn<T, Pot, DMAX, UNIT>(pot, q, p, v, m, prm.h, prm.L, true);
    const PhiloxOut x = rng_block(prm.seed, /*PBBI_STREAM_STEPS*/ 3u, prm.iter, chain, 0xFFFFFFFFu);
    int L = 1 + (int)(u53(x.x0, x.x1) * (double)tau_f);
    L = L > tau_f ? tau_f : L;
    load_q();
    draw_momentum<T, DMAX>(p, D, prm.seed, prm.iter, chain, pstd, f64);
    integrate_chain_dyn<T, Pot, DMAX, UNIT>(pot, q, p, v, m, prm.h, L, false);
    const T newH = hamiltonian<T, Pot, DMAX>(pot, q, p, m);
#pragma unroll
    for (int d = 0; d < DMAX; ++d) store_row<T, FULL>(bqo, voff, rout, d, D, q[d]);
    if (prm.p_out) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) store_row<T, FULL>(bpo, voff, rout, d, D, p[d]);
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d) p[d] = -p[d];
    const int tau_b = integrate_chain_dyn<T, Pot, DMAX, UNIT>(pot, q, p, v, m, prm.h, prm.L, true);
    const double hr = (double)exp((oldH - newH) * (T)pbbi_accept_beta(prm.flags, prm.kT));
    const double ratio = (L <= tau_b) ? hr * ((double)tau_f / (double)tau_b) : 0.0;
    const double u = rng_uniform(prm.seed, prm.iter, chain);
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    if (reject) {
        load_q();
#pragma unroll
        for (int d = 0; d < DMAX; ++d) store_row<T, FULL>(bqo, voff, rout, d, D, q[d]);
        if (prm.p_out) {
            if (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) {
#pragma unroll
                for (int d = 0; d < DMAX; ++d) p[d] = q[d];
            } else {
                draw_momentum<T, DMAX>(p, D, prm.seed, prm.iter, chain, pstd, f64);
            }
#pragma unroll
            for (int d = 0; d < DMAX; ++d) store_row<T, FULL>(bpo, voff, rout, d, D, p[d]);
        }
    }
    if (prm.ratio_out) prm.ratio_out[n] = (T)ratio;
    if (prm.reject_out) prm.reject_out[n] = reject ? 1 : 0;
    if (dp.steps_out) {
        dp.steps_out[n] = tau_f;
        dp.steps_out[prm.N + n] = L;
        dp.steps_out[2 * prm.N + n] = tau_b;
    }
}

template <typename T>
struct IntPrm {
    T* q;
    T* p;
    const T* mass;
    T* v_out;
    int64_t N, ldn;
    T h;
    int L, D;
};

template <typename T, typename Pot, int DMAX, int METHOD, bool FULL, bool UNIT>
__global__ void __launch_bounds__(BLOCK) k_lane_integrate(IntPrm<T> prm, Pot pot) {
    const int64_t n0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t n = n0 + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint32_t voff = (uint32_t)threadIdx.x * (uint32_t)sizeof(T);
    const uint32_t row = (uint32_t)prm.ldn * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t bq = rows_rsrc<T, FULL>(prm.q, D, prm.ldn, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bp = rows_rsrc<T, FULL>(prm.p, D, prm.ldn, prm.N, n0);
    const __amdgpu_buffer_rsrc_t bv = rows_rsrc<T, FULL>(prm.v_out, D, prm.ldn, prm.N, n0);
    T q[DMAX], p[DMAX], v[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        q[d] = load_row<T, FULL>(bq, voff, row, d, D);
        p[d] = load_row<T, FULL>(bp, voff, row, d, D);
    }
    integrate_chain<T, Pot, DMAX, METHOD, UNIT>(pot, q, p, v, m, prm.h, prm.L);
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
        {
            store_row<T, FULL>(bq, voff, row, d, D, q[d]);
            store_row<T, FULL>(bp, voff, row, d, D, p[d]);
            if (prm.v_out) store_row<T, FULL>(bv, voff, row, d, D, v[d]);
        }
}

template <typename T>
struct EvalPrm {
    const T* q;
    const T* p;
    const T* mass;
    T* U_out;
    T* grad_out;
    T* w_out;
    int64_t N, ldn;
    int D, mode;  // mode 0: eval (U, grad); 1: energy H/w; 2: ratio finish U_out = exp(U_out - H)
};

template <typename T, typename Pot, int DMAX, bool FULL>
__global__ void __launch_bounds__(BLOCK) k_lane_eval(EvalPrm<T> prm, Pot pot) {
    const int64_t n0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t n = n0 + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const uint32_t voff = (uint32_t)threadIdx.x * (uint32_t)sizeof(T);
    const uint32_t row = (uint32_t)prm.ldn * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t bq = rows_rsrc<T, FULL>(prm.q, D, prm.ldn, prm.N, n0);
    T q[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) q[d] = load_row<T, FULL>(bq, voff, row, d, D);
    if (prm.mode == 0) {
        if (prm.U_out) prm.U_out[n] = pot.U(q);
        if (prm.grad_out) {
            const __amdgpu_buffer_rsrc_t bg = rows_rsrc<T, FULL>(prm.grad_out, D, prm.ldn, prm.N, n0);
            T g[DMAX];
            pot.grad(q, g);
#pragma unroll
            for (int d = 0; d < DMAX; ++d)
                store_row<T, FULL>(bg, voff, row, d, D, g[d]);
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t bp = rows_rsrc<T, FULL>(prm.p, D, prm.ldn, prm.N, n0);
    T p[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) p[d] = load_row<T, FULL>(bp, voff, row, d, D);
    const T m = prm.mass ? prm.mass[n] : T(1);
    const T H = hamiltonian<T, Pot, DMAX>(pot, q, p, m);
    if (prm.mode == 1) {
        if (prm.U_out) prm.U_out[n] = H;
        if (prm.w_out) prm.w_out[n] = exp(-H);  // src/HMC.py:103
    } else {
        prm.U_out[n] = exp(prm.U_out[n] - H);  // src/HMC.py:115
    }
}

// ------------------------------------------------------------------- dispatch
int pick_dmax(int D) {
    for (int dm : {2, 4, 8, 16, 32, 64})
        if (D <= dm) return dm;
    return 0;
}

template <typename T, int DMAX, bool FULL>
SeparablePot<T, DMAX, FULL> make_sep(const pbbi_potential* pot) {
    return SeparablePot<T, DMAX, FULL>{(const T*)pot->d_mean, (const T*)pot->d_prec, (T)pot->cst,
                                       pot->D, pot->kind == KIND_HARMONIC ? 1 : 0};
}
template <typename T, int DMAX, bool FULL>
RosenbrockPot<T, DMAX, FULL> make_ros(const pbbi_potential* pot) {
    const double inv_s = 1.0 / pot->s;
    return RosenbrockPot<T, DMAX, FULL>{(T)pot->a, (T)pot->b, (T)inv_s, (T)pot->cst,
                                        (T)((-4.0 * pot->b) * inv_s), (T)(2.0 * inv_s),
                                        (T)((2.0 * pot->b) * inv_s), pot->D};
}

inline dim3 grid_for(int64_t N) { return dim3((unsigned)((N + BLOCK - 1) / BLOCK)); }

// run-time value -> compile-time template argument
template <typename F>
void with_dmax(int D, F&& f) {
    switch (pick_dmax(D)) {
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 16: f(std::integral_constant<int, 16>{}); break;
    case 32: f(std::integral_constant<int, 32>{}); break;
    case 64: f(std::integral_constant<int, 64>{}); break;
    default: break;
    }
}
template <typename F>
void with_bool(bool b, F&& f) {
    if (b) f(std::true_type{});
    else f(std::false_type{});
}

template <typename T, int DM, bool FULL, int METHOD, bool UNIT>
void launch_hmc_k(const pbbi_potential* pot, const HmcPrm<T>& prm, dim3 grid, hipStream_t st) {
    if (pot->kind == KIND_ROSENBROCK) {
        auto f = make_ros<T, DM, FULL>(pot);
        hipLaunchKernelGGL((k_lane_hmc<T, decltype(f), DM, METHOD, FULL, UNIT>), grid, dim3(BLOCK), 0,
                           st, prm, f);
    } else {
        auto f = make_sep<T, DM, FULL>(pot);
        hipLaunchKernelGGL((k_lane_hmc<T, decltype(f), DM, METHOD, FULL, UNIT>), grid, dim3(BLOCK), 0,
                           st, prm, f);
    }
}

template <typename T, int DM, bool FULL, int METHOD, bool UNIT>
void launch_int_k(const pbbi_potential* pot, const IntPrm<T>& prm, dim3 grid, hipStream_t st) {
    if (pot->kind == KIND_ROSENBROCK) {
        auto f = make_ros<T, DM, FULL>(pot);
        hipLaunchKernelGGL((k_lane_integrate<T, decltype(f), DM, METHOD, FULL, UNIT>), grid,
                           dim3(BLOCK), 0, st, prm, f);
    } else {
        auto f = make_sep<T, DM, FULL>(pot);
        hipLaunchKernelGGL((k_lane_integrate<T, decltype(f), DM, METHOD, FULL, UNIT>), grid,
                           dim3(BLOCK), 0, st, prm, f);
    }
}

template <typename T, int DM, bool FULL>
void launch_eval_k(const pbbi_potential* pot, const EvalPrm<T>& prm, dim3 grid, hipStream_t st) {
    if (pot->kind == KIND_ROSENBROCK) {
        auto f = make_ros<T, DM, FULL>(pot);
        hipLaunchKernelGGL((k_lane_eval<T, decltype(f), DM, FULL>), grid, dim3(BLOCK), 0, st, prm, f);
    } else {
        auto f = make_sep<T, DM, FULL>(pot);
        hipLaunchKernelGGL((k_lane_eval<T, decltype(f), DM, FULL>), grid, dim3(BLOCK), 0, st, prm, f);
    }
}

template <typename T>
int launch_hmc(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    HmcPrm<T> prm{(const T*)a.q_in, (const T*)a.p_in, (const T*)a.u_in, (const T*)a.mass,
                  (T*)a.q_out, (T*)a.p_out, (T*)a.ratio_out, a.reject_out,
                  a.N, a.ldn_in, a.ldn_out, (T)a.h, a.L, pot->D, a.flags, a.rng,
                  a.seed, a.iter, a.chain0, a.kT, a.fuse_S, a.fuse_wrap2, a.fuse_slab0,
                  (int64_t)pot->D * a.N, (T*)a.fuse_q_base};
    const dim3 grid = grid_for(a.N);
    with_dmax(pot->D, [&](auto dm) {
        constexpr int DM = decltype(dm)::value;
        with_bool(pot->D == DM, [&](auto full) {
            constexpr bool FULL = decltype(full)::value;
            with_bool(a.mass == nullptr, [&](auto unit) {
                constexpr bool UNIT = decltype(unit)::value;
                if (a.method == PBBI_LEAPFROG)
                    launch_hmc_k<T, DM, FULL, PBBI_LEAPFROG, UNIT>(pot, prm, grid, a.stream);
                else
                    launch_hmc_k<T, DM, FULL, PBBI_STORMER_VERLET, UNIT>(pot, prm, grid, a.stream);
            });
        });
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

template <typename T>
int launch_integrate(const IntegrateArgs& a) {
    const pbbi_potential* pot = a.pot;
    IntPrm<T> prm{(T*)a.q, (T*)a.p, (const T*)a.mass, (T*)a.v_out, a.N, a.ldn, (T)a.h, a.L, pot->D};
    const dim3 grid = grid_for(a.N);
    with_dmax(pot->D, [&](auto dm) {
        constexpr int DM = decltype(dm)::value;
        with_bool(pot->D == DM, [&](auto full) {
            constexpr bool FULL = decltype(full)::value;
            with_bool(a.mass == nullptr, [&](auto unit) {
                constexpr bool UNIT = decltype(unit)::value;
                if (a.method == PBBI_LEAPFROG)
                    launch_int_k<T, DM, FULL, PBBI_LEAPFROG, UNIT>(pot, prm, grid, a.stream);
                else
                    launch_int_k<T, DM, FULL, PBBI_STORMER_VERLET, UNIT>(pot, prm, grid, a.stream);
            });
        });
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

template <typename T>
int launch_eval(const EvalArgs& a, int mode) {
    const pbbi_potential* pot = a.pot;
    EvalPrm<T> prm{(const T*)a.q, (const T*)a.p, (const T*)a.mass, (T*)a.U_out, (T*)a.grad_out,
                   (T*)a.w_out, a.N, a.ldn, pot->D, mode};
    const dim3 grid = grid_for(a.N);
    with_dmax(pot->D, [&](auto dm) {
        constexpr int DM = decltype(dm)::value;
        with_bool(pot->D == DM, [&](auto full) {
            launch_eval_k<T, DM, decltype(full)::value>(pot, prm, grid, a.stream);
        });
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int check_ld(const pbbi_potential* pot, int64_t ld) {
    if ((int64_t)pot->D * ld >= ((int64_t)1 << 28))
        return pbbi_fail(PBBI_ERR_UNSUPPORTED,
                         "chain-per-lane kernels address rows with 32-bit byte offsets: D * leading "
                         "stride must be < 2^28 elements; shard the ensemble");
    return PBBI_OK;
}

// D > 64 or fp32: the chain's state lives in a device workspace (kernels_stream.hip)
inline bool streams(const pbbi_potential* pot) { return pot->dtype != PBBI_F64 || pick_dmax(pot->D) == 0; }

}  // namespace

int lane_hmc_iter(const IterArgs& a) {
    // per-chain step counts WITHOUT the U-turn stop, kick-drift-kick form, D > 32: the multi-lane kernels freeze
    // finished chains by per-lane coefficients (k_sep_hmc / k_rosg_hmc <..., DYN>)
    if (pbbi_dyn(a) && !(a.flags & PBBI_UTURN_STOP) && a.method == PBBI_LEAPFROG && a.pot->D > 32) {
        if (sepn_applies(a)) return sepn_hmc_iter(a);
        if (rosg_applies(a)) return rosg_hmc_iter(a);
    }
    if (pbbi_dyn(a)) {  // per-chain trajectory lengths: the two-lane Rosenbrock kernel or k_lane_dyn_hmc
        static const bool no_lane2_dyn = (getenv("PBBI_NO_LANE2") != nullptr);
        if (!no_lane2_dyn && lane2_applies(a) && !streams(a.pot) && a.N > 0 &&
            check_ld(a.pot, a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) == PBBI_OK)
            return lane2_hmc_iter(a);
        return lane_dyn_hmc_iter(a);
    }
    if (sepn_applies(a)) return sepn_hmc_iter(a);
    if (sepx_applies(a)) return sepx_hmc_iter(a);  // separable, 16 < D <= 256, reference operation order
    if (rosgx_applies(a)) return rosgx_hmc_iter(a);  // Rosenbrock, 32 < D <= 128, reference operation order
    if (rosg_applies(a)) return rosg_hmc_iter(a);  // PBBI_KDK_FMA, Rosenbrock, 32 < D <= 128: 4 / 8 lanes of one wave
    if (rosn_applies(a)) return rosn_hmc_iter(a);  // PBBI_KDK_FMA, Rosenbrock, 128 < D <= 256: parts in waves  // PBBI_KDK_FMA, separable, 16 < D <= 256
    if (streams(a.pot)) return stream_hmc_iter(a);
    if (int rc = check_ld(a.pot, a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out)) return rc;
    if (a.N == 0) return PBBI_OK;
    static const bool no_lane2 = (getenv("PBBI_NO_LANE2") != nullptr);  // A/B switch
    if (!no_lane2 && lane2_applies(a)) return lane2_hmc_iter(a);
    return launch_hmc<double>(a);
}
// which kernel family lane_hmc_iter hands these arguments to (pbbi_describe_run)
const char* lane_route_name(const IterArgs& a) {
    if (pbbi_dyn(a) && !(a.flags & PBBI_UTURN_STOP) && a.method == PBBI_LEAPFROG && a.pot->D > 32) {
        if (sepn_applies(a)) return "k_sep_hmc<DYN>: separable potential, parts in waves, kick-drift-kick, per-chain step counts";
        if (rosg_applies(a)) return "k_rosg_hmc<DYN>: Rosenbrock, 4 / 8 lanes per chain, kick-drift-kick, per-chain step counts";
    }
    if (pbbi_dyn(a)) {
        if (getenv("PBBI_NO_LANE2") == nullptr && lane2_applies(a) && !streams(a.pot))
            return "k_ros2_hmc<DYN>: two lanes per chain, per-chain trajectory lengths";
        return "k_lane_dyn_hmc: one chain per lane, per-chain trajectory lengths";
    }
    if (sepn_applies(a)) return "k_sep_hmc: separable potential, 16-dim parts in the waves of a workgroup, kick-drift-kick with FMA";
    if (sepx_applies(a)) return "k_sep_exact_hmc: separable potential, 16-dim parts in the waves of a workgroup, reference operation order";
    if (rosgx_applies(a)) return "k_rosg_exact_hmc: Rosenbrock, 4 / 8 lanes of a wave per chain, reference operation order";
    if (rosg_applies(a)) return "k_rosg_hmc: Rosenbrock, 4 / 8 lanes of a wave per chain, kick-drift-kick with FMA";
    if (rosn_applies(a)) return "k_rosn_hmc: Rosenbrock, 16-dim parts in the waves of a workgroup, kick-drift-kick with FMA";
    if (streams(a.pot)) return "k_stream_hmc: one chain per lane, state in a device workspace (D > 64 or fp32)";
    if (getenv("PBBI_NO_LANE2") == nullptr && lane2_applies(a))
        return (a.flags & PBBI_KDK_FMA) ? "k_ros2_hmc: Rosenbrock 16 < D <= 32, two lanes per chain, kick-drift-kick with FMA"
                                        : "k_ros2_hmc: Rosenbrock 16 < D <= 32, two lanes per chain, reference operation order";
    return "k_lane_hmc: one chain per lane, state in registers, reference operation order";
}

int lane_dyn_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (pot->dtype != PBBI_F64 || pot->D > 32 || a.method != PBBI_LEAPFROG)
        return pbbi_fail(PBBI_ERR_UNSUPPORTED,
                         "per-chain trajectory lengths: the chain-per-lane kernels take fp64, D <= 32, Leapfrog");
    if (int rc = check_ld(pot, a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out)) return rc;
    if (a.N == 0) return PBBI_OK;
    typedef double T;
    DynPrm<T> prm{{(const T*)a.q_in, (const T*)a.p_in, (const T*)a.u_in, (const T*)a.mass, (T*)a.q_out,
                   (T*)a.p_out, (T*)a.ratio_out, a.reject_out, a.N, a.ldn_in, a.ldn_out, (T)a.h, a.L, pot->D,
                   a.flags, a.rng, a.seed, a.iter, a.chain0, a.kT},
                  a.steps_in, a.steps_out};
    const dim3 grid = grid_for(a.N);
    with_dmax(pot->D, [&](auto dm) {
        constexpr int DM = decltype(dm)::value;
        if constexpr (DM <= 32) {
            with_bool(pot->D == DM, [&](auto full) {
                constexpr bool FULL = decltype(full)::value;
                with_bool(a.mass == nullptr, [&](auto unit) {
                    constexpr bool UNIT = decltype(unit)::value;
                    if (pot->kind == KIND_ROSENBROCK) {
                        auto f = make_ros<T, DM, FULL>(pot);
                        hipLaunchKernelGGL((k_lane_dyn_hmc<T, decltype(f), DM, FULL, UNIT>), grid, dim3(BLOCK), 0,
                                           a.stream, prm, f);
                    } else {
                        auto f = make_sep<T, DM, FULL>(pot);
                        hipLaunchKernelGGL((k_lane_dyn_hmc<T, decltype(f), DM, FULL, UNIT>), grid, dim3(BLOCK), 0,
                                           a.stream, prm, f);
                    }
                });
            });
        }
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// one fused GIST iteration (k_lane_gist_hmc): elementwise potentials, fp64, D <= 32, in-kernel draws; a.L = L_max,
// a.steps_out = (3, N) tau_f / L / tau_b or nullptr.  false: the caller composes the iteration from the masked
// kernels instead.
bool lane_gist_applies(const pbbi_potential* pot) {
    return pot->kind != KIND_GAUSS_DENSE && pot->kind != KIND_CUSTOM && pot->dtype == PBBI_F64 && pot->D <= 32 &&
           getenv("PBBI_GIST_COMPOSED") == nullptr;
}
int lane_gist_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (int rc = check_ld(pot, a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out)) return rc;
    if (a.N == 0) return PBBI_OK;
    if (a.q_in == a.q_out) return pbbi_fail(PBBI_ERR_INVALID, "GIST: q_out must not alias q_in");
    typedef double T;
    DynPrm<T> prm{{(const T*)a.q_in, nullptr, nullptr, (const T*)a.mass, (T*)a.q_out, (T*)a.p_out, (T*)a.ratio_out,
                   a.reject_out, a.N, a.ldn_in, a.ldn_out, (T)a.h, a.L, pot->D, a.flags, 1, a.seed, a.iter, a.chain0,
                   a.kT},
                  nullptr, a.steps_out};
    const dim3 grid = grid_for(a.N);
    with_dmax(pot->D, [&](auto dm) {
        constexpr int DM = decltype(dm)::value;
        if constexpr (DM <= 32) {
            with_bool(pot->D == DM, [&](auto full) {
                constexpr bool FULL = decltype(full)::value;
                with_bool(a.mass == nullptr, [&](auto unit) {
                    constexpr bool UNIT = decltype(unit)::value;
                    if (pot->kind == KIND_ROSENBROCK) {
                        auto f = make_ros<T, DM, FULL>(pot);
                        hipLaunchKernelGGL((k_lane_gist_hmc<T, decltype(f), DM, FULL, UNIT>), grid, dim3(BLOCK), 0,
                                           a.stream, prm, f);
                    } else {
                        auto f = make_sep<T, DM, FULL>(pot);
                        hipLaunchKernelGGL((k_lane_gist_hmc<T, decltype(f), DM, FULL, UNIT>), grid, dim3(BLOCK), 0,
                                           a.stream, prm, f);
                    }
                });
            });
        }
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int lane_fused_iterations(const IterArgs& a) {
    // kernels that keep a chain in registers across iterations: the two-lane Rosenbrock kernel, its 4 / 8-lane
    // form and the parts-in-waves separable kernel, the latter two in their kick-drift-kick forms (the order
    // of lane_hmc_iter's dispatch)
    static const int fuse = getenv("PBBI_FUSE_ITERS") ? atoi(getenv("PBBI_FUSE_ITERS")) : 16;
    static const bool no_lane2 = (getenv("PBBI_NO_LANE2") != nullptr);
    static const bool no_sep_fuse = (getenv("PBBI_NO_SEP_FUSE") != nullptr);  // A/B switch
    if (fuse <= 1 || !a.rng || a.N == 0 || pbbi_dyn(a)) return 1;
    if (sepn_applies(a)) return no_sep_fuse ? 1 : fuse;
    if (sepx_applies(a) || rosgx_applies(a)) return 1;   // (reference-order forms of those layouts: one iteration per launch)
    if (rosg_applies(a)) return no_sep_fuse ? 1 : fuse;  // Rosenbrock 32 < D <= 128, kick-drift-kick form
    if (rosn_applies(a) || streams(a.pot)) return 1;
    if (check_ld(a.pot, a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) != PBBI_OK) return 1;
    if (!no_lane2 && lane2_applies(a)) return fuse;
    // the plain chain-per-lane kernel: small chains (D <= 16, fp64), where the launch is most of an iteration
    static const bool no_lane_fuse = (getenv("PBBI_NO_LANE_FUSE") != nullptr);  // A/B switch
    return (!no_lane_fuse && a.pot->dtype == PBBI_F64 && a.pot->D <= 16) ? fuse : 1;
}
int lane_integrate(const IntegrateArgs& a) {
    if (streams(a.pot)) return stream_integrate(a);
    if (int rc = check_ld(a.pot, a.ldn)) return rc;
    if (a.N == 0) return PBBI_OK;
    return launch_integrate<double>(a);
}
int lane_eval(const EvalArgs& a) {
    if (streams(a.pot)) return stream_eval(a);
    if (int rc = check_ld(a.pot, a.ldn)) return rc;
    if (a.N == 0) return PBBI_OK;
    return launch_eval<double>(a, 0);
}
int lane_energy(const EvalArgs& a) {
    if (streams(a.pot)) return stream_energy(a);
    if (int rc = check_ld(a.pot, a.ldn)) return rc;
    if (a.N == 0) return PBBI_OK;
    const int mode = a.ratio_finish ? 2 : 1;
    return launch_eval<double>(a, mode);
}
