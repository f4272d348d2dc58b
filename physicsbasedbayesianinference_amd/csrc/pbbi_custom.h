// pbbi_custom.h -- the ensemble-HMC kernels of a USER-DEFINED potential, gfx950.
//
// The reference accepts any Python callable as potential / gradient (src/HMC.py:52-60,
// src/integrator.py:73).  A GPU kernel cannot call Python, so the user states the two functions
// in C++ (custom.py documents the contract); custom.py generates a translation unit
//
//     #include "pbbi_internal.h" / "pbbi_rng.h"
//     using T = double;  (or float)            #define PBBI_FN __device__ __forceinline__
//     namespace user { <the user's source> }
//     #include "pbbi_custom.h"
//
// and hipcc builds it into a plugin .so that libpbbi.so loads (pbbi_potential_create_custom).
// The user's functions are inlined into the kernels below -- no indirect calls on the device.
//
//     template <class Q>          PBBI_FN T    potential(const Q& q, int D, const T* prm);
//     template <class Q, class G> PBBI_FN void gradient (const Q& q, G& g, int D, const T* prm);
//
// q[j] reads and g[j] = ... writes element j of ONE chain (both are strided views of (D, N)
// arrays, chain index fastest, so a wave's accesses are 512-byte contiguous).
//
// One chain per lane; q, v, a and the gradient live in a (D, N) device workspace.  Operation
// order is the reference's (src/integrator.py:105-120, 142-163; src/HMC.py:100-115, 164-179),
// the same as kernels_stream.hip; the oracle restates it around the same user source compiled
// for the host (oracle/oracle.py::pot_custom).
#pragma once
#include <vector>

namespace pbbi_custom {

constexpr int SB = 64;  // chains per workgroup (one wave)
constexpr int CH = 8;   // rows per batch of loads

template <class E>
struct Col {  // element j of one chain's column of a (D, N) array
    E* base;
    int64_t ld;
    __device__ __forceinline__ E& operator[](int j) const { return base[(int64_t)j * ld]; }
};

template <int CNT>
__device__ __forceinline__ void load_rows(const T* col, int64_t ld, int j0, int D, T (&x)[CNT]) {
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const int j = j0 + k < D ? j0 + k : D - 1;  // clamped: no branch, always a valid address
        x[k] = col[(int64_t)j * ld];
    }
}

template <typename F>
__device__ __forceinline__ void draw_rows(uint64_t seed, uint64_t iter, uint64_t chain, int D,
                                          double pstd, bool f64, F&& visit) {
    for (int G = 0; 16 * G < D; ++G) {
        double z[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            rng_normal4d(seed, PBBI_STREAM_MOMENTUM, iter, chain, (uint32_t)((G << 2) | r), f64, z[r]);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = 16 * G + k;
            if (d < D) visit(d, (T)(z[k & 3][k >> 2] * pstd));
        }
    }
}

struct Chain {  // one chain's columns of the workspaces
    T *q, *v, *a, *g;
    int64_t ld;
    int D;
    const T* prm;
    __device__ __forceinline__ T potential() const {
        return user::potential(Col<const T>{q, ld}, D, prm);
    }
    __device__ __forceinline__ void gradient() const {
        Col<T> gg{g, ld};
        user::gradient(Col<const T>{q, ld}, gg, D, prm);
    }
};

// after gradient(): Leapfrog a0 = -g/m (:108); Stormer-Verlet qpast = q0 (kept in a),
// q1 = (q0 + v h) + (0.5 a0) h^2 (:147-150)
template <int METHOD, bool UNIT>
__device__ __forceinline__ void first_sweep(const Chain& c, T m, T h) {
    const T h2 = h * h, half = T(0.5);
    for (int j0 = 0; j0 < c.D; j0 += CH) {
        T g[CH], q[CH], v[CH];
        load_rows<CH>(c.g, c.ld, j0, c.D, g);
        if constexpr (METHOD == PBBI_STORMER_VERLET) {
            load_rows<CH>(c.q, c.ld, j0, c.D, q);
            load_rows<CH>(c.v, c.ld, j0, c.D, v);
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < c.D) {
                const T a = UNIT ? -g[k] : -g[k] / m;  // src/integrator.py:73
                if constexpr (METHOD == PBBI_LEAPFROG) {
                    c.a[(int64_t)j * c.ld] = a;
                } else {
                    c.a[(int64_t)j * c.ld] = q[k];
                    c.q[(int64_t)j * c.ld] = (q[k] + v[k] * h) + (half * a) * h2;
                }
            }
        }
    }
}

template <int METHOD, bool UNIT>
__device__ __forceinline__ void trajectory(const Chain& c, T m, T h, int L) {
    const T h2 = h * h, hh = T(0.5) * h, hh2 = T(0.5) * h2;
    c.gradient();
    first_sweep<METHOD, UNIT>(c, m, h);
    for (int s = 0; s < L; ++s) {
        if constexpr (METHOD == PBBI_LEAPFROG) {
            for (int j0 = 0; j0 < c.D; j0 += CH) {  // q += v h + a (0.5 h^2)   (:112-115)
                T q[CH], v[CH], a[CH];
                load_rows<CH>(c.q, c.ld, j0, c.D, q);
                load_rows<CH>(c.v, c.ld, j0, c.D, v);
                load_rows<CH>(c.a, c.ld, j0, c.D, a);
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (j0 + k < c.D) c.q[(int64_t)(j0 + k) * c.ld] = q[k] + (v[k] * h + a[k] * hh2);
            }
            c.gradient();
            for (int j0 = 0; j0 < c.D; j0 += CH) {  // v += (a + a') (0.5 h); a = a'   (:116-118)
                T g[CH], v[CH], a[CH];
                load_rows<CH>(c.g, c.ld, j0, c.D, g);
                load_rows<CH>(c.v, c.ld, j0, c.D, v);
                load_rows<CH>(c.a, c.ld, j0, c.D, a);
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (j0 + k < c.D) {
                        const T an = UNIT ? -g[k] : -g[k] / m;
                        c.v[(int64_t)(j0 + k) * c.ld] = v[k] + (a[k] + an) * hh;
                        c.a[(int64_t)(j0 + k) * c.ld] = an;
                    }
            }
        } else {
            c.gradient();
            for (int j0 = 0; j0 < c.D; j0 += CH) {  // q' = (2 q - qpast) + a h^2   (:152-158)
                T g[CH], q[CH], qp[CH];
                load_rows<CH>(c.g, c.ld, j0, c.D, g);
                load_rows<CH>(c.q, c.ld, j0, c.D, q);
                load_rows<CH>(c.a, c.ld, j0, c.D, qp);
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (j0 + k < c.D) {
                        const T a = UNIT ? -g[k] : -g[k] / m;
                        c.q[(int64_t)(j0 + k) * c.ld] = (T(2) * q[k] - qp[k]) + a * h2;
                        c.a[(int64_t)(j0 + k) * c.ld] = q[k];
                    }
            }
        }
    }
}

// PBBI_KDK_FMA on the workspace: vh = v + a0 h/2;  L x { q += vh h; vh += a(q) h }, last kick half.
// The kick of step s and the drift of step s+1 share one sweep (v, g, q in; v, q out): 5 accesses
// per element-step beside the user's gradient (reference order: 9), no acceleration array.
template <bool UNIT>
__device__ __forceinline__ void trajectory_kdk(const Chain& c, T m, T h, int L) {
    if (L <= 0) return;
    const T hm = UNIT ? h : h / m, hhm = T(0.5) * hm;
    c.gradient();
    for (int s = 0; s < L; ++s) {  // kick with the gradient of the current q, then drift
        const T kk = (s == 0) ? hhm : hm;
        for (int j0 = 0; j0 < c.D; j0 += CH) {
            T g[CH], v[CH], q[CH];
            load_rows<CH>(c.g, c.ld, j0, c.D, g);
            load_rows<CH>(c.v, c.ld, j0, c.D, v);
            load_rows<CH>(c.q, c.ld, j0, c.D, q);
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (j0 + k < c.D) {
                    const T vn = fma(-g[k], kk, v[k]);
                    c.v[(int64_t)(j0 + k) * c.ld] = vn;
                    c.q[(int64_t)(j0 + k) * c.ld] = fma(vn, h, q[k]);
                }
        }
        c.gradient();
    }
    for (int j0 = 0; j0 < c.D; j0 += CH) {  // closing half kick
        T g[CH], v[CH];
        load_rows<CH>(c.g, c.ld, j0, c.D, g);
        load_rows<CH>(c.v, c.ld, j0, c.D, v);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (j0 + k < c.D) c.v[(int64_t)(j0 + k) * c.ld] = fma(-g[k], hhm, v[k]);
    }
}

template <int METHOD>
__device__ __forceinline__ T final_v(T q, T v_or_qpast, T h) {
    if constexpr (METHOD == PBBI_STORMER_VERLET) return (q - v_or_qpast) / h;  // :160
    else return v_or_qpast;
}

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
    const T* prm;
    T *Wq, *Wv, *Wa, *Wg;
    // fused run (IterArgs::fuse_*, register kernels): iteration k of the call writes position slab
    // (fuse_slab0 + k) of fuse_q_base (modulo 2 when fuse_wrap2), momentum slab k, ratio / reject rows k
    int fuse_S, fuse_wrap2;
    int64_t fuse_slab0, fuse_slab;
    T* fuse_q_base;
};

template <int METHOD, bool UNIT, bool KDK = false>
__global__ void __launch_bounds__(SB) k_custom_hmc(HmcPrm prm) {
    const int64_t n = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const int64_t ld = prm.N;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint64_t chain = prm.chain0 + (uint64_t)n;
    const double pstd = prm.rng ? sqrt((double)m * prm.kT) : 1.0;  // src/ensemble.py:88
    const Chain c{prm.Wq + n, prm.Wv + n, prm.Wa + n, prm.Wg + n, ld, D, prm.prm};
    const T* qin = prm.q_in + n;
    const T* pin = prm.rng ? nullptr : prm.p_in + n;

    T pp = T(0), u;
    if (prm.rng) {
        draw_rows(prm.seed, prm.iter, chain, D, pstd, (prm.flags & PBBI_DRAW_F64) != 0, [&](int d, T p) {
            pp += p * p;
            c.v[(int64_t)d * ld] = UNIT ? p : p / m;
        });
        u = (T)rng_uniform(prm.seed, prm.iter, chain);
    } else {
        for (int j0 = 0; j0 < D; j0 += CH) {
            T p[CH];
            load_rows<CH>(pin, prm.ldn_in, j0, D, p);
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (j0 + k < D) {
                    pp += p[k] * p[k];
                    c.v[(int64_t)(j0 + k) * ld] = UNIT ? p[k] : p[k] / m;
                }
        }
        u = prm.u_in[n];
    }
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH];
        load_rows<CH>(qin, prm.ldn_in, j0, D, q);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (j0 + k < D) c.q[(int64_t)(j0 + k) * ld] = q[k];
    }
    const T oldH = T(0.5) * pp / m + c.potential();  // src/HMC.py:100-102
    if constexpr (KDK) trajectory_kdk<UNIT>(c, m, prm.h, prm.L);
    else trajectory<METHOD, UNIT>(c, m, prm.h, prm.L);
    T pp1 = T(0);
    const T* vsrc = METHOD == PBBI_STORMER_VERLET ? c.a : c.v;
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH], v[CH];
        load_rows<CH>(c.q, ld, j0, D, q);
        load_rows<CH>(vsrc, ld, j0, D, v);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (j0 + k < D) {
                const T vj = final_v<METHOD>(q[k], v[k], prm.h);
                const T p = UNIT ? vj : vj * m;
                pp1 += p * p;
            }
    }
    const T newH = T(0.5) * pp1 / m + c.potential();
    const T ratio = exp((oldH - newH) * (T)pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    // mask = u > min(1, ratio); NaN ratio compares False => accepted (src/HMC.py:168-173)
    const bool reject = (ratio == ratio) && (u > (ratio < T(1) ? ratio : T(1)));

    T* qo = prm.q_out + n;
    T* po = prm.p_out ? prm.p_out + n : nullptr;
    const bool compat = (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) != 0;
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH], v[CH], qi[CH];
        load_rows<CH>(c.q, ld, j0, D, q);
        load_rows<CH>(vsrc, ld, j0, D, v);
        load_rows<CH>(qin, prm.ldn_in, j0, D, qi);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                qo[(int64_t)j * prm.ldn_out] = reject ? qi[k] : q[k];  // :175
                if (po) {
                    const T vj = final_v<METHOD>(q[k], v[k], prm.h);
                    T pj = UNIT ? vj : vj * m;
                    if (reject) pj = compat ? qi[k] : (pin ? pin[(int64_t)j * prm.ldn_in] : T(0));
                    po[(int64_t)j * prm.ldn_out] = pj;  // :176 (compat: p <- oldQ)
                }
            }
        }
    }
    if (po && reject && !compat && prm.rng)
        draw_rows(prm.seed, prm.iter, chain, D, pstd, (prm.flags & PBBI_DRAW_F64) != 0,
                  [&](int d, T p) { po[(int64_t)d * prm.ldn_out] = p; });
    if (prm.ratio_out) prm.ratio_out[n] = ratio;
    if (prm.reject_out) prm.reject_out[n] = reject ? 1 : 0;
}

struct IntPrm {
    T* q;
    T* p;
    const T* mass;
    T* v_out;
    int64_t N, ldn;
    T h;
    int L, D;
    const T* prm;
    T *Wq, *Wv, *Wa, *Wg;
};

template <int METHOD, bool UNIT>
__global__ void __launch_bounds__(SB) k_custom_integrate(IntPrm prm) {
    const int64_t n = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const int64_t ld = prm.N;
    const T m = UNIT ? T(1) : prm.mass[n];
    const Chain c{prm.Wq + n, prm.Wv + n, prm.Wa + n, prm.Wg + n, ld, D, prm.prm};
    T *q = prm.q + n, *p = prm.p + n;
    T* vo = prm.v_out ? prm.v_out + n : nullptr;
    for (int j0 = 0; j0 < D; j0 += CH) {
        T x[CH], y[CH];
        load_rows<CH>(q, prm.ldn, j0, D, x);
        load_rows<CH>(p, prm.ldn, j0, D, y);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (j0 + k < D) {
                c.q[(int64_t)(j0 + k) * ld] = x[k];
                c.v[(int64_t)(j0 + k) * ld] = UNIT ? y[k] : y[k] / m;
            }
    }
    trajectory<METHOD, UNIT>(c, m, prm.h, prm.L);
    const T* vsrc = METHOD == PBBI_STORMER_VERLET ? c.a : c.v;
    for (int j0 = 0; j0 < D; j0 += CH) {
        T x[CH], v[CH];
        load_rows<CH>(c.q, ld, j0, D, x);
        load_rows<CH>(vsrc, ld, j0, D, v);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                const T vj = final_v<METHOD>(x[k], v[k], prm.h);
                q[(int64_t)j * prm.ldn] = x[k];
                p[(int64_t)j * prm.ldn] = UNIT ? vj : vj * m;
                if (vo) vo[(int64_t)j * prm.ldn] = vj;
            }
        }
    }
}

struct EvalPrm {
    const T* q;
    const T* p;
    const T* mass;
    T* U_out;
    T* grad_out;
    T* w_out;
    int64_t N, ldn;
    int D, mode;  // 0: eval (U, grad); 1: energy H / w; 2: ratio finish U_out = exp(U_out - H)
    const T* prm;
};

__global__ void __launch_bounds__(SB) k_custom_eval(EvalPrm prm) {
    const int64_t n = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (n >= prm.N) return;
    const int D = prm.D;
    const Col<const T> q{prm.q + n, prm.ldn};
    if (prm.mode == 0) {
        if (prm.U_out) prm.U_out[n] = user::potential(q, D, prm.prm);
        if (prm.grad_out) {
            Col<T> g{prm.grad_out + n, prm.ldn};
            user::gradient(q, g, D, prm.prm);
        }
        return;
    }
    T pp = T(0);
    for (int j0 = 0; j0 < D; j0 += CH) {
        T p[CH];
        load_rows<CH>(prm.p + n, prm.ldn, j0, D, p);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (j0 + k < D) pp += p[k] * p[k];
    }
    const T m = prm.mass ? prm.mass[n] : T(1);
    const T H = T(0.5) * pp / m + user::potential(q, D, prm.prm);
    if (prm.mode == 1) {
        if (prm.U_out) prm.U_out[n] = H;
        if (prm.w_out) prm.w_out[n] = exp(-H);  // src/HMC.py:103
    } else {
        prm.U_out[n] = exp(prm.U_out[n] - H);  // src/HMC.py:115
    }
}


// ---- register-resident variant ----------------------------------------------------------------
// custom.py compiles the plugin for ONE dimension (-DPBBI_D=<D>).  When the chain fits a lane's
// registers (q, v, a, g: 4*D values) the user's functions are called on register arrays: their
// loops over D have a constant trip count after inlining, unroll, and index the arrays with
// constants -- the same shape as kernels_lane.hip (one chain per lane, state on chip for all L
// steps, HBM touched once on the way in and once on the way out), at the same operation order.
#if defined(PBBI_D)
// reference operation order: q, v, a, g live (4*D values); kick-drift-kick/FMA form (PBBI_KDK_FMA):
// q, v, g (3*D values) -- twice the dimension fits
constexpr int REG_DMAX = sizeof(T) == 8 ? 16 : 32;
constexpr bool REG_OK = (PBBI_D) <= REG_DMAX;
constexpr bool REG_OK_KDK = (PBBI_D) <= 2 * REG_DMAX;
constexpr int RD = REG_OK_KDK ? (PBBI_D) : 1;
#else
constexpr bool REG_OK = false, REG_OK_KDK = false;
constexpr int RD = 1;
#endif
constexpr int RB = 256;  // chains per workgroup

struct Vec {
    T x[RD];
    __device__ __forceinline__ T& operator[](int j) { return x[j]; }
    __device__ __forceinline__ const T& operator[](int j) const { return x[j]; }
};

template <int METHOD, bool UNIT, bool KDK = false>
__device__ __forceinline__ void reg_trajectory(Vec& q, Vec& v, const T* prm, T m, T h, int L) {
    const T h2 = h * h, half = T(0.5), hh = half * h, hh2 = half * h2;
    if constexpr (KDK) {
        // PBBI_KDK_FMA: vh = v + a0 h/2;  L x { q += vh h; vh += a(q) h }, the last kick a half kick
        const T hm = UNIT ? h : h / m, hhm = half * hm;
        Vec g;
        if (L > 0) {
            user::gradient(q, g, RD, prm);
#pragma unroll
            for (int j = 0; j < RD; ++j) v[j] = fma(-g[j], hhm, v[j]);
            for (int s = 0; s < L; ++s) {
#pragma unroll
                for (int j = 0; j < RD; ++j) q[j] = fma(v[j], h, q[j]);
                user::gradient(q, g, RD, prm);
                const T kk = (s + 1 < L) ? hm : hhm;
#pragma unroll
                for (int j = 0; j < RD; ++j) v[j] = fma(-g[j], kk, v[j]);
            }
        }
        return;
    }
    Vec a, g;
    if constexpr (METHOD == PBBI_LEAPFROG) {
        user::gradient(q, g, RD, prm);
#pragma unroll
        for (int j = 0; j < RD; ++j) a[j] = UNIT ? -g[j] : -g[j] / m;  // src/integrator.py:73,108
        for (int s = 0; s < L; ++s) {
#pragma unroll
            for (int j = 0; j < RD; ++j) q[j] += (v[j] * h + a[j] * hh2);  // :112-115
            user::gradient(q, g, RD, prm);
#pragma unroll
            for (int j = 0; j < RD; ++j) {  // :116-118
                const T an = UNIT ? -g[j] : -g[j] / m;
                v[j] += (a[j] + an) * hh;
                a[j] = an;
            }
        }
    } else {
        Vec qpast = q;  // :147
        user::gradient(q, g, RD, prm);
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const T aj = UNIT ? -g[j] : -g[j] / m;
            q[j] = (q[j] + v[j] * h) + (half * aj) * h2;  // :148-150
        }
        for (int s = 0; s < L; ++s) {
            user::gradient(q, g, RD, prm);
#pragma unroll
            for (int j = 0; j < RD; ++j) {  // :152-158
                const T aj = UNIT ? -g[j] : -g[j] / m;
                const T cur = q[j];
                q[j] = (T(2) * cur - qpast[j]) + aj * h2;
                qpast[j] = cur;
            }
        }
#pragma unroll
        for (int j = 0; j < RD; ++j) v[j] = (q[j] - qpast[j]) / h;  // :160
    }
}

template <int METHOD, bool UNIT, bool KDK = false>
__global__ void __launch_bounds__(RB) k_custom_reg_hmc(HmcPrm prm) {
    const int64_t n = (int64_t)blockIdx.x * RB + threadIdx.x;
    if (n >= prm.N) return;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint64_t chain = prm.chain0 + (uint64_t)n;
    const double pstd = prm.rng ? sqrt((double)m * prm.kT) : 1.0;  // src/ensemble.py:88
    const T* pin = prm.rng ? nullptr : prm.p_in + n;
    Vec q, v;
#pragma unroll
    for (int j = 0; j < RD; ++j) q[j] = prm.q_in[(int64_t)j * prm.ldn_in + n];
    // A fused run keeps the chain in these registers for fuse_S iterations; the potential energy of the
    // position an iteration starts from is the one the previous iteration already evaluated (its proposal's
    // if it accepted, its own start's if it rejected) and is carried instead of evaluated again -- one call
    // of the user's potential per iteration instead of two, the same value.
    T U_cur = T(0);
    const int nfuse = prm.fuse_S > 1 ? prm.fuse_S : 1;
#pragma nounroll
    for (int kf = 0; kf < nfuse; ++kf) {
    const uint64_t iter_k = prm.iter + (uint64_t)kf;
    const int64_t s_out = prm.fuse_wrap2 ? ((prm.fuse_slab0 + kf) & 1) : prm.fuse_slab0 + kf;
    const int64_t s_prev = prm.fuse_wrap2 ? ((prm.fuse_slab0 + kf - 1) & 1) : prm.fuse_slab0 + kf - 1;
    const T* qin = (kf > 0 ? prm.fuse_q_base + s_prev * prm.fuse_slab : prm.q_in) + n;
    const int64_t ld_in = kf > 0 ? prm.ldn_out : prm.ldn_in;
    T* const q_out_k = nfuse > 1 ? prm.fuse_q_base + s_out * prm.fuse_slab : prm.q_out;
    T* const p_out_k = (prm.p_out && nfuse > 1) ? prm.p_out + (int64_t)kf * prm.fuse_slab : prm.p_out;
    auto draw = [&]() {
#pragma unroll
        for (int G = 0; G < (RD + 15) / 16; ++G)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * G + r < RD) {
                    double z[4];
                    rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, iter_k, chain, (uint32_t)((G << 2) | r),
                                 (prm.flags & PBBI_DRAW_F64) != 0, z);
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int d = 16 * G + r + 4 * sl;
                        if (d < RD) v[d] = (T)(z[sl] * pstd);
                    }
                }
            }
    };
    T u;
    if (prm.rng) {
        draw();
        u = (T)rng_uniform(prm.seed, iter_k, chain);
    } else {
#pragma unroll
        for (int j = 0; j < RD; ++j) v[j] = pin[(int64_t)j * prm.ldn_in];
        u = prm.u_in[n];
    }
    T pp = T(0);
#pragma unroll
    for (int j = 0; j < RD; ++j) pp += v[j] * v[j];
    const T U_old = kf > 0 ? U_cur : user::potential(q, RD, prm.prm);
    const T oldH = T(0.5) * pp / m + U_old;  // src/HMC.py:100-102
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < RD; ++j) v[j] = v[j] / m;
    }
    reg_trajectory<METHOD, UNIT, KDK>(q, v, prm.prm, m, prm.h, prm.L);
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < RD; ++j) v[j] = v[j] * m;  // p = v*m
    }
    T pp1 = T(0);
#pragma unroll
    for (int j = 0; j < RD; ++j) pp1 += v[j] * v[j];
    const T U_new = user::potential(q, RD, prm.prm);
    const T newH = T(0.5) * pp1 / m + U_new;
    const T ratio = exp((oldH - newH) * (T)pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const bool reject = (ratio == ratio) && (u > (ratio < T(1) ? ratio : T(1)));
    U_cur = reject ? U_old : U_new;
    if (reject) {
#pragma unroll
        for (int j = 0; j < RD; ++j) q[j] = qin[(int64_t)j * ld_in];  // :175
        if (prm.p_out) {
            if (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) {  // :176  p <- oldQ
#pragma unroll
                for (int j = 0; j < RD; ++j) v[j] = q[j];
            } else if (prm.rng) {
                draw();
            } else {
#pragma unroll
                for (int j = 0; j < RD; ++j) v[j] = pin[(int64_t)j * prm.ldn_in];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RD; ++j) q_out_k[(int64_t)j * prm.ldn_out + n] = q[j];
    if (prm.p_out) {
#pragma unroll
        for (int j = 0; j < RD; ++j) p_out_k[(int64_t)j * prm.ldn_out + n] = v[j];
    }
    if (prm.ratio_out) prm.ratio_out[(int64_t)kf * prm.N + n] = ratio;
    if (prm.reject_out) prm.reject_out[(int64_t)kf * prm.N + n] = reject ? 1 : 0;
    }  // kf
}

template <int METHOD, bool UNIT>
__global__ void __launch_bounds__(RB) k_custom_reg_integrate(IntPrm prm) {
    const int64_t n = (int64_t)blockIdx.x * RB + threadIdx.x;
    if (n >= prm.N) return;
    const T m = UNIT ? T(1) : prm.mass[n];
    Vec q, v;
#pragma unroll
    for (int j = 0; j < RD; ++j) {
        q[j] = prm.q[(int64_t)j * prm.ldn + n];
        const T p = prm.p[(int64_t)j * prm.ldn + n];
        v[j] = UNIT ? p : p / m;
    }
    reg_trajectory<METHOD, UNIT>(q, v, prm.prm, m, prm.h, prm.L);
#pragma unroll
    for (int j = 0; j < RD; ++j) {
        prm.q[(int64_t)j * prm.ldn + n] = q[j];
        prm.p[(int64_t)j * prm.ldn + n] = UNIT ? v[j] : v[j] * m;
        if (prm.v_out) prm.v_out[(int64_t)j * prm.ldn + n] = v[j];
    }
}


inline dim3 grid_for(int64_t N) { return dim3((unsigned)((N + SB - 1) / SB)); }

template <typename F>
void with_method_unit(int method, bool unit, F&& f) {
    if (method == PBBI_LEAPFROG) {
        if (unit) f(std::integral_constant<int, PBBI_LEAPFROG>{}, std::true_type{});
        else f(std::integral_constant<int, PBBI_LEAPFROG>{}, std::false_type{});
    } else {
        if (unit) f(std::integral_constant<int, PBBI_STORMER_VERLET>{}, std::true_type{});
        else f(std::integral_constant<int, PBBI_STORMER_VERLET>{}, std::false_type{});
    }
}

// launchers of the register-resident kernels; templates so that a plugin whose D does not fit
// never instantiates them (nor the user's functions on a one-element array)
template <bool R, bool KDK = false>
int reg_hmc(const IterArgs* a) {
    if constexpr (R) {
        const pbbi_potential* pot = a->pot;
        HmcPrm prm{(const T*)a->q_in, (const T*)a->p_in, (const T*)a->u_in, (const T*)a->mass,
                   (T*)a->q_out, (T*)a->p_out, (T*)a->ratio_out, a->reject_out,
                   a->N, a->ldn_in, a->ldn_out, (T)a->h, a->L, pot->D, a->flags, a->rng,
                   a->seed, a->iter, a->chain0, a->kT, (const T*)pot->d_params, nullptr, nullptr,
                   nullptr, nullptr, a->fuse_S, a->fuse_wrap2, a->fuse_slab0, (int64_t)pot->D * a->N,
                   (T*)a->fuse_q_base};
        if constexpr (KDK) {  // Leapfrog only
            if (a->mass == nullptr)
                hipLaunchKernelGGL((k_custom_reg_hmc<PBBI_LEAPFROG, true, true>),
                                   dim3((unsigned)((a->N + RB - 1) / RB)), dim3(RB), 0, a->stream, prm);
            else
                hipLaunchKernelGGL((k_custom_reg_hmc<PBBI_LEAPFROG, false, true>),
                                   dim3((unsigned)((a->N + RB - 1) / RB)), dim3(RB), 0, a->stream, prm);
        } else {
            with_method_unit(a->method, a->mass == nullptr, [&](auto meth, auto unit) {
                hipLaunchKernelGGL((k_custom_reg_hmc<decltype(meth)::value, decltype(unit)::value>),
                                   dim3((unsigned)((a->N + RB - 1) / RB)), dim3(RB), 0, a->stream, prm);
            });
        }
        return (int)hipGetLastError();
    } else {
        return -1;
    }
}
template <bool R>
int reg_integrate(const IntegrateArgs* a) {
    if constexpr (R) {
        const pbbi_potential* pot = a->pot;
        IntPrm prm{(T*)a->q, (T*)a->p, (const T*)a->mass, (T*)a->v_out, a->N, a->ldn, (T)a->h, a->L,
                   pot->D, (const T*)pot->d_params, nullptr, nullptr, nullptr, nullptr};
        with_method_unit(a->method, a->mass == nullptr, [&](auto meth, auto unit) {
            hipLaunchKernelGGL((k_custom_reg_integrate<decltype(meth)::value, decltype(unit)::value>),
                               dim3((unsigned)((a->N + RB - 1) / RB)), dim3(RB), 0, a->stream, prm);
        });
        return (int)hipGetLastError();
    } else {
        return -1;
    }
}

}  // namespace pbbi_custom

// ---- the plugin's exported surface (resolved by pbbi_potential_create_custom) -----------------
// return value: 0, a hipError_t, or -1 when the workspace could not be allocated
extern "C" {

int pbbi_plugin_abi(void) { return PBBI_PLUGIN_ABI; }
int pbbi_plugin_dtype(void) { return sizeof(T) == 8 ? PBBI_F64 : PBBI_F32; }

int pbbi_plugin_hmc_iter(const IterArgs* a) {
    using namespace pbbi_custom;
    const pbbi_potential* pot = a->pot;
    // the chain fits the lane's registers: kick-drift-kick/FMA form up to twice the dimension
    if (REG_OK_KDK && pot->D == RD && (a->flags & PBBI_KDK_FMA) && a->method == PBBI_LEAPFROG)
        return reg_hmc<REG_OK_KDK, true>(a);
    if (REG_OK && pot->D == RD) return reg_hmc<REG_OK>(a);
    if (a->fuse_S > 1) {  // the workspace kernels take one iteration per launch: unroll the fused call here
        const int64_t slab_e = (int64_t)pot->D * a->N;
        for (int k = 0; k < a->fuse_S; ++k) {
            IterArgs it = *a;
            const int64_t s_out = a->fuse_wrap2 ? ((a->fuse_slab0 + k) & 1) : a->fuse_slab0 + k;
            const int64_t s_prev = a->fuse_wrap2 ? ((a->fuse_slab0 + k - 1) & 1) : a->fuse_slab0 + k - 1;
            it.fuse_S = 1;
            if (k > 0) { it.q_in = (const T*)a->fuse_q_base + s_prev * slab_e; it.ldn_in = a->ldn_out; }
            it.q_out = (T*)a->fuse_q_base + s_out * slab_e;
            if (a->p_out) it.p_out = (T*)a->p_out + (int64_t)k * slab_e;
            if (a->ratio_out) it.ratio_out = (T*)a->ratio_out + (int64_t)k * a->N;
            if (a->reject_out) it.reject_out = a->reject_out + (int64_t)k * a->N;
            it.iter = a->iter + (uint64_t)k;
            if (k > 0) it.scratch_used = nullptr;
            if (int rc = pbbi_plugin_hmc_iter(&it)) return rc;
        }
        return 0;
    }
    Scratch ws(*a);
    const size_t slab = (size_t)pot->D * a->N * sizeof(T);
    T *Wq = (T*)ws.get(slab), *Wv = (T*)ws.get(slab), *Wa = (T*)ws.get(slab), *Wg = (T*)ws.get(slab);
    if (!Wq || !Wv || !Wa || !Wg) return -1;
    HmcPrm prm{(const T*)a->q_in, (const T*)a->p_in, (const T*)a->u_in, (const T*)a->mass,
               (T*)a->q_out, (T*)a->p_out, (T*)a->ratio_out, a->reject_out,
               a->N, a->ldn_in, a->ldn_out, (T)a->h, a->L, pot->D, a->flags, a->rng,
               a->seed, a->iter, a->chain0, a->kT, (const T*)pot->d_params, Wq, Wv, Wa, Wg};
    if ((a->flags & PBBI_KDK_FMA) && a->method == PBBI_LEAPFROG) {
        if (a->mass == nullptr)
            hipLaunchKernelGGL((k_custom_hmc<PBBI_LEAPFROG, true, true>), grid_for(a->N), dim3(SB), 0,
                               a->stream, prm);
        else
            hipLaunchKernelGGL((k_custom_hmc<PBBI_LEAPFROG, false, true>), grid_for(a->N), dim3(SB), 0,
                               a->stream, prm);
    } else {
        with_method_unit(a->method, a->mass == nullptr, [&](auto meth, auto unit) {
            hipLaunchKernelGGL((k_custom_hmc<decltype(meth)::value, decltype(unit)::value>),
                               grid_for(a->N), dim3(SB), 0, a->stream, prm);
        });
    }
    return (int)hipGetLastError();
}

int pbbi_plugin_integrate(const IntegrateArgs* a) {
    using namespace pbbi_custom;
    const pbbi_potential* pot = a->pot;
    if (REG_OK && pot->D == RD) return reg_integrate<REG_OK>(a);
    Scratch ws(a->stream);
    const size_t slab = (size_t)pot->D * a->N * sizeof(T);
    T *Wq = (T*)ws.get(slab), *Wv = (T*)ws.get(slab), *Wa = (T*)ws.get(slab), *Wg = (T*)ws.get(slab);
    if (!Wq || !Wv || !Wa || !Wg) return -1;
    IntPrm prm{(T*)a->q, (T*)a->p, (const T*)a->mass, (T*)a->v_out, a->N, a->ldn, (T)a->h, a->L,
               pot->D, (const T*)pot->d_params, Wq, Wv, Wa, Wg};
    with_method_unit(a->method, a->mass == nullptr, [&](auto meth, auto unit) {
        hipLaunchKernelGGL((k_custom_integrate<decltype(meth)::value, decltype(unit)::value>),
                           grid_for(a->N), dim3(SB), 0, a->stream, prm);
    });
    return (int)hipGetLastError();
}

int pbbi_plugin_eval(const EvalArgs* a, int mode) {
    using namespace pbbi_custom;
    const pbbi_potential* pot = a->pot;
    EvalPrm prm{(const T*)a->q, (const T*)a->p, (const T*)a->mass, (T*)a->U_out, (T*)a->grad_out,
                (T*)a->w_out, a->N, a->ldn, pot->D, mode, (const T*)pot->d_params};
    hipLaunchKernelGGL(k_custom_eval, grid_for(a->N), dim3(SB), 0, a->stream, prm);
    return (int)hipGetLastError();
}

}  // extern "C"
