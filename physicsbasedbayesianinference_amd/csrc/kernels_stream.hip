// kernels_stream.hip -- chain-per-lane kernels for ANY D and for fp32: the state of a chain no
// longer fits its lane's registers (kernels_lane.hip stops at D = 64, fp64), so q, v, a live in
// a (D, N) device workspace and every leapfrog step is ONE sequential sweep over the dimensions
// of the chain (harmonic, diagonal Gaussian, Rosenbrock), gfx950.
//
// Same arithmetic, same order as kernels_lane.hip / the oracle (src/integrator.py:105-120,
// 142-163; src/HMC.py:100-115,164-179): fp64 results are bit-identical to the oracle.  A sweep
// fuses "drift element j+1" with "kick element j" (the nearest-neighbour gradient of element j
// needs the drifted q_j and q_{j+1}), so a step reads q, v, a once and writes them once:
// 48 B (fp64) per element-step, HBM / L2 bound.  Rows are loaded CH at a time so one wave keeps
// 3*CH 512-byte loads in flight; one wave per workgroup (N/64 workgroups) for dispatch balance.
#include <vector>

#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

constexpr int SB = 64;  // chains per workgroup (one wave)
constexpr int CH = 8;   // rows per batch of loads

enum { K_HARM = 0, K_GAUSS = 1, K_ROS = 2 };

template <typename T>
struct SPot {
    const T* mean;
    const T* prec;
    T a, b, inv_s, cst, c1, c2, c3;  // Rosenbrock: c1 = (-4b)/s, c2 = 2/s, c3 = (2b)/s
    int D;
};

__device__ __forceinline__ double fm(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// g_j from q_j, q_{j+1} and the term carried from element j-1; j ascending (kernels_lane.hip
// SeparablePot::grad_each / RosenbrockPot::grad_each element by element)
template <typename T, int K>
__device__ __forceinline__ T grad_elem(const SPot<T>& pot, int j, T qj, T qj1, T& carry) {
    if constexpr (K == K_ROS) {
        T gi = carry;
        carry = T(0);
        if (j + 1 < pot.D) {
            const T t = fm(-qj, qj, qj1);
            gi += fm(pot.c1 * qj, t, -(pot.c2 * (pot.a - qj)));
            carry = pot.c3 * t;
        }
        return gi;
    } else {
        return pot.prec[j] * (qj - pot.mean[j]);
    }
}

template <typename T, int K>
struct UAcc {  // U(q) accumulated in the order of SeparablePot::U / RosenbrockPot::U
    T s1 = T(0), s2 = T(0);
    __device__ __forceinline__ void add(const SPot<T>& pot, int j, T qj, T qj1) {
        if constexpr (K == K_ROS) {
            if (j + 1 < pot.D) {
                const T t = fm(-qj, qj, qj1);
                s1 = fm(pot.b * t, t, s1);
                const T r = pot.a - qj;
                s2 = fm(r, r, s2);
            }
        } else if constexpr (K == K_HARM) {
            s1 += pot.prec[j] * (qj * qj);
        } else {
            const T x = qj - pot.mean[j];
            s1 += (pot.prec[j] * x) * x;
        }
    }
    __device__ __forceinline__ T finish(const SPot<T>& pot) const {
        if constexpr (K == K_ROS) return (s1 + s2) * pot.inv_s + pot.cst;
        else return T(0.5) * s1 + pot.cst;
    }
};

// rows j0 .. j0+CNT-1 of one chain's column (row index clamped to D-1: no branch, always valid)
template <typename T, int CNT>
__device__ __forceinline__ void load_rows(const T* col, int64_t ld, int j0, int D, T (&x)[CNT]) {
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const int j = j0 + k < D ? j0 + k : D - 1;
        x[k] = col[(int64_t)j * ld];
    }
}

// the momentum of one chain in dimension order: one Philox block per four dims (include/pbbi.h)
template <typename T, typename F>
__device__ __forceinline__ void draw_rows(uint64_t seed, uint64_t iter, uint64_t chain, int D,
                                          double pstd, bool f64, F&& visit) {
    for (int G = 0; 16 * G < D; ++G) {
        double z[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            rng_normal4d(seed, PBBI_STREAM_MOMENTUM, iter, chain, (uint32_t)((G << 2) | r), f64, z[r]);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = 16 * G + k;  // = 16G + r + 4*slot with r = k & 3, slot = k >> 2
            if (d < D) visit(d, (T)(z[k & 3][k >> 2] * pstd));
        }
    }
}

// ---- sweeps over one chain; wq / wv / wa point at the chain's column of the workspaces ---------
// first sweep: working copy of q, U(q0) and
//   Leapfrog        a0 = getAccel(q0)                                   (src/integrator.py:108)
//   Stormer-Verlet  qpast = q0 (kept in wa), q1 = (q0 + v h) + (0.5 a0) h^2   (:147-150)
template <typename T, int K, int METHOD, bool UNIT>
__device__ __forceinline__ T init_sweep(const SPot<T>& pot, const T* qs, int64_t lds, T* wq,
                                        const T* wv, T* wa, int64_t ld, T m, T h) {
    const int D = pot.D;
    const T h2 = h * h, half = T(0.5);
    UAcc<T, K> u;
    T carry = T(0);
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH + 1], v[CH];
        load_rows<T, CH + 1>(qs, lds, j0, D, q);
        if constexpr (METHOD == PBBI_STORMER_VERLET) load_rows<T, CH>(wv, ld, j0, D, v);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                const T qj1 = j + 1 < D ? q[k + 1] : T(0);
                u.add(pot, j, q[k], qj1);
                const T g = grad_elem<T, K>(pot, j, q[k], qj1, carry);
                const T a = UNIT ? -g : -g / m;  // src/integrator.py:73
                if constexpr (METHOD == PBBI_LEAPFROG) {
                    wq[(int64_t)j * ld] = q[k];
                    wa[(int64_t)j * ld] = a;
                } else {
                    wa[(int64_t)j * ld] = q[k];
                    wq[(int64_t)j * ld] = (q[k] + v[k] * h) + (half * a) * h2;
                }
            }
        }
    }
    return u.finish(pot);
}

// one Leapfrog step (src/integrator.py:111-118): drift of element j+1 fused with the kick of j
template <typename T, int K, bool UNIT>
__device__ __forceinline__ void leapfrog_sweep(const SPot<T>& pot, T* wq, T* wv, T* wa, int64_t ld,
                                               T m, T h, T hh, T hh2) {
    const int D = pot.D;
    T carry = T(0);
    T pq = T(0), pv = T(0), pa = T(0);  // drifted element j0-1, waiting for q_{j0}
    auto kick = [&](int j, T qj, T qj1, T vj, T aj) {
        const T g = grad_elem<T, K>(pot, j, qj, qj1, carry);
        const T an = UNIT ? -g : -g / m;
        wv[(int64_t)j * ld] = vj + (aj + an) * hh;
        wa[(int64_t)j * ld] = an;
    };
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH], v[CH], a[CH];
        load_rows<T, CH>(wq, ld, j0, D, q);
        load_rows<T, CH>(wv, ld, j0, D, v);
        load_rows<T, CH>(wa, ld, j0, D, a);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            q[k] += (v[k] * h + a[k] * hh2);
            if (j0 + k < D) wq[(int64_t)(j0 + k) * ld] = q[k];
        }
        if (j0 > 0) kick(j0 - 1, pq, q[0], pv, pa);
#pragma unroll
        for (int k = 0; k + 1 < CH; ++k) {
            const int j = j0 + k;
            if (j < D) kick(j, q[k], j + 1 < D ? q[k + 1] : T(0), v[k], a[k]);
        }
        pq = q[CH - 1];
        pv = v[CH - 1];
        pa = a[CH - 1];
    }
    if (D % CH == 0) kick(D - 1, pq, T(0), pv, pa);
}

// one Stormer-Verlet step (src/integrator.py:152-158), in place: q_{j+1} is read before the
// sweep overwrites it; wa holds qpast
template <typename T, int K, bool UNIT>
__device__ __forceinline__ void verlet_sweep(const SPot<T>& pot, T* wq, T* wa, int64_t ld, T m,
                                             T h2) {
    const int D = pot.D;
    T carry = T(0);
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH + 1], qp[CH];
        load_rows<T, CH + 1>(wq, ld, j0, D, q);
        load_rows<T, CH>(wa, ld, j0, D, qp);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                const T g = grad_elem<T, K>(pot, j, q[k], j + 1 < D ? q[k + 1] : T(0), carry);
                const T a = UNIT ? -g : -g / m;
                wq[(int64_t)j * ld] = (T(2) * q[k] - qp[k]) + a * h2;
                wa[(int64_t)j * ld] = q[k];
            }
        }
    }
}

// velocity of element j at the end of the trajectory (Stormer-Verlet: (q - qpast)/h, :160)
template <typename T, int METHOD>
__device__ __forceinline__ T final_v(T q, T v, T qpast, T h) {
    if constexpr (METHOD == PBBI_STORMER_VERLET) return (q - qpast) / h;
    else return v;
}

// H(q, p = v m) of the integrated state (src/HMC.py:100-102)
template <typename T, int K, int METHOD, bool UNIT>
__device__ __forceinline__ T final_energy(const SPot<T>& pot, const T* wq, const T* wv, const T* wa,
                                          int64_t ld, T m, T h) {
    const int D = pot.D;
    UAcc<T, K> u;
    T pp = T(0);
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH + 1], v[CH];
        load_rows<T, CH + 1>(wq, ld, j0, D, q);
        load_rows<T, CH>(METHOD == PBBI_STORMER_VERLET ? wa : wv, ld, j0, D, v);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                const T vj = final_v<T, METHOD>(q[k], v[k], v[k], h);
                const T p = UNIT ? vj : vj * m;
                pp += p * p;
                u.add(pot, j, q[k], j + 1 < D ? q[k + 1] : T(0));
            }
        }
    }
    return T(0.5) * pp / m + u.finish(pot);
}

template <typename T, int K, int METHOD, bool UNIT>
__device__ __forceinline__ void trajectory(const SPot<T>& pot, T* wq, T* wv, T* wa, int64_t ld, T m,
                                           T h, int L) {
    const T h2 = h * h, hh = T(0.5) * h, hh2 = T(0.5) * h2;
    for (int s = 0; s < L; ++s) {
        if constexpr (METHOD == PBBI_LEAPFROG) leapfrog_sweep<T, K, UNIT>(pot, wq, wv, wa, ld, m, h, hh, hh2);
        else verlet_sweep<T, K, UNIT>(pot, wq, wa, ld, m, h2);
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
    int L, flags, rng;
    uint64_t seed, iter, chain0;
    double kT;
    T *Wq, *Wv, *Wa;  // (D, N) workspaces
};

template <typename T, int K, int METHOD, bool UNIT>
__global__ void __launch_bounds__(SB) k_stream_hmc(HmcPrm<T> prm, SPot<T> pot) {
    const int64_t n = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (n >= prm.N) return;
    const int D = pot.D;
    const int64_t ld = prm.N;
    const T m = UNIT ? T(1) : prm.mass[n];
    const uint64_t chain = prm.chain0 + (uint64_t)n;
    const double pstd = prm.rng ? sqrt((double)m * prm.kT) : 1.0;  // src/ensemble.py:88
    T *wq = prm.Wq + n, *wv = prm.Wv + n, *wa = prm.Wa + n;
    const T* qin = prm.q_in + n;
    const T* pin = prm.rng ? nullptr : prm.p_in + n;

    // momentum: v = p/m into the workspace, dot(p, p) in dimension order
    T pp = T(0), u;
    if (prm.rng) {
        draw_rows<T>(prm.seed, prm.iter, chain, D, pstd, (prm.flags & PBBI_DRAW_F64) != 0, [&](int d, T p) {
            pp += p * p;
            wv[(int64_t)d * ld] = UNIT ? p : p / m;
        });
        u = (T)rng_uniform(prm.seed, prm.iter, chain);
    } else {
        for (int j0 = 0; j0 < D; j0 += CH) {
            T p[CH];
            load_rows<T, CH>(pin, prm.ldn_in, j0, D, p);
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (j0 + k < D) {
                    pp += p[k] * p[k];
                    wv[(int64_t)(j0 + k) * ld] = UNIT ? p[k] : p[k] / m;
                }
        }
        u = prm.u_in[n];
    }
    const T U0 = init_sweep<T, K, METHOD, UNIT>(pot, qin, prm.ldn_in, wq, wv, wa, ld, m, prm.h);
    const T oldH = T(0.5) * pp / m + U0;
    trajectory<T, K, METHOD, UNIT>(pot, wq, wv, wa, ld, m, prm.h, prm.L);
    const T newH = final_energy<T, K, METHOD, UNIT>(pot, wq, wv, wa, ld, m, prm.h);
    const T ratio = exp((oldH - newH) * (T)pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    // mask = u > min(1, ratio); NaN ratio compares False => accepted (src/HMC.py:168-173)
    const bool reject = (ratio == ratio) && (u > (ratio < T(1) ? ratio : T(1)));

    T* qo = prm.q_out + n;
    T* po = prm.p_out ? prm.p_out + n : nullptr;
    const bool compat = (prm.flags & PBBI_COMPAT_P_FROM_OLDQ) != 0;
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH], v[CH], qi[CH];
        load_rows<T, CH>(wq, ld, j0, D, q);
        load_rows<T, CH>(METHOD == PBBI_STORMER_VERLET ? wa : wv, ld, j0, D, v);
        load_rows<T, CH>(qin, prm.ldn_in, j0, D, qi);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                qo[(int64_t)j * prm.ldn_out] = reject ? qi[k] : q[k];  // src/HMC.py:175
                if (po) {
                    const T vj = final_v<T, METHOD>(q[k], v[k], v[k], prm.h);
                    T pj = UNIT ? vj : vj * m;
                    if (reject) pj = compat ? qi[k] : (pin ? pin[(int64_t)j * prm.ldn_in] : T(0));
                    po[(int64_t)j * prm.ldn_out] = pj;  // :176 (compat: p <- oldQ)
                }
            }
        }
    }
    if (po && reject && !compat && prm.rng)  // the proposal's momentum is the draw of this iteration
        draw_rows<T>(prm.seed, prm.iter, chain, D, pstd, (prm.flags & PBBI_DRAW_F64) != 0,
                     [&](int d, T p) { po[(int64_t)d * prm.ldn_out] = p; });
    if (prm.ratio_out) prm.ratio_out[n] = ratio;
    if (prm.reject_out) prm.reject_out[n] = reject ? 1 : 0;
}

template <typename T>
struct IntPrm {
    T* q;
    T* p;
    const T* mass;
    T* v_out;
    int64_t N, ldn;
    T h;
    int L;
    T *Wq, *Wv, *Wa;
};

template <typename T, int K, int METHOD, bool UNIT>
__global__ void __launch_bounds__(SB) k_stream_integrate(IntPrm<T> prm, SPot<T> pot) {
    const int64_t n = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (n >= prm.N) return;
    const int D = pot.D;
    const int64_t ld = prm.N;
    const T m = UNIT ? T(1) : prm.mass[n];
    T *wq = prm.Wq + n, *wv = prm.Wv + n, *wa = prm.Wa + n;
    T *q = prm.q + n, *p = prm.p + n;
    T* vo = prm.v_out ? prm.v_out + n : nullptr;
    for (int j0 = 0; j0 < D; j0 += CH) {
        T x[CH];
        load_rows<T, CH>(p, prm.ldn, j0, D, x);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (j0 + k < D) wv[(int64_t)(j0 + k) * ld] = UNIT ? x[k] : x[k] / m;
    }
    (void)init_sweep<T, K, METHOD, UNIT>(pot, q, prm.ldn, wq, wv, wa, ld, m, prm.h);
    trajectory<T, K, METHOD, UNIT>(pot, wq, wv, wa, ld, m, prm.h, prm.L);
    for (int j0 = 0; j0 < D; j0 += CH) {
        T x[CH], v[CH];
        load_rows<T, CH>(wq, ld, j0, D, x);
        load_rows<T, CH>(METHOD == PBBI_STORMER_VERLET ? wa : wv, ld, j0, D, v);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                const T vj = final_v<T, METHOD>(x[k], v[k], v[k], prm.h);
                q[(int64_t)j * prm.ldn] = x[k];
                p[(int64_t)j * prm.ldn] = UNIT ? vj : vj * m;
                if (vo) vo[(int64_t)j * prm.ldn] = vj;
            }
        }
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
    int mode;  // 0: eval (U, grad); 1: energy H / w; 2: ratio finish U_out = exp(U_out - H)
};

template <typename T, int K>
__global__ void __launch_bounds__(SB) k_stream_eval(EvalPrm<T> prm, SPot<T> pot) {
    const int64_t n = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (n >= prm.N) return;
    const int D = pot.D;
    const T* qc = prm.q + n;
    const T* pc = prm.mode ? prm.p + n : nullptr;
    T* gc = (prm.mode == 0 && prm.grad_out) ? prm.grad_out + n : nullptr;
    UAcc<T, K> u;
    T carry = T(0), pp = T(0);
    for (int j0 = 0; j0 < D; j0 += CH) {
        T q[CH + 1], p[CH];
        load_rows<T, CH + 1>(qc, prm.ldn, j0, D, q);
        if (pc) load_rows<T, CH>(pc, prm.ldn, j0, D, p);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int j = j0 + k;
            if (j < D) {
                const T qj1 = j + 1 < D ? q[k + 1] : T(0);
                u.add(pot, j, q[k], qj1);
                if (gc) gc[(int64_t)j * prm.ldn] = grad_elem<T, K>(pot, j, q[k], qj1, carry);
                if (pc) pp += p[k] * p[k];
            }
        }
    }
    const T U = u.finish(pot);
    if (prm.mode == 0) {
        if (prm.U_out) prm.U_out[n] = U;
        return;
    }
    const T m = prm.mass ? prm.mass[n] : T(1);
    const T H = T(0.5) * pp / m + U;
    if (prm.mode == 1) {
        if (prm.U_out) prm.U_out[n] = H;
        if (prm.w_out) prm.w_out[n] = exp(-H);  // src/HMC.py:103
    } else {
        prm.U_out[n] = exp(prm.U_out[n] - H);  // src/HMC.py:115
    }
}

// ------------------------------------------------------------------- dispatch
template <typename T>
SPot<T> make_pot(const pbbi_potential* pot) {
    const double inv_s = 1.0 / pot->s;
    return SPot<T>{(const T*)pot->d_mean, (const T*)pot->d_prec, (T)pot->a, (T)pot->b, (T)inv_s,
                   (T)pot->cst, (T)((-4.0 * pot->b) * inv_s), (T)(2.0 * inv_s),
                   (T)((2.0 * pot->b) * inv_s), pot->D};
}

template <typename F>
void with_kind(int kind, F&& f) {
    if (kind == KIND_HARMONIC) f(std::integral_constant<int, K_HARM>{});
    else if (kind == KIND_GAUSS_DIAG) f(std::integral_constant<int, K_GAUSS>{});
    else f(std::integral_constant<int, K_ROS>{});
}
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


inline dim3 grid_for(int64_t N) { return dim3((unsigned)((N + SB - 1) / SB)); }

template <typename T>
int run_hmc(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    Scratch ws(a);
    const size_t slab = (size_t)pot->D * a.N * sizeof(T);
    T *Wq = (T*)ws.get(slab), *Wv = (T*)ws.get(slab), *Wa = (T*)ws.get(slab);
    if (!Wq || !Wv || !Wa) return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the chain workspace");
    HmcPrm<T> prm{(const T*)a.q_in, (const T*)a.p_in, (const T*)a.u_in, (const T*)a.mass,
                  (T*)a.q_out, (T*)a.p_out, (T*)a.ratio_out, a.reject_out,
                  a.N, a.ldn_in, a.ldn_out, (T)a.h, a.L, a.flags, a.rng,
                  a.seed, a.iter, a.chain0, a.kT, Wq, Wv, Wa};
    const SPot<T> sp = make_pot<T>(pot);
    with_kind(pot->kind, [&](auto k) {
        with_method_unit(a.method, a.mass == nullptr, [&](auto meth, auto unit) {
            hipLaunchKernelGGL((k_stream_hmc<T, decltype(k)::value, decltype(meth)::value,
                                             decltype(unit)::value>),
                               grid_for(a.N), dim3(SB), 0, a.stream, prm, sp);
        });
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

template <typename T>
int run_integrate(const IntegrateArgs& a) {
    const pbbi_potential* pot = a.pot;
    Scratch ws(a.stream);
    const size_t slab = (size_t)pot->D * a.N * sizeof(T);
    T *Wq = (T*)ws.get(slab), *Wv = (T*)ws.get(slab), *Wa = (T*)ws.get(slab);
    if (!Wq || !Wv || !Wa) return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the chain workspace");
    IntPrm<T> prm{(T*)a.q, (T*)a.p, (const T*)a.mass, (T*)a.v_out, a.N, a.ldn, (T)a.h, a.L, Wq, Wv, Wa};
    const SPot<T> sp = make_pot<T>(pot);
    with_kind(pot->kind, [&](auto k) {
        with_method_unit(a.method, a.mass == nullptr, [&](auto meth, auto unit) {
            hipLaunchKernelGGL((k_stream_integrate<T, decltype(k)::value, decltype(meth)::value,
                                                   decltype(unit)::value>),
                               grid_for(a.N), dim3(SB), 0, a.stream, prm, sp);
        });
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

template <typename T>
int run_eval(const EvalArgs& a, int mode) {
    const pbbi_potential* pot = a.pot;
    EvalPrm<T> prm{(const T*)a.q, (const T*)a.p, (const T*)a.mass, (T*)a.U_out, (T*)a.grad_out,
                   (T*)a.w_out, a.N, a.ldn, mode};
    const SPot<T> sp = make_pot<T>(pot);
    with_kind(pot->kind, [&](auto k) {
        hipLaunchKernelGGL((k_stream_eval<T, decltype(k)::value>), grid_for(a.N), dim3(SB), 0,
                           a.stream, prm, sp);
    });
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

}  // namespace

int stream_hmc_iter(const IterArgs& a) {
    if (a.N == 0) return PBBI_OK;
    return a.pot->dtype == PBBI_F64 ? run_hmc<double>(a) : run_hmc<float>(a);
}
int stream_integrate(const IntegrateArgs& a) {
    if (a.N == 0) return PBBI_OK;
    return a.pot->dtype == PBBI_F64 ? run_integrate<double>(a) : run_integrate<float>(a);
}
int stream_eval(const EvalArgs& a) {
    if (a.N == 0) return PBBI_OK;
    return a.pot->dtype == PBBI_F64 ? run_eval<double>(a, 0) : run_eval<float>(a, 0);
}
int stream_energy(const EvalArgs& a) {
    if (a.N == 0) return PBBI_OK;
    const int mode = a.ratio_finish ? 2 : 1;
    return a.pot->dtype == PBBI_F64 ? run_eval<double>(a, mode) : run_eval<float>(a, mode);
}
