// kernels_rosg.hip -- Rosenbrock, 32 < D <= 128, Leapfrog in the PBBI_KDK_FMA form: G = 4 or 8 LANES
// PER CHAIN inside one wave (the two-lane kernel of kernels_lane2.hip generalised), gfx950.
//
// lane l -> chain c = l % (64/G), part = l / (64/G), dims 16*part + j.  The nearest-neighbour term
// crosses a part boundary through in-wave shuffles (q_{16(part+1)} comes down from the next part,
// t_{16 part - 1} comes up from the previous one): no LDS round trip through a workgroup barrier,
// which is what limits the parts-in-waves kernel (kernels_rosn.hip: 11 barriers per iteration).
// The price is narrower row segments (64/G chains x 8 B per part) -- measured against
// kernels_rosn.hip the in-wave form wins at every D it covers: 0.70 vs 0.45 of the HBM roofline at
// D = 64, 0.51 vs 0.34 at D = 128 (PBBI_ROSG_MAX_D moves the switch for A/B runs).
//
// Arithmetic: kernels_lane2.hip's kdk_kick / U_pair / pp_pair (7 fp64 instructions per element-step,
// state q and the half-step velocity), energies summed over the G parts by an xor butterfly.
// q, p within 1e-12 of the oracle, accept masks equal (test_rosenbrock_multiwave_kdk covers both).
#include <cstdlib>

#include "pbbi_buf.h"
#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

constexpr int DL = 16;  // dims per lane

struct RosgPrm {
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
    double cst;
    // PBBI_PER_CHAIN_STEPS (k_rosg_hmc<..., DYN>): the chains' own step counts, uploaded or drawn (include/pbbi.h)
    const int32_t* steps_in;
    int32_t* steps_out;
};

template <int G>
__device__ __forceinline__ double part_sum(double x) {  // sum over the G lanes of a chain
    constexpr int CPW = 64 / G;
#pragma unroll
    for (int s = CPW; s < 64; s <<= 1) x += __shfl_xor(x, s, 64);
    return x;
}

// Where the iterations of a fused run put their results (pbbi_hmc_run, IterArgs::fuse_*; as Ros2Run in
// kernels_lane2.hip).
struct RosgRun {
    int S;            // iterations in this launch (1: plain pbbi_hmc_iter semantics)
    int wrap2;        // position slabs alternate between slab 0 and 1 of q_base (burn-in)
    int64_t slab0;    // index of the first iteration's position slab
    int64_t slab;     // elements per slab (D * N)
    double* q_base;   // slab 0 of the position slabs
};

// run.S > 1: the wave keeps its chains in registers for run.S consecutive iterations, and the potential
// energy of the position an iteration starts from is the one the previous iteration formed (carried in a
// register, the same value), as in k_ros2_hmc.
// DYN (PBBI_PER_CHAIN_STEPS, Leapfrog): chain c takes its own L_c <= L steps; every lane keeps executing (the
// kick's shuffles need the whole wave), a finished chain is frozen by per-lane coefficients -- drift step 0,
// kick 0 -- its last kick is its own half kick, the wave stops with its longest chain.
template <int G, bool UNIT, bool FULL, int METHOD, bool DYN = false>
__global__ void __launch_bounds__(64, 3) k_rosg_hmc(RosgPrm prm, RosgRun run) {
    static_assert(!DYN || METHOD == PBBI_LEAPFROG, "per-chain lengths: Leapfrog");
    constexpr int CPW = 64 / G;  // chains per wave
    const int lane = threadIdx.x;
    const int part = lane / CPW, c = lane % CPW;
    const int64_t n0 = (int64_t)blockIdx.x * CPW;  // block-uniform
    const int64_t left = prm.N - n0;
    const bool valid = c < left;
    const int cc = valid ? c : (int)left - 1;
    const int D = prm.D;
    const double m = UNIT ? 1.0 : prm.mass[n0 + cc];
    const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);
    const uint32_t rin = 8u * (uint32_t)prm.ldn_in, rout = 8u * (uint32_t)prm.ldn_out;
    const uint32_t vin = 8u * (uint32_t)cc + (uint32_t)(DL * part) * rin;
    const uint32_t vout = 8u * (uint32_t)cc + (uint32_t)(DL * part) * rout;
    // descriptors bounded to the array (pbbi_buf.h::buf_make_rows): rows past D read 0 / drop stores
    const __amdgpu_buffer_rsrc_t bq = buf_make_rows(prm.q_in + n0, D, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bp = buf_make_rows(prm.p_in + n0, D, prm.ldn_in, prm.N, n0, 8);
    auto exists = [&](int j) { return FULL ? true : DL * part + j < D; };
    auto has_next = [&](int j) {  // dim 16*part + j has a right neighbour
        if constexpr (FULL) return j + 1 < DL ? true : part + 1 < G;
        return DL * part + j + 1 < D;
    };

    double q[DL], v[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) q[j] = buf_load<double>(bq, vin, (uint32_t)j * rin);
    const double pstd = prm.rng ? sqrt(m * prm.kT) : 1.0;  // src/ensemble.py:88
    double U_carry = 0.0;  // U at the position in q[], from the second iteration of the launch on
#pragma nounroll
    for (int kf = 0; kf < run.S; ++kf) {
    const uint64_t iter_k = prm.iter + (uint64_t)kf;
    const int64_t s_out = run.wrap2 ? ((run.slab0 + kf) & 1) : run.slab0 + kf;
    const int64_t s_prev = run.wrap2 ? ((run.slab0 + kf - 1) & 1) : run.slab0 + kf - 1;
    const double* q_in_k = kf > 0 ? run.q_base + s_prev * run.slab : prm.q_in;
    const int64_t ld_in_k = kf > 0 ? prm.ldn_out : prm.ldn_in;
    double* q_out_k = run.S > 1 ? run.q_base + s_out * run.slab : prm.q_out;
    double* p_out_k = (prm.p_out && run.S > 1) ? prm.p_out + (int64_t)kf * run.slab : prm.p_out;
    const uint32_t rin_k = 8u * (uint32_t)ld_in_k;
    const uint32_t vin_k = 8u * (uint32_t)cc + (uint32_t)(DL * part) * rin_k;
    const __amdgpu_buffer_rsrc_t bq_k = buf_make_rows(q_in_k + n0, D, ld_in_k, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bqo = buf_make_rows(q_out_k + n0, D, prm.ldn_out, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bpo = buf_make_rows(p_out_k + n0, D, prm.ldn_out, prm.N, n0, 8);
    auto draw = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // this part's group of 16 dims: blocks (part<<2)|r
            double z[4];
            rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, iter_k, chain, (uint32_t)((part << 2) | r), (prm.flags & PBBI_DRAW_F64) != 0, z);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) v[r + 4 * sl] = exists(r + 4 * sl) ? z[sl] * pstd : 0.0;
        }
    };
    auto load_p = [&]() {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = buf_load<double>(bp, vin, (uint32_t)j * rin);
    };
    if (prm.rng) draw(); else load_p();

    const double nc1 = -prm.c1, nc2 = -prm.c2, c2a = prm.c2 * prm.a, nc3 = -prm.c3;
    // value held by the same chain's next / previous part (garbage in the last / first part: unused)
    auto from_next = [&](double x) { return __shfl_down(x, CPW, 64); };
    auto from_prev = [&](double x) { return __shfl_up(x, CPW, 64); };
    // H = 0.5 p.p / m + (sum b t^2 + sum (a - q)^2) / s, summed over the chain's parts
    auto kinetic = [&]() {
        double pp = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) pp = fma(v[j], v[j], pp);
        return 0.5 * part_sum<G>(pp) / m;
    };
    auto potential = [&]() {
        const double q_ext = from_next(q[0]);
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            const double t = fma(-q[j], q[j], qn);
            const double r = prm.a - q[j];
            const double n1 = fma(prm.b * t, t, s1), n2 = fma(r, r, s2);
            const bool hn = has_next(j);
            s1 = hn ? n1 : s1;
            s2 = hn ? n2 : s2;
        }
        return part_sum<G>(s1 + s2) * prm.inv_s;
    };
    // v_j += kk * (-g_j): kernels_lane2.hip::kdk_kick with the boundary terms from the neighbour parts
    auto kick = [&](double kk) {
        const double q_ext = from_next(q[0]);
        const double kn3 = kk * nc3;
        const double t15 = fma(-q[DL - 1], q[DL - 1], q_ext);
        const double t_prev = from_prev(has_next(DL - 1) ? t15 : 0.0);
        const double v0 = fma(kn3, t_prev, v[0]);
        v[0] = (part > 0 && exists(0)) ? v0 : v[0];
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
    };

    const double U_old = kf > 0 ? U_carry : potential();
    const double oldH = kinetic() + U_old;
    // ---- Leapfrog, kick-drift-kick: vh = v + a0 h/2;  L x { q += vh h; vh += a(q) h }, last kick half
    const double h = prm.h, hm = UNIT ? h : h / m, hhm = 0.5 * hm;
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] / m;
    }
    if constexpr (DYN) {
        int Ln = prm.rng ? rng_steps(prm.seed, iter_k, chain, prm.L) : (prm.steps_in ? prm.steps_in[n0 + cc] : prm.L);
        Ln = Ln < 0 ? 0 : (Ln > prm.L ? prm.L : Ln);
        kick(Ln > 0 ? hhm : 0.0);
        for (int s = 0; s < prm.L; ++s) {
            const bool act = s < Ln;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;  // wave-uniform
            const double hq = act ? h : 0.0;
#pragma unroll
            for (int j = 0; j < DL; ++j) q[j] = fma(v[j], hq, q[j]);
            kick(act ? (s + 1 == Ln ? hhm : hm) : 0.0);
        }
        if (valid && part == 0 && prm.steps_out) prm.steps_out[(int64_t)kf * prm.N + n0 + c] = Ln;
    } else if constexpr (METHOD == PBBI_LEAPFROG) {
        if (prm.L > 0) {
            kick(hhm);
            for (int s = 0; s < prm.L; ++s) {
#pragma unroll
                for (int j = 0; j < DL; ++j) q[j] = fma(v[j], h, q[j]);
                kick((s + 1 < prm.L) ? hm : hhm);
            }
        }
    } else {
        // Stormer-Verlet (src/integrator.py:142-163) with d = q_n - q_{n-1} = vh h: the same recurrence
        // without the closing half kick and with one more drift; the final velocity is vh
        kick(hhm);
        for (int s = 0; s < prm.L; ++s) {
#pragma unroll
            for (int j = 0; j < DL; ++j) q[j] = fma(v[j], h, q[j]);
            kick(hm);
        }
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] = fma(v[j], h, q[j]);
    }
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] * m;  // p = v*m
    }
    const double U_new = potential();
    const double newH = kinetic() + U_new;
    const double ratio = exp((oldH - newH) * pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const double u = prm.rng ? rng_uniform(prm.seed, iter_k, chain) : prm.u_in[n0 + cc];
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    U_carry = reject ? U_old : U_new;  // (all lanes of a chain decide alike)
    if (reject) {
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] = buf_load<double>(bq_k, vin_k, (uint32_t)j * rin_k);  // :175
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
        for (int j = 0; j < DL; ++j) buf_store(bqo, vout, (uint32_t)j * rout, q[j]);
        if (prm.p_out) {
#pragma unroll
            for (int j = 0; j < DL; ++j) buf_store(bpo, vout, (uint32_t)j * rout, v[j]);
        }
        if (part == 0) {
            if (prm.ratio_out) prm.ratio_out[(int64_t)kf * prm.N + n0 + c] = ratio;
            if (prm.reject_out) prm.reject_out[(int64_t)kf * prm.N + n0 + c] = reject ? 1 : 0;
        }
    }
    }  // kf
}

// ------------------------------------------------------------------------------------------------
// k_rosg_exact_hmc: the same G-lanes-per-chain layout in the REFERENCE'S OPERATION ORDER (Leapfrog,
// src/integrator.py:105-120): kernels_lane2.hip's reference-order form generalised from 2 to G = 4 / 8
// parts.  State per lane q, v, a of 16 dims; the gradient's nearest-neighbour terms cross the part
// boundaries by shuffles (q of the next part's first dim comes down, the carried c3*t of the previous
// part's last dim comes up); the energy sums run over the dimensions IN ORDER: the terms are formed
// once, then G short passes pass the running sums from part to part (part k's lanes are final after
// pass k).  q, p, the ratio's decision: bit-exact with the oracle.  Serves the drop-in's default
// (PBBI_KDK_FMA not set) for 32 < D <= 128, where the workspace kernel ran at 0.10 / 0.04 of the HBM
// roofline (D = 64 / 128).
// ------------------------------------------------------------------------------------------------
template <int G, bool UNIT, bool FULL>
__global__ void __launch_bounds__(64, 2) k_rosg_exact_hmc(RosgPrm prm) {
    constexpr int CPW = 64 / G;  // chains per wave
    const int lane = threadIdx.x;
    const int part = lane / CPW, c = lane % CPW;
    const int64_t n0 = (int64_t)blockIdx.x * CPW;  // block-uniform
    const int64_t left = prm.N - n0;
    const bool valid = c < left;
    const int cc = valid ? c : (int)left - 1;
    const int D = prm.D;
    const double m = UNIT ? 1.0 : prm.mass[n0 + cc];
    const uint64_t chain = prm.chain0 + (uint64_t)(n0 + cc);
    const uint32_t rin = 8u * (uint32_t)prm.ldn_in, rout = 8u * (uint32_t)prm.ldn_out;
    const uint32_t vin = 8u * (uint32_t)cc + (uint32_t)(DL * part) * rin;
    const uint32_t vout = 8u * (uint32_t)cc + (uint32_t)(DL * part) * rout;
    const __amdgpu_buffer_rsrc_t bq = buf_make_rows(prm.q_in + n0, D, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bp = buf_make_rows(prm.p_in + n0, D, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bqo = buf_make_rows(prm.q_out + n0, D, prm.ldn_out, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bpo = buf_make_rows(prm.p_out + n0, D, prm.ldn_out, prm.N, n0, 8);
    auto exists = [&](int j) { return FULL ? true : DL * part + j < D; };
    auto has_next = [&](int j) {  // dim 16*part + j has a right neighbour
        if constexpr (FULL) return j + 1 < DL ? true : part + 1 < G;
        return DL * part + j + 1 < D;
    };
    auto from_next = [&](double x) { return __shfl_down(x, CPW, 64); };
    auto from_prev = [&](double x) { return __shfl_up(x, CPW, 64); };

    double q[DL], v[DL], a[DL];  // v holds p, then the velocity, then p again
#pragma unroll
    for (int j = 0; j < DL; ++j) q[j] = buf_load<double>(bq, vin, (uint32_t)j * rin);
    const double pstd = prm.rng ? sqrt(m * prm.kT) : 1.0;  // src/ensemble.py:88
    auto draw = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double z[4];
            rng_normal4d(prm.seed, PBBI_STREAM_MOMENTUM, prm.iter, chain, (uint32_t)((part << 2) | r), (prm.flags & PBBI_DRAW_F64) != 0, z);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) v[r + 4 * sl] = exists(r + 4 * sl) ? z[sl] * pstd : 0.0;
        }
    };
    auto load_p = [&]() {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = buf_load<double>(bp, vin, (uint32_t)j * rin);
    };
    if (prm.rng) draw(); else load_p();

    // H = 0.5 p.p / m + ((sum b t^2 + sum (a - q)^2) / s + cst), every sum in dimension order
    auto hamiltonian = [&]() {
        const double q_ext = from_next(q[0]);
        double t[DL], bt[DL], r[DL], p2[DL];
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            t[j] = fma(-q[j], q[j], qn);
            bt[j] = prm.b * t[j];
            r[j] = prm.a - q[j];
            p2[j] = v[j] * v[j];
        }
        double s1 = 0.0, s2 = 0.0, pp = 0.0;
#pragma unroll
        for (int k = 0; k < G; ++k) {  // pass k: part k continues part k-1's sums and becomes final
            double r1 = k ? from_prev(s1) : 0.0, r2 = k ? from_prev(s2) : 0.0, r3 = k ? from_prev(pp) : 0.0;
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
#pragma unroll
            for (int j = 0; j < DL; ++j) r3 += p2[j];
            if (k == 0 || part >= k) { s1 = r1; s2 = r2; pp = r3; }
        }
        // the last part holds the totals: every lane of the chain takes them
        const int src = (G - 1) * CPW + c;
        s1 = __shfl(s1, src, 64); s2 = __shfl(s2, src, 64); pp = __shfl(pp, src, 64);
        return 0.5 * pp / m + ((s1 + s2) * prm.inv_s + prm.cst);
    };
    // visit(j, -g_j) in the oracle's operation order (kernels_lane2.hip::neg_grad_each, G parts)
    const double nc1 = -prm.c1, nc3 = -prm.c3;
    auto neg_grad_each = [&](auto&& visit) {
        const double q_ext = from_next(q[0]);
        const double t15 = fma(-q[DL - 1], q[DL - 1], q_ext);
        const double nsec15 = has_next(DL - 1) ? nc3 * t15 : 0.0;
        const double carry_ext = from_prev(nsec15);
        double carry = part > 0 ? carry_ext : 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double qn = (j + 1 < DL) ? q[(j + 1) & (DL - 1)] : q_ext;
            const double tj = fma(-q[j], q[j], qn);
            const double nfirst = fma(nc1 * q[j], tj, prm.c2 * (prm.a - q[j]));
            const bool hn = has_next(j);
            const double ngj = hn ? carry + nfirst : carry;
            carry = hn ? nc3 * tj : 0.0;
            visit(j, exists(j) ? ngj : 0.0);
        }
    };

    const double oldH = hamiltonian();
    // ---- Leapfrog.integrate, src/integrator.py:105-120 (operation order of kernels_lane.hip::integrate_chain)
    const double h = prm.h, hh2 = 0.5 * (prm.h * prm.h), hh = 0.5 * prm.h;
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] / m;
    }
    neg_grad_each([&](int j, double ng) { a[j] = UNIT ? ng : ng / m; });
    for (int s = 0; s < prm.L; ++s) {
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] += (v[j] * h + a[j] * hh2);
        neg_grad_each([&](int j, double ng) {
            const double an = UNIT ? ng : ng / m;
            v[j] += (a[j] + an) * hh;
            a[j] = an;
        });
    }
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] * m;  // p = v*m
    }
    const double newH = hamiltonian();
    const double ratio = exp((oldH - newH) * pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const double u = prm.rng ? rng_uniform(prm.seed, prm.iter, chain) : prm.u_in[n0 + cc];
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    if (reject) {
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] = buf_load<double>(bq, vin, (uint32_t)j * rin);  // :175
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
        for (int j = 0; j < DL; ++j) buf_store(bqo, vout, (uint32_t)j * rout, q[j]);
        if (prm.p_out) {
#pragma unroll
            for (int j = 0; j < DL; ++j) buf_store(bpo, vout, (uint32_t)j * rout, v[j]);
        }
        if (part == 0) {
            if (prm.ratio_out) prm.ratio_out[n0 + c] = ratio;
            if (prm.reject_out) prm.reject_out[n0 + c] = reject ? 1 : 0;
        }
    }
}

template <int G>
void launch_exact(const IterArgs& a, const RosgPrm& prm, bool full) {
    constexpr int CPW = 64 / G;
    const dim3 grid((unsigned)((a.N + CPW - 1) / CPW)), block(64);
    if (a.mass) {
        if (full) hipLaunchKernelGGL((k_rosg_exact_hmc<G, false, true>), grid, block, 0, a.stream, prm);
        else hipLaunchKernelGGL((k_rosg_exact_hmc<G, false, false>), grid, block, 0, a.stream, prm);
    } else {
        if (full) hipLaunchKernelGGL((k_rosg_exact_hmc<G, true, true>), grid, block, 0, a.stream, prm);
        else hipLaunchKernelGGL((k_rosg_exact_hmc<G, true, false>), grid, block, 0, a.stream, prm);
    }
}

template <int G>
void launch(const IterArgs& a, const RosgPrm& prm, bool full) {
    constexpr int CPW = 64 / G;
    const dim3 grid((unsigned)((a.N + CPW - 1) / CPW)), block(64);
    RosgRun run{1, 0, 0, (int64_t)a.pot->D * a.N, (double*)a.q_out};
    if (a.fuse_S > 1) run = RosgRun{a.fuse_S, a.fuse_wrap2, a.fuse_slab0, (int64_t)a.pot->D * a.N, (double*)a.fuse_q_base};
#define ROSG_LAUNCH(U_, M_)                                                                          \
    {                                                                                                \
        if (full) hipLaunchKernelGGL((k_rosg_hmc<G, U_, true, M_>), grid, block, 0, a.stream, prm, run);  \
        else hipLaunchKernelGGL((k_rosg_hmc<G, U_, false, M_>), grid, block, 0, a.stream, prm, run);      \
    }
    if (pbbi_dyn(a)) {   // (lane_hmc_iter sends PBBI_PER_CHAIN_STEPS without PBBI_UTURN_STOP only)
        if (a.mass) {
            if (full) hipLaunchKernelGGL((k_rosg_hmc<G, false, true, PBBI_LEAPFROG, true>), grid, block, 0, a.stream, prm, run);
            else hipLaunchKernelGGL((k_rosg_hmc<G, false, false, PBBI_LEAPFROG, true>), grid, block, 0, a.stream, prm, run);
        } else {
            if (full) hipLaunchKernelGGL((k_rosg_hmc<G, true, true, PBBI_LEAPFROG, true>), grid, block, 0, a.stream, prm, run);
            else hipLaunchKernelGGL((k_rosg_hmc<G, true, false, PBBI_LEAPFROG, true>), grid, block, 0, a.stream, prm, run);
        }
    } else if (a.method == PBBI_LEAPFROG) {
        if (a.mass) ROSG_LAUNCH(false, PBBI_LEAPFROG) else ROSG_LAUNCH(true, PBBI_LEAPFROG)
    } else {
        if (a.mass) ROSG_LAUNCH(false, PBBI_STORMER_VERLET) else ROSG_LAUNCH(true, PBBI_STORMER_VERLET)
    }
#undef ROSG_LAUNCH
}

}  // namespace

// true if this path takes the call: Rosenbrock, fp64, Leapfrog, PBBI_KDK_FMA, 32 < D <= PBBI_ROSG_MAX_D
bool rosg_applies(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    static const int max_d = getenv("PBBI_ROSG_MAX_D") ? atoi(getenv("PBBI_ROSG_MAX_D")) : 128;
    static const int min_d = getenv("PBBI_ROSG_MIN_D") ? atoi(getenv("PBBI_ROSG_MIN_D")) : 33;
    // Leapfrog at D <= 32 belongs to kernels_lane2.hip; Stormer-Verlet has no other kick-drift-kick
    // kernel, so it comes here from D = 17 on (two lanes per chain)
    const int lo = a.method == PBBI_LEAPFROG ? min_d : 17;
    return pot->kind == KIND_ROSENBROCK && pot->dtype == PBBI_F64 &&
           (a.flags & PBBI_KDK_FMA) != 0 && pot->D >= lo && pot->D > 16 && pot->D <= max_d && pot->D <= 128 &&
           (int64_t)pot->D * (a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) < ((int64_t)1 << 28);
}

int rosg_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (a.N == 0) return PBBI_OK;
    const double inv_s = 1.0 / pot->s;
    RosgPrm prm{(const double*)a.q_in, (const double*)a.p_in, (const double*)a.u_in,
                (const double*)a.mass, (double*)a.q_out, (double*)a.p_out, (double*)a.ratio_out,
                a.reject_out, a.N, a.ldn_in, a.ldn_out, a.h, pot->a, pot->b, inv_s, a.kT,
                (-4.0 * pot->b) * inv_s, 2.0 * inv_s, (2.0 * pot->b) * inv_s, a.L, pot->D, a.flags,
                a.rng, a.seed, a.iter, a.chain0, pot->cst, a.steps_in, a.steps_out};
    if (pot->D <= 32) launch<2>(a, prm, pot->D == 32);
    else if (pot->D <= 64) launch<4>(a, prm, pot->D == 64);
    else launch<8>(a, prm, pot->D == 128);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// the reference-order form: Rosenbrock, fp64, Leapfrog, 32 < D <= 128, PBBI_KDK_FMA not set
bool rosgx_applies(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    static const bool off = (getenv("PBBI_NO_ROSGX") != nullptr);  // A/B switch
    return !off && pot->kind == KIND_ROSENBROCK && pot->dtype == PBBI_F64 && a.method == PBBI_LEAPFROG &&
           (a.flags & PBBI_KDK_FMA) == 0 && pot->D > 32 && pot->D <= 128 &&
           (int64_t)pot->D * (a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) < ((int64_t)1 << 28);
}

int rosgx_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (a.N == 0) return PBBI_OK;
    const double inv_s = 1.0 / pot->s;
    RosgPrm prm{(const double*)a.q_in, (const double*)a.p_in, (const double*)a.u_in,
                (const double*)a.mass, (double*)a.q_out, (double*)a.p_out, (double*)a.ratio_out,
                a.reject_out, a.N, a.ldn_in, a.ldn_out, a.h, pot->a, pot->b, inv_s, a.kT,
                (-4.0 * pot->b) * inv_s, 2.0 * inv_s, (2.0 * pot->b) * inv_s, a.L, pot->D, a.flags,
                a.rng, a.seed, a.iter, a.chain0, pot->cst};
    if (pot->D <= 64) launch_exact<4>(a, prm, pot->D == 64);
    else launch_exact<8>(a, prm, pot->D == 128);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}
