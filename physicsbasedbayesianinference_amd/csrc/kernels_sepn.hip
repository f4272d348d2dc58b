// kernels_sepn.hip -- harmonic / diagonal-Gaussian potentials, 16 < D <= 256, Leapfrog in the
// PBBI_KDK_FMA form (Stormer-Verlet likewise, as the same recurrence): a chain's dimensions are cut into 16-dim PARTS held by different WAVES of one
// workgroup, gfx950.
//
// With one chain per lane (kernels_lane.hip) a D = 64 chain needs 192 VGPRs of state and runs one
// wave per SIMD at 24 % of the HBM roofline; D = 128 does not fit at all.  A separable potential has
// no coupling between dimensions: workgroup b owns chains 64b .. 64b+63, its wave w the dims
// 16w .. 16w+15 of all of them.  The part index is wave-uniform, so the part's means and precisions
// sit in SGPRs (scalar loads), the per-lane state is x = q - mu and the velocity only (64 VGPRs),
// and every load / store is a full 512-byte row segment.  The parts meet once per iteration: each
// wave's share of oldH - newH goes through LDS, one __syncthreads, and every wave takes the same
// accept decision.
//
// Kick-drift-kick with fused multiply-adds (include/pbbi.h, PBBI_KDK_FMA): per element-step
//     x = fma(v, h, x);  v = fma(prec*x, -h/m, v)     (x = q - mu)      -- 3 fp64 instructions,
// so the kernel is bound by HBM (4*D*8 B per chain per iteration), not by the vector ALU.
// Algebraically src/integrator.py:105-120; agrees with the oracle to ~1e-13 relative, accept masks
// equal (tests/test_gpu_parity.py::test_separable_multilane_kdk).  The bit-exact reference-order
// kernels stay the default; this path is taken only when the caller passes PBBI_KDK_FMA.
#include <cstdlib>

#include "pbbi_buf.h"
#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

constexpr int DL = 16;      // dims per wave
constexpr int MAXG = 16;    // waves per workgroup (D <= 256)

struct SepPrm {
    const double* q_in;
    const double* p_in;
    const double* u_in;
    const double* mass;
    double* q_out;
    double* p_out;
    double* ratio_out;
    uint8_t* reject_out;
    const double* mean;  // D
    const double* prec;  // D (spring constants for the harmonic potential)
    int64_t N, ldn_in, ldn_out;
    double h, cst, kT;
    int L, D, flags, rng;
    uint64_t seed, iter, chain0;
    int harmonic;  // k_sep_exact_hmc: U = 0.5 sum k q^2 in the harmonic potential's own operation order
    // PBBI_PER_CHAIN_STEPS (k_sep_hmc<..., DYN>): the chains' own step counts, uploaded or drawn (include/pbbi.h)
    const int32_t* steps_in;
    int32_t* steps_out;
};

// Where the iterations of a fused run put their results (pbbi_hmc_run, IterArgs::fuse_*; as Ros2Run in
// kernels_lane2.hip): iteration k of the launch writes position slab (slab0 + k), modulo 2 for a burn-in's
// two scratch slabs, momentum slab k, ratio / reject rows k.
struct SepRun {
    int S;            // iterations in this launch (1: plain pbbi_hmc_iter semantics)
    int wrap2;
    int64_t slab0;    // index of the first iteration's position slab
    int64_t slab;     // elements per slab (D * N)
    double* q_base;   // slab 0 of the position slabs
};

// FULL: D is a multiple of 16, every dim of every part exists: no guards (as scalar branches they
// put an s_waitcnt between consecutive loads / stores)
// run.S > 1: the workgroup keeps its 64 chains in registers for run.S consecutive iterations -- the
// kernel is bound by HBM, and a chain that is not re-read every iteration moves 3 instead of 4 slabs.
// Every iteration ends with the position it stored (x + mu, or the old position of a rejected chain) and
// starts the next one from that value minus mu, exactly what a launch of its own would load and form.
// DYN (PBBI_PER_CHAIN_STEPS, Leapfrog): chain c takes its own L_c <= L steps.  No lane leaves the loop -- a
// finished chain is frozen by per-lane coefficients (drift step 0, kick 0), its last kick is its own half kick
// -- and the wave stops when its longest chain has; every wave of the workgroup holds the same 64 chains, so
// they all run the same count.
template <bool UNIT, bool FULL, int METHOD, bool DYN = false>
__global__ void __launch_bounds__(64 * MAXG) k_sep_hmc(SepPrm prm, SepRun run) {
    static_assert(!DYN || METHOD == PBBI_LEAPFROG, "per-chain lengths: Leapfrog");
    __shared__ double dH[2][MAXG][64];  // by iteration parity: one barrier per iteration is enough
    const int c = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform
    const int G = (int)(blockDim.x >> 6);
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
    auto exists = [&](int j) { return FULL || d0 + j < D; };  // wave-uniform
    auto ld = [&](__amdgpu_buffer_rsrc_t r, int j) { return buf_load<double>(r, vo, (uint32_t)j * rin); };
    auto st = [&](__amdgpu_buffer_rsrc_t r, int j, double x) { buf_store(r, vo, (uint32_t)j * rout, x); };

    // this part's constants (SGPRs); a dim past D gets prec = 0 and stays at x = v = 0
    double mu[DL], pr[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        mu[j] = prm.mean[d0 + j];  // parameter vectors are zero-padded to a multiple of 64
        pr[j] = prm.prec[d0 + j];
    }
    const double h = prm.h, nhm = UNIT ? -h : -(h / m), nhh = 0.5 * nhm;
    double q[DL], v[DL];  // q holds x = q - mu between the load and the store
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        const double x = ld(bq, j) - mu[j];
        q[j] = exists(j) ? x : 0.0;
    }
    const double pstd = prm.rng ? sqrt(m * prm.kT) : 1.0;  // src/ensemble.py:88
#pragma nounroll
    for (int kf = 0; kf < run.S; ++kf) {
    const uint64_t iter_k = prm.iter + (uint64_t)kf;
    // this iteration's view: where a rejected chain re-reads its position, where the results go
    const int64_t s_out = run.wrap2 ? ((run.slab0 + kf) & 1) : run.slab0 + kf;
    const int64_t s_prev = run.wrap2 ? ((run.slab0 + kf - 1) & 1) : run.slab0 + kf - 1;
    const double* q_in_k = kf > 0 ? run.q_base + s_prev * run.slab : prm.q_in;
    const int64_t ld_in_k = kf > 0 ? prm.ldn_out : prm.ldn_in;
    double* q_out_k = run.S > 1 ? run.q_base + s_out * run.slab : prm.q_out;
    double* p_out_k = (prm.p_out && run.S > 1) ? prm.p_out + (int64_t)kf * run.slab : prm.p_out;
    const uint32_t rin_k = 8u * (uint32_t)ld_in_k;
    const __amdgpu_buffer_rsrc_t bq_k = buf_make_rows(q_in_k + n0 + (int64_t)d0 * ld_in_k, D - d0, ld_in_k, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bqo = buf_make_rows(q_out_k + n0 + (int64_t)d0 * prm.ldn_out, D - d0, prm.ldn_out, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bpo = buf_make_rows(p_out_k + n0 + (int64_t)d0 * prm.ldn_out, D - d0, prm.ldn_out, prm.N, n0, 8);
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
        for (int j = 0; j < DL; ++j) {
            const double x = ld(bp, j);
            v[j] = exists(j) ? x : 0.0;
        }
    };
    if (prm.rng) draw(); else load_p();

    // this part's share of H = 0.5 p.p / m + 0.5 sum prec x^2 (+ cst, which cancels in oldH - newH)
    auto energy = [&]() {
        double pp = 0.0, xx = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            pp = fma(v[j], v[j], pp);
            xx = fma(pr[j] * q[j], q[j], xx);
        }
        return 0.5 * pp / m + 0.5 * xx;
    };
    const double oldE = energy();

    // ---- Leapfrog, kick-drift-kick: vh = v + a0 h/2;  L x { x += vh h; vh += a(x) h }, last kick half
    // a = -prec x / m:  vh = fma(prec x, -h/m, vh)
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] / m;
    }
    if constexpr (DYN) {
        int Ln = prm.rng ? rng_steps(prm.seed, iter_k, chain, prm.L) : (prm.steps_in ? prm.steps_in[n0 + cc] : prm.L);
        Ln = Ln < 0 ? 0 : (Ln > prm.L ? prm.L : Ln);
        const double c0 = Ln > 0 ? nhh : 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = fma(pr[j] * q[j], c0, v[j]);
        for (int s = 0; s < prm.L; ++s) {
            const bool act = s < Ln;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;  // wave-uniform
            const double hq = act ? h : 0.0, ck = act ? (s + 1 == Ln ? nhh : nhm) : 0.0;
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                q[j] = fma(v[j], hq, q[j]);
                v[j] = fma(pr[j] * q[j], ck, v[j]);
            }
        }
        if (valid && part == 0 && prm.steps_out) prm.steps_out[(int64_t)kf * prm.N + n0 + c] = Ln;
    } else if constexpr (METHOD == PBBI_LEAPFROG) {
        if (prm.L > 0) {
#pragma unroll
            for (int j = 0; j < DL; ++j) v[j] = fma(pr[j] * q[j], nhh, v[j]);
            for (int s = 0; s + 1 < prm.L; ++s) {
#pragma unroll
                for (int j = 0; j < DL; ++j) {
                    q[j] = fma(v[j], h, q[j]);
                    v[j] = fma(pr[j] * q[j], nhm, v[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                q[j] = fma(v[j], h, q[j]);
                v[j] = fma(pr[j] * q[j], nhh, v[j]);
            }
        }
    } else {
        // Stormer-Verlet (src/integrator.py:142-163) with d = q_n - q_{n-1} = vh h: the same recurrence
        // without the closing half kick and with one more drift; the final velocity is vh
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = fma(pr[j] * q[j], nhh, v[j]);
        for (int s = 0; s < prm.L; ++s) {
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                q[j] = fma(v[j], h, q[j]);
                v[j] = fma(pr[j] * q[j], nhm, v[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] = fma(v[j], h, q[j]);
    }
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] * m;  // p = v*m
    }

    // oldH - newH: the parts' shares summed in part order (the same value in every wave)
    dH[kf & 1][part][c] = oldE - energy();
    __syncthreads();
    double dsum = 0.0;
    for (int g = 0; g < G; ++g) dsum += dH[kf & 1][g][c];
    const double ratio = exp(dsum * pbbi_accept_beta(prm.flags, prm.kT));  // src/HMC.py:115
    const double u = prm.rng ? rng_uniform(prm.seed, iter_k, chain) : prm.u_in[n0 + cc];
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    // back to positions: q = x + mu, or the untouched old position for a rejected chain (:175)
#pragma unroll
    for (int j = 0; j < DL; ++j) q[j] = q[j] + mu[j];
    if (reject) {
#pragma unroll
        for (int j = 0; j < DL; ++j) q[j] = buf_load<double>(bq_k, vo, (uint32_t)j * rin_k);
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
            if (prm.ratio_out) prm.ratio_out[(int64_t)kf * prm.N + n0 + c] = ratio;
            if (prm.reject_out) prm.reject_out[(int64_t)kf * prm.N + n0 + c] = reject ? 1 : 0;
        }
    }
    if (kf + 1 < run.S) {  // the next iteration's x, from the position just stored (what a launch would load)
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double x = q[j] - mu[j];
            q[j] = exists(j) ? x : 0.0;
        }
    }
    }  // kf
}

// ------------------------------------------------------------------------------------------------
// k_sep_exact_hmc: the same parts-in-waves layout in the REFERENCE'S OPERATION ORDER (Leapfrog,
// src/integrator.py:105-120, and Stormer-Verlet, :142-163, exactly as kernels_lane.hip::integrate_chain) -- the default of the drop-in
// (PBBI_KDK_FMA not set).  State per lane q, v, a (96 VGPRs per 16 dims, four waves per SIMD); the
// gradient of a separable potential is elementwise, so the trajectory needs nothing from the other
// parts.  What ties the parts together is the ORDER of the two energy sums (p.p and the potential's
// terms run over d = 0 .. D-1 in the oracle): part g continues part g-1's running sums through LDS, G
// short dependent steps per Hamiltonian, and every wave reads the totals -> H, the ratio and the
// decision have the oracle's bits, like q and p.  Before: one chain per lane up to D = 64 (0.44 / 0.24
// of the HBM roofline at D = 32 / 64) and the workspace kernels beyond (0.04).
// ------------------------------------------------------------------------------------------------
template <bool UNIT, bool FULL, int METHOD>
__global__ void __launch_bounds__(64 * MAXG) k_sep_exact_hmc(SepPrm prm) {
    __shared__ double run_pp[64], run_u[64];
    const int c = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform
    const int G = (int)(blockDim.x >> 6);
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
    const __amdgpu_buffer_rsrc_t bq = buf_make_rows(prm.q_in + n0 + (int64_t)d0 * prm.ldn_in, D - d0, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bp = buf_make_rows(prm.p_in + n0 + (int64_t)d0 * prm.ldn_in, D - d0, prm.ldn_in, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bqo = buf_make_rows(prm.q_out + n0 + (int64_t)d0 * prm.ldn_out, D - d0, prm.ldn_out, prm.N, n0, 8);
    const __amdgpu_buffer_rsrc_t bpo = buf_make_rows(prm.p_out + n0 + (int64_t)d0 * prm.ldn_out, D - d0, prm.ldn_out, prm.N, n0, 8);
    auto exists = [&](int j) { return FULL || d0 + j < D; };  // wave-uniform
    auto ld = [&](__amdgpu_buffer_rsrc_t r, int j) { return buf_load<double>(r, vo, (uint32_t)j * rin); };
    auto st = [&](__amdgpu_buffer_rsrc_t r, int j, double x) { buf_store(r, vo, (uint32_t)j * rout, x); };

    // this part's constants (SGPRs); the parameter vectors are zero-padded: a dim past D has prec = 0,
    // mean = 0 and q = v = a = 0 throughout, its terms add exact zeros
    double mu[DL], pr[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        mu[j] = prm.mean[d0 + j];
        pr[j] = prm.prec[d0 + j];
    }
    double q[DL], v[DL], a[DL];  // v holds p, then the velocity, then p again
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        const double x = ld(bq, j);
        q[j] = exists(j) ? x : 0.0;
    }
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
        for (int j = 0; j < DL; ++j) {
            const double x = ld(bp, j);
            v[j] = exists(j) ? x : 0.0;
        }
    };
    if (prm.rng) draw(); else load_p();

    // H = 0.5 p.p / m + (0.5 acc + cst), both sums in dimension order across the parts
    // (src/HMC.py:100-102; oracle pot_U): part g starts from part g-1's running sums
    auto hamiltonian = [&]() {
        for (int g = 0; g < G; ++g) {
            if (part == g) {
                double pp = g ? run_pp[c] : 0.0, acc = g ? run_u[c] : 0.0;
#pragma unroll
                for (int j = 0; j < DL; ++j) pp += v[j] * v[j];
                if (prm.harmonic) {
#pragma unroll
                    for (int j = 0; j < DL; ++j) acc += pr[j] * (q[j] * q[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < DL; ++j) {
                        const double x = q[j] - mu[j];
                        acc += (pr[j] * x) * x;
                    }
                }
                run_pp[c] = pp;
                run_u[c] = acc;
            }
            __syncthreads();
        }
        const double H = 0.5 * run_pp[c] / m + (0.5 * run_u[c] + prm.cst);
        __syncthreads();  // everyone has read the totals before the next Hamiltonian overwrites them
        return H;
    };
    const double oldH = hamiltonian();

    // ---- Leapfrog.integrate, src/integrator.py:105-120 (operation order of integrate_chain)
    const double h = prm.h, h2 = prm.h * prm.h, hh2 = 0.5 * h2, hh = 0.5 * prm.h;
    auto accel = [&](int j) {  // -gradient / mass (src/integrator.py:73)
        const double g = pr[j] * (q[j] - mu[j]);
        return UNIT ? -g : -g / m;
    };
    if constexpr (!UNIT) {
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = v[j] / m;
    }
    if constexpr (METHOD == PBBI_LEAPFROG) {
#pragma unroll
        for (int j = 0; j < DL; ++j) a[j] = accel(j);
        for (int s = 0; s < prm.L; ++s) {
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                q[j] += (v[j] * h + a[j] * hh2);
                const double an = accel(j);
                v[j] += (a[j] + an) * hh;
                a[j] = an;
            }
        }
    } else {  // Stormer-Verlet, src/integrator.py:142-163 (a[] becomes qPast after the first step)
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const double q0 = q[j];
            q[j] = (q0 + v[j] * h) + (0.5 * accel(j)) * h2;  // accel at q0: evaluated before q[j] changes
            a[j] = q0;
        }
        for (int s = 0; s < prm.L; ++s) {
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                const double cur = q[j];
                q[j] = (2.0 * cur - a[j]) + accel(j) * h2;
                a[j] = cur;
            }
        }
#pragma unroll
        for (int j = 0; j < DL; ++j) v[j] = (q[j] - a[j]) / h;
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

// true if this path takes the call: harmonic / diagonal Gaussian, fp64, Leapfrog, 16 < D <= 256,
// and the caller allowed the kick-drift-kick/FMA form
bool sepn_applies(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    return (pot->kind == KIND_HARMONIC || pot->kind == KIND_GAUSS_DIAG) && pot->dtype == PBBI_F64 &&
           (a.flags & PBBI_KDK_FMA) != 0 && pot->D > 16 && pot->D <= DL * MAXG &&
           (int64_t)DL * (a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) < ((int64_t)1 << 28);
}

int sepn_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (a.N == 0) return PBBI_OK;
    SepPrm prm{(const double*)a.q_in, (const double*)a.p_in, (const double*)a.u_in,
               (const double*)a.mass, (double*)a.q_out, (double*)a.p_out, (double*)a.ratio_out,
               a.reject_out, (const double*)pot->d_mean, (const double*)pot->d_prec, a.N, a.ldn_in,
               a.ldn_out, a.h, pot->cst, a.kT, a.L, pot->D, a.flags, a.rng, a.seed, a.iter, a.chain0,
               pot->kind == KIND_HARMONIC ? 1 : 0, a.steps_in, a.steps_out};
    const int G = (pot->D + DL - 1) / DL;
    const dim3 grid((unsigned)((a.N + 63) / 64)), block(64 * G);
    const bool full = (pot->D % DL == 0);
    const bool dyn = pbbi_dyn(a);   // (lane_hmc_iter sends PBBI_PER_CHAIN_STEPS without PBBI_UTURN_STOP only)
    SepRun run{1, 0, 0, (int64_t)pot->D * a.N, (double*)a.q_out};
    if (a.fuse_S > 1) run = SepRun{a.fuse_S, a.fuse_wrap2, a.fuse_slab0, (int64_t)pot->D * a.N, (double*)a.fuse_q_base};
#define SEP_LAUNCH(U_, F_)                                                                              \
    {                                                                                                   \
        if (dyn)                                                                                        \
            hipLaunchKernelGGL((k_sep_hmc<U_, F_, PBBI_LEAPFROG, true>), grid, block, 0, a.stream, prm, run); \
        else if (a.method == PBBI_LEAPFROG)                                                             \
            hipLaunchKernelGGL((k_sep_hmc<U_, F_, PBBI_LEAPFROG>), grid, block, 0, a.stream, prm, run); \
        else                                                                                            \
            hipLaunchKernelGGL((k_sep_hmc<U_, F_, PBBI_STORMER_VERLET>), grid, block, 0, a.stream, prm, run);\
    }
    if (a.mass) {
        if (full) SEP_LAUNCH(false, true) else SEP_LAUNCH(false, false)
    } else {
        if (full) SEP_LAUNCH(true, true) else SEP_LAUNCH(true, false)
    }
#undef SEP_LAUNCH
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// the reference-order form of the same layout: harmonic / diagonal Gaussian, fp64, both integrators,
// 16 < D <= 256, PBBI_KDK_FMA not set
bool sepx_applies(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    static const bool off = (getenv("PBBI_NO_SEPX") != nullptr);  // A/B switch
    return !off && (pot->kind == KIND_HARMONIC || pot->kind == KIND_GAUSS_DIAG) && pot->dtype == PBBI_F64 &&
           (a.flags & PBBI_KDK_FMA) == 0 && pot->D > 16 && pot->D <= DL * MAXG &&
           (int64_t)DL * (a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out) < ((int64_t)1 << 28);
}

int sepx_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (a.N == 0) return PBBI_OK;
    SepPrm prm{(const double*)a.q_in, (const double*)a.p_in, (const double*)a.u_in,
               (const double*)a.mass, (double*)a.q_out, (double*)a.p_out, (double*)a.ratio_out,
               a.reject_out, (const double*)pot->d_mean, (const double*)pot->d_prec, a.N, a.ldn_in,
               a.ldn_out, a.h, pot->cst, a.kT, a.L, pot->D, a.flags, a.rng, a.seed, a.iter, a.chain0,
               pot->kind == KIND_HARMONIC ? 1 : 0};
    const int G = (pot->D + DL - 1) / DL;
    const dim3 grid((unsigned)((a.N + 63) / 64)), block(64 * G);
    const bool full = (pot->D % DL == 0);
#define SEPX_LAUNCH(U_, F_)                                                                                  \
    {                                                                                                        \
        if (a.method == PBBI_LEAPFROG)                                                                       \
            hipLaunchKernelGGL((k_sep_exact_hmc<U_, F_, PBBI_LEAPFROG>), grid, block, 0, a.stream, prm);     \
        else                                                                                                 \
            hipLaunchKernelGGL((k_sep_exact_hmc<U_, F_, PBBI_STORMER_VERLET>), grid, block, 0, a.stream, prm); \
    }
    if (a.mass) {
        if (full) SEPX_LAUNCH(false, true) else SEPX_LAUNCH(false, false)
    } else {
        if (full) SEPX_LAUNCH(true, true) else SEPX_LAUNCH(true, false)
    }
#undef SEPX_LAUNCH
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}
