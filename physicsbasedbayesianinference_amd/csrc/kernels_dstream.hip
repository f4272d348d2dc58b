// kernels_dstream.hip -- dense-precision Gaussian potential, 128 < D <= 256, fp64, gfx950: the register-resident
// MFMA kernel of kernels_dense.hip with P STREAMED through the LDS instead of resident in it.
//
// Why: beyond D = 128 the path used to be one fused GEMM per leapfrog step over the whole ensemble
// (kernels_big.hip), which at D = 256 moves the chain state through HBM every step -- 10 KB per chain and step
// against 131 kflop: HBM time and MFMA time are equal and the kernel reaches 0.46 of the fp64 MFMA peak.  The
// state of 16 chains at D = 256 (q, vh: 2 x 64 doubles per lane) still fits ONE wave's 512 registers, so the
// trajectory can stay on chip as it does at D <= 128; what does not fit is P (512 KiB against 160 KiB of LDS).
// P is the same for every chain and every step, so it is streamed: 16 KiB chunks in the order the mat-vec
// consumes them, L2 -> LDS by LDS-DMA, RING - 1 chunks ahead of the MFMAs (kernels_dense_dev.h, StreamCfg).
// Per mat-vec a workgroup (64 chains, 4 waves, one per SIMD) reads P once from L2: 512 KiB per 27 us and CU,
// 4.9 TB/s over the chip, all of it L2 hits (P fits every XCD's L2).  HBM sees the draw-free iteration only:
// position in, position / momentum / carried gradient out.
//
// Served here: pbbi_hmc_iter / pbbi_hmc_run iterations (Leapfrog and Stormer-Verlet) with the gradient carried
// between the iterations of a run and up to 64 iterations per launch, per-chain trajectory lengths / the U-turn
// stop (Leapfrog; hence GIST), and integrate() with L >= 1.  The evaluations (potential, gradient, energies),
// L = 0 and fp32 at these D stay on kernels_big.hip, which is built for the same handle.
#include "kernels_dense_dev.h"

namespace {

inline size_t stream_lds_bytes(int DP) {   // ring | mu | the per-chain-length kernels' step votes (2 x 4 ints)
    return (DP == 256 ? (size_t)StreamCfg<16>::LDS_RING + 256 * 8 : (size_t)StreamCfg<12>::LDS_RING + 192 * 8) + 32;
}

template <typename K>
int set_lds_s(K kernel, size_t bytes) {
    PBBI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)bytes));
    return PBBI_OK;
}

bool ld_fits(const pbbi_potential* pot, int64_t ld) { return (int64_t)pot->DPS * ld < ((int64_t)1 << 29); }

}  // namespace

// P (row-major, D x D) -> the stream image: chunk ci = pass * CPP + kc holds, for its KC K-steps s8 and the HP
// fragment pairs t2 of its row pass, [s8][t2][lane][e] = P[16 * (pass * NTP + 2 * t2 + e) + (lane & 15)]
//                                                         [4 * (kc * KC + s8) + (lane >> 4)], zero padded to DPS.
int dense_stream_build(pbbi_potential* pot, const double* P, const double* mean) {
    pot->DPS = 0;
    pot->d_sfrag = nullptr;
    pot->d_smean = nullptr;
    const int D = pot->D;
    if (D <= 128 || D > 256 || pot->dtype != PBBI_F64) return PBBI_OK;
    const int DPS = D <= 192 ? 192 : 256;
    const int NT = DPS / 16, NTP = NT / 2, HP = NTP / 2, KC = 4, CPP = NT;
    std::vector<double> frag((size_t)DPS * DPS, 0.0), mu((size_t)DPS, 0.0);
    for (int pass = 0; pass < 2; ++pass)
        for (int kc = 0; kc < CPP; ++kc)
            for (int s8 = 0; s8 < KC; ++s8)
                for (int t2 = 0; t2 < HP; ++t2)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 2; ++e) {
                            const int i = 16 * (pass * NTP + 2 * t2 + e) + (l & 15);
                            const int k = 4 * (kc * KC + s8) + (l >> 4);
                            const size_t ci = (size_t)pass * CPP + kc;
                            frag[((((ci * KC + s8) * HP + t2) * 64) + l) * 2 + e] =
                                (i < D && k < D) ? P[(size_t)i * D + k] : 0.0;
                        }
    for (int d = 0; d < D; ++d) mu[d] = mean ? mean[d] : 0.0;
    PBBI_HIP(hipMalloc(&pot->d_sfrag, frag.size() * sizeof(double)));
    PBBI_HIP(hipMalloc(&pot->d_smean, mu.size() * sizeof(double)));
    PBBI_HIP(hipMemcpy(pot->d_sfrag, frag.data(), frag.size() * sizeof(double), hipMemcpyHostToDevice));
    PBBI_HIP(hipMemcpy(pot->d_smean, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice));
    pot->DPS = DPS;
    return PBBI_OK;
}

// Does the streamed kernel serve these arguments?  (PBBI_NO_DENSE_STREAM: A/B switch back to the GEMM path.)
bool dense_stream_applies(const IterArgs& a) {
    static const bool off = (getenv("PBBI_NO_DENSE_STREAM") != nullptr);
    const pbbi_potential* pot = a.pot;
    if (off || (a.route_hint & PBBI_ROUTE_NO_DENSE_STREAM) || pot->kind != KIND_GAUSS_DENSE || pot->DPS == 0) return false;
    if (a.L < 1) return false;
    if (a.method != PBBI_LEAPFROG && a.method != PBBI_STORMER_VERLET) return false;
    if (pbbi_dyn(a) && a.method != PBBI_LEAPFROG) return false;   // per-chain lengths: Leapfrog (as at D <= 128)
    const int64_t ld = a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out;
    return ld_fits(pot, ld > a.N ? ld : a.N);  // 32-bit row offsets: DPS * stride < 2^29 elements (N < 2^21 at D = 256)
}

// ... and may a run on them carry the gradient (two DPS x N slabs behind one descriptor)?
bool dense_stream_carry_applies(const IterArgs& a) {
    static const bool off = (getenv("PBBI_NO_CARRY") != nullptr);
    return !off && dense_stream_applies(a) && !pbbi_dyn(a) && a.N > 0 &&
           (uint64_t)a.pot->DPS * (uint64_t)a.N * 16u < PBBI_CARRY_MAX_BYTES;
}

int dense_stream_fused_iterations(const IterArgs& a) {
    static const int chunk = [] {
        const char* e = getenv("PBBI_DENSE_FUSE");
        const int v = e ? atoi(e) : 64;
        return v < 1 ? 1 : v;
    }();
    if (a.carry == 0 || !a.carry_g || !a.carry_sel || !a.rng || !dense_stream_carry_applies(a)) return 1;
    if (a.ldn_in != a.ldn_out) return 1;  // a run's first iteration reads the caller's stride: a fused launch of one
    return chunk;
}

int dense_stream_hmc_iter(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (!dense_stream_applies(a)) return pbbi_fail(PBBI_ERR_INVALID, "streamed dense kernel: not applicable (internal)");
    if (a.N == 0) return PBBI_OK;
    DensePrm prm{};
#ifdef PBBI_STAMPS
    prm.stamps = g_stamp_buf;
#endif
    prm.frag = (const double*)pot->d_sfrag;
    prm.mu = (const double*)pot->d_smean;
    prm.q_in = (const double*)a.q_in;
    prm.p_in = (const double*)a.p_in;
    prm.u_in = (const double*)a.u_in;
    prm.mass = (const double*)a.mass;
    prm.q_out = (double*)a.q_out;
    prm.p_out = (double*)a.p_out;
    prm.ratio_out = (double*)a.ratio_out;
    prm.reject_out = a.reject_out;
    prm.N = a.N; prm.ldn_in = a.ldn_in; prm.ldn_out = a.ldn_out;
    prm.h = a.h; prm.cst = pot->cst; prm.kT = a.kT;
    prm.L = a.L; prm.D = pot->D; prm.flags = a.flags; prm.rng = a.rng; prm.mode = 0;
    prm.seed = a.seed; prm.iter = a.iter; prm.chain0 = a.chain0;
    prm.steps_in = a.steps_in; prm.steps_out = a.steps_out;
    const bool dyn = pbbi_dyn(a);
    // A carried iteration always runs in the fused kernel (one shape for a run's first and later iterations and
    // for one or many of them per launch): a call that covers one iteration is a fused launch of length 1.
    const bool carried = a.carry && a.carry_g && a.carry_sel && a.rng && dense_stream_carry_applies(a);
    if (a.fuse_S > 1 && !carried)
        return pbbi_fail(PBBI_ERR_INVALID, "fused dense iterations belong to a carried run (internal)");
    if (carried) {
        prm.carry_g = (double*)a.carry_g;
        prm.carry_sel = a.carry_sel;
        prm.carry_slab_bytes = (uint32_t)((uint64_t)pot->DPS * (uint64_t)a.N * 8u);
        prm.fuse_first = (a.carry == 1);
        prm.fuse_slab = (int64_t)pot->D * a.N;
        if (a.fuse_S > 1) {
            if (a.ldn_in != a.ldn_out) return pbbi_fail(PBBI_ERR_INVALID, "fused dense iterations: one stride (internal)");
            prm.fuse_S = a.fuse_S;
            prm.fuse_wrap2 = a.fuse_wrap2;
            prm.fuse_slab0 = a.fuse_slab0;
            prm.fuse_q_base = (double*)a.fuse_q_base;
        } else {  // iteration 0 of a "launch" of one: slab 0 of a base that is this call's q_out
            prm.fuse_S = 1;
            prm.fuse_wrap2 = 0;
            prm.fuse_slab0 = 0;
            prm.fuse_q_base = (double*)a.q_out;
        }
    }
    const size_t lds = stream_lds_bytes(pot->DPS);
    const dim3 grid((unsigned)((a.N + CHAINS_PER_WG - 1) / CHAINS_PER_WG)), block(BLOCK);
#define LAUNCH_S(NT_, F_)                                                                                            \
    {                                                                                                            \
        if (dyn) { /* PBBI_PER_CHAIN_STEPS / PBBI_UTURN_STOP: the padded-rows form serves full tiles too */       \
            auto k = k_dense_hmc<NT_, false, 0, false, PBBI_LEAPFROG, true, 0, false, 2, true>;                  \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        } else if (carried && a.method == PBBI_STORMER_VERLET && (a.flags & PBBI_DRAW_F64)) {                           \
            auto k = k_dense_hmc<NT_, F_, 0, false, PBBI_STORMER_VERLET, false, 2, true, 1, true>;            \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        } else if (carried && a.method == PBBI_STORMER_VERLET) {                                                 \
            auto k = k_dense_hmc<NT_, F_, 0, false, PBBI_STORMER_VERLET, false, 2, true, 0, true>;            \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        } else if (carried && (a.flags & PBBI_DRAW_F64)) { /* the draw's precision at compile time: registers */ \
            auto k = k_dense_hmc<NT_, F_, 0, false, PBBI_LEAPFROG, false, 2, true, 1, true>;                  \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        } else if (carried) {                                                                                    \
            auto k = k_dense_hmc<NT_, F_, 0, false, PBBI_LEAPFROG, false, 2, true, 0, true>;                  \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        } else if (a.method == PBBI_LEAPFROG) {                                                                  \
            auto k = k_dense_hmc<NT_, F_, 0, false, PBBI_LEAPFROG, false, 0, false, 2, true>;                 \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        } else {                                                                                                 \
            auto k = k_dense_hmc<NT_, F_, 0, false, PBBI_STORMER_VERLET, false, 0, false, 2, true>;           \
            if (int rc = set_lds_s(k, lds)) return rc;                                                           \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                              \
        }                                                                                                        \
    }
    // (D == DPS: no padded rows, no column skipping -- the exits cost the full tile 4 %)
    const bool full = (pot->D == pot->DPS);
    if (pot->DPS == 256) {
        if (full) LAUNCH_S(16, true)
        else LAUNCH_S(16, false)
    } else {
        if (full) LAUNCH_S(12, true)
        else LAUNCH_S(12, false)
    }
#undef LAUNCH_S
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// integrate() (Leapfrog.integrate / StormerVerlet.integrate, src/integrator.py:105-163) on the streamed kernel: MODE 1
// of the same template -- q, p advanced in place, Integrator.v optional.  L >= 1; the GEMM path keeps L = 0.
bool dense_stream_integrate_applies(const IntegrateArgs& a) {
    static const bool off = (getenv("PBBI_NO_DENSE_STREAM") != nullptr);
    const pbbi_potential* pot = a.pot;
    if (off || pot->kind != KIND_GAUSS_DENSE || pot->DPS == 0 || a.L < 1 || a.N < 1) return false;
    if (a.method != PBBI_LEAPFROG && a.method != PBBI_STORMER_VERLET) return false;
    return ld_fits(pot, a.ldn);
}

int dense_stream_integrate(const IntegrateArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (!dense_stream_integrate_applies(a)) return pbbi_fail(PBBI_ERR_INVALID, "streamed dense integrate: not applicable (internal)");
    DensePrm prm{};
    prm.frag = (const double*)pot->d_sfrag;
    prm.mu = (const double*)pot->d_smean;
    prm.q_in = (const double*)a.q;
    prm.p_in = (const double*)a.p;
    prm.mass = (const double*)a.mass;
    prm.q_out = (double*)a.q;
    prm.p_out = (double*)a.p;
    prm.v_out = (double*)a.v_out;
    prm.N = a.N; prm.ldn_in = a.ldn; prm.ldn_out = a.ldn;
    prm.h = a.h; prm.cst = pot->cst; prm.kT = 1.0;
    prm.L = a.L; prm.D = pot->D; prm.mode = 1;
    const size_t lds = stream_lds_bytes(pot->DPS);
    const dim3 grid((unsigned)((a.N + CHAINS_PER_WG - 1) / CHAINS_PER_WG)), block(BLOCK);
#define LAUNCH_I(NT_)                                                                                     \
    {                                                                                                     \
        if (a.method == PBBI_LEAPFROG) {                                                                  \
            auto k = k_dense_hmc<NT_, false, 1, false, PBBI_LEAPFROG, false, 0, false, 2, true>;          \
            if (int rc = set_lds_s(k, lds)) return rc;                                                    \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                       \
        } else {                                                                                          \
            auto k = k_dense_hmc<NT_, false, 1, false, PBBI_STORMER_VERLET, false, 0, false, 2, true>;    \
            if (int rc = set_lds_s(k, lds)) return rc;                                                    \
            hipLaunchKernelGGL(k, grid, block, lds, a.stream, prm);                                       \
        }                                                                                                 \
    }
    if (pot->DPS == 256) LAUNCH_I(16)
    else LAUNCH_I(12)
#undef LAUNCH_I
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}
