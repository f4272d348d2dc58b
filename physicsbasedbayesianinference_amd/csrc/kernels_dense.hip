// kernels_dense.hip -- dense-precision Gaussian potential, D <= 128, fp64, gfx950.
//
// The gradient of U = 0.5 x^T P x is the batched mat-vec G = P X over the ensemble
// (x = q - mu): a (DP x DP)·(DP x 16) product per 16-chain tile, done on the matrix cores
// with v_mfma_f64_16x16x4_f64.  Design (MI355X-first, not a translation of anything):
//
//   * One wave owns a tile of 16 chains for the WHOLE trajectory; its state and the mat-vec
//     accumulator live in registers: lane (g = lane>>4, c = lane&15) holds, for chain c, the
//     dims {4s+g : s = 0..DP/4-1}.  That is simultaneously
//        - the B-operand layout of K-step s      (B[k = lane>>4][n = lane&15]), and
//        - the C/D layout of row tile t, reg r   (row = (lane>>4) + 4r -> dim 16t+4r+g, s = 4t+r),
//     so the MFMA output lands exactly where the next step's operand lives: the
//     leapfrog update is lane-local, no shuffles, no LDS round trip for state.
//   * P (128 KiB at D=128) is staged once per workgroup into LDS, pre-swizzled on the host
//     into A-fragment order [s][t/2][lane][2] so that each ds_read_b128 is lane-linear
//     (conflict-free) and feeds two MFMAs.  mu sits next to it (1 KiB).  129 KiB of LDS
//     pins one workgroup per CU.  P is shared through L2 only (every workgroup reads the
//     same 128 KiB), chains share nothing: there is no inter-tile reuse for an XCD-aware
//     block mapping to exploit, so blockIdx maps to tiles directly.
//   * U(q) = 0.5 x.(P x) reuses the gradient mat-vec (first and last evaluation of the
//     trajectory), so an HMC iteration costs exactly L+1 mat-vecs.
//   * Kinetic/potential sums: lane-local terms, then a 2-step xor butterfly over the 4 lanes
//     that share a chain (every lane ends with identical bits -> uniform decision).
//
// Two kernels share these pieces:
//   k_dense_hmc   production path, both integrators with L >= 1: two waves per SIMD,
//                 kick-drift-kick state, row passes (see the comment above it; 0.86 of the fp64
//                 MFMA peak on BASELINE config 2).
//   k_dense_traj  general path: L = 0, and the reference's literal operation order for both
//                 integrators (A/B switch PBBI_DENSE_V1): one wave per SIMD, 512-register budget,
//                 matrix pipe ~50 % busy.
//
// Only the summation ORDER inside dot products (MFMA k-ordered fma chain, lane-group
// butterfly) and, in k_dense_hmc, the algebraically equivalent kick-drift-kick update differ
// from the oracle, so parity is to fp64 tolerance with equal reject masks, not bitwise
// (src/integrator.py:105-120, :142-163; src/HMC.py:100-102,115,164-179).
#include "kernels_dense_dev.h"

namespace {

struct DenseEvalPrm {
    const double* frag;
    const double* mu;
    const double* q;
    const double* p;
    const double* mass;
    double* U_out;
    double* grad_out;
    double* w_out;
    int64_t N, ldn;
    double cst;
    int D, mode;  // 0: U / grad; 1: H / exp(-H); 2: U_out = exp(U_out - H)
};

template <int NT, bool FULL>
__global__ void __launch_bounds__(BLOCK, 1) k_dense_eval(DenseEvalPrm prm) {
    constexpr int DP = 16 * NT;
    constexpr int KS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f64* frag2 = reinterpret_cast<v2f64*>(smem);
    double* mu = reinterpret_cast<double*>(smem + (size_t)DP * DP * sizeof(double));
    stage_lds<NT>(prm.frag, prm.mu, frag2, mu);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const v2f64* fragL = frag2 + lane;
    const double* muG = mu + g;
    const int D = prm.D;
    const int64_t n_wg_tiles = (prm.N + CHAINS_PER_WG - 1) / CHAINS_PER_WG;
    for (int64_t wt = blockIdx.x; wt < n_wg_tiles; wt += gridDim.x) {
        const int64_t n0 = (wt * 4 + wave) * CHAINS_PER_WAVE;
        if (n0 >= prm.N) continue;
        const int64_t left = prm.N - n0;
        const bool valid = c < left;
        const int cc = valid ? c : (int)left - 1;
        const uint32_t ld = 8u * (uint32_t)prm.ldn;  // byte offsets (pbbi_buf.h)
        const uint32_t voff = (uint32_t)g * ld + 8u * (uint32_t)cc, s4 = 4u * ld;
        const __amdgpu_buffer_rsrc_t qin = buf_make(prm.q + n0);
        double q[KS];
        v4f64 acc[NT];
#pragma unroll
        for (int s = 0; s < KS; ++s) q[s] = load_elem<FULL>(qin, voff, s4, ld, s, g, D);
        matvec<NT>(fragL, muG, q, acc);
        const double xg = chain_sum(dot_x_acc<NT>(muG, q, acc));
        const double U = 0.5 * xg + prm.cst;
        if (prm.mode == 0) {
            if (prm.grad_out) {
                const __amdgpu_buffer_rsrc_t gout = buf_make(prm.grad_out + n0);
                if (valid) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            store_elem<FULL>(gout, voff, s4, 4 * t + r, g, D, acc[t][r]);
                }
            }
            if (prm.U_out && valid && g == 0) prm.U_out[n0 + c] = U;
            continue;
        }
        const __amdgpu_buffer_rsrc_t pin = buf_make(prm.p + n0);
        double pp = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double pv = load_elem<FULL>(pin, voff, s4, ld, s, g, D);
            pp += pv * pv;
        }
        pp = chain_sum(pp);
        const double m = prm.mass ? prm.mass[n0 + cc] : 1.0;
        const double H = 0.5 * pp / m + U;
        if (valid && g == 0) {
            if (prm.mode == 1) {
                if (prm.U_out) prm.U_out[n0 + c] = H;
                if (prm.w_out) prm.w_out[n0 + c] = exp(-H);
            } else {
                prm.U_out[n0 + c] = exp(prm.U_out[n0 + c] - H);
            }
        }
    }
}

int num_cus(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
        cus <= 0)
        cus = 256;
    if (device >= 0 && device < 64) cached[device] = cus;
    return cus;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    PBBI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PBBI_OK;
}

inline size_t lds_bytes(int DP) { return (size_t)DP * DP * 8 + (size_t)DP * 8; }

inline unsigned grid_size(const pbbi_potential* pot, int64_t N) {
    const int64_t tiles = (N + CHAINS_PER_WG - 1) / CHAINS_PER_WG;
    const int64_t cus = num_cus(pot->device);
    return (unsigned)(tiles < cus ? tiles : cus);
}

int launch_traj(const pbbi_potential* pot, int method, const DensePrm& prm, int64_t N,
                hipStream_t stream, int carry_mode = 0) {
    const size_t lds = lds_bytes(pot->DP);
    const dim3 grid(grid_size(pot, N)), block(BLOCK);
    const bool full = (pot->D == pot->DP);
    static const bool use_v1 = (getenv("PBBI_DENSE_V1") != nullptr);  // A/B switch for profiling
    if (prm.L >= 1 && !use_v1) {
        const int64_t tiles2 = (N + CHAINS_PER_WG2 - 1) / CHAINS_PER_WG2;
        const dim3 grid2((unsigned)tiles2), block2(BLOCK2);
        const bool zmean = pot->zero_mean;
        const bool dyn = prm.mode == 0 && (prm.flags & (PBBI_PER_CHAIN_STEPS | PBBI_UTURN_STOP)) != 0;
        const int carry = (!dyn && prm.mode == 0 && prm.carry_g) ? carry_mode : 0;
#define LAUNCH3(NT_, F_, M_, Z_)                                                                  \
    {                                                                                             \
        constexpr bool CARRYK = M_ == 0;   /* carried / fused forms exist */ \
        constexpr int NTC = NT_;                                                                  \
        constexpr bool HEAD = (NT_ == 8) && F_;   /* the C2 shape: draw specialised at compile time */ \
        if (dyn && M_ == 0) {                                                                     \
            if (int rc = set_lds(k_dense_hmc<NT_, F_, 0, Z_, PBBI_LEAPFROG, true>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<NT_, F_, 0, Z_, PBBI_LEAPFROG, true>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else if (method == PBBI_LEAPFROG && CARRYK && carry == 1) {                             \
            if (int rc = set_lds(k_dense_hmc<NTC, F_, 0, Z_, PBBI_LEAPFROG, false, 1>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<NTC, F_, 0, Z_, PBBI_LEAPFROG, false, 1>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else if (method == PBBI_LEAPFROG && CARRYK && carry == 2 && prm.fuse_S > 1 && HEAD &&   \
                   (prm.flags & PBBI_DRAW_F64)) {                                                 \
            if (int rc = set_lds(k_dense_hmc<8, true, 0, Z_, PBBI_LEAPFROG, false, 2, true, 1>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<8, true, 0, Z_, PBBI_LEAPFROG, false, 2, true, 1>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else if (method == PBBI_LEAPFROG && CARRYK && carry == 2 && prm.fuse_S > 1 && HEAD) {   \
            if (int rc = set_lds(k_dense_hmc<8, true, 0, Z_, PBBI_LEAPFROG, false, 2, true, 0>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<8, true, 0, Z_, PBBI_LEAPFROG, false, 2, true, 0>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else if (method == PBBI_LEAPFROG && CARRYK && carry == 2 && prm.fuse_S > 1) {           \
            if (int rc = set_lds(k_dense_hmc<NTC, F_, 0, Z_, PBBI_LEAPFROG, false, 2, true, 2>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<NTC, F_, 0, Z_, PBBI_LEAPFROG, false, 2, true, 2>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else if (method == PBBI_LEAPFROG && CARRYK && carry == 2) {                             \
            if (int rc = set_lds(k_dense_hmc<NTC, F_, 0, Z_, PBBI_LEAPFROG, false, 2>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<NTC, F_, 0, Z_, PBBI_LEAPFROG, false, 2>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else if (method == PBBI_STORMER_VERLET && CARRYK && carry == 2) { /* always the fused form */ \
            if (int rc = set_lds(k_dense_hmc<NTC, F_, 0, Z_, PBBI_STORMER_VERLET, false, 2, true, 2>, lds)) return rc; \
            hipLaunchKernelGGL((k_dense_hmc<NTC, F_, 0, Z_, PBBI_STORMER_VERLET, false, 2, true, 2>), grid2, block2, \
                               lds, stream, prm);                                                 \
        } else if (method == PBBI_LEAPFROG) {                                                     \
            if (int rc = set_lds(k_dense_hmc<NT_, F_, M_, Z_, PBBI_LEAPFROG>, lds)) return rc;    \
            hipLaunchKernelGGL((k_dense_hmc<NT_, F_, M_, Z_, PBBI_LEAPFROG>), grid2, block2, lds, \
                               stream, prm);                                                      \
        } else {                                                                                  \
            if (int rc = set_lds(k_dense_hmc<NT_, F_, M_, Z_, PBBI_STORMER_VERLET>, lds))         \
                return rc;                                                                        \
            hipLaunchKernelGGL((k_dense_hmc<NT_, F_, M_, Z_, PBBI_STORMER_VERLET>), grid2,        \
                               block2, lds, stream, prm);                                         \
        }                                                                                         \
    }
#define CASE3F(NT_, F_)                                   \
    {                                                     \
        if (prm.mode == 0) {                              \
            if (zmean) LAUNCH3(NT_, F_, 0, true)          \
            else LAUNCH3(NT_, F_, 0, false)               \
        } else {                                          \
            if (zmean) LAUNCH3(NT_, F_, 1, true)          \
            else LAUNCH3(NT_, F_, 1, false)               \
        }                                                 \
    }
#define CASE3(NT_)                    \
    if (pot->DP == 16 * NT_) {        \
        if (full) CASE3F(NT_, true)   \
        else CASE3F(NT_, false)       \
    }
        CASE3(2) CASE3(4) CASE3(6) CASE3(8)
#undef CASE3
#undef CASE3F
#undef LAUNCH3
        PBBI_HIP(hipGetLastError());
        return PBBI_OK;
    }
#define LAUNCH(NT_, M_, F_)                                                          \
    {                                                                                \
        if (int rc = set_lds(k_dense_traj<NT_, M_, F_>, lds)) return rc;             \
        hipLaunchKernelGGL((k_dense_traj<NT_, M_, F_>), grid, block, lds, stream, prm); \
    }
#define CASE(NT_)                                                 \
    if (pot->DP == 16 * NT_) {                                    \
        if (method == PBBI_LEAPFROG) {                            \
            if (full) LAUNCH(NT_, PBBI_LEAPFROG, true)            \
            else LAUNCH(NT_, PBBI_LEAPFROG, false)                \
        } else {                                                  \
            if (full) LAUNCH(NT_, PBBI_STORMER_VERLET, true)      \
            else LAUNCH(NT_, PBBI_STORMER_VERLET, false)          \
        }                                                         \
    }
    CASE(2) CASE(4) CASE(6) CASE(8)
#undef CASE
#undef LAUNCH
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int check_ld(const pbbi_potential* pot, int64_t ld) {
    if ((int64_t)pot->DP * ld >= ((int64_t)1 << 29))
        return pbbi_fail(PBBI_ERR_UNSUPPORTED,
                         "dense kernels address a lane's rows with 32-bit offsets: padded D * "
                         "leading stride must be < 2^29 elements; shard the ensemble");
    return PBBI_OK;
}

int check(const pbbi_potential* pot) {
    if (pot->dtype != PBBI_F64)
        return pbbi_fail(PBBI_ERR_UNSUPPORTED, "dense Gaussian kernels are fp64 only in this build");
    if (pot->DP == 0 || pot->d_frag == nullptr)
        return pbbi_fail(PBBI_ERR_UNSUPPORTED,
                         "dense Gaussian: register-resident MFMA path needs D <= 128 (D = " +
                             std::to_string(pot->D) + ")");
    return PBBI_OK;
}

}  // namespace

// P (row-major, D x D) -> A-fragment order [s][t/2][lane][t&1], zero padded to DP.
// Lane l of fragment (s, t) supplies A[i = l&15][k = l>>4] = P[16t + (l&15)][4s + (l>>4)].
int dense_build_fragments(pbbi_potential* pot, const double* P, const double* mean) {
    const int D = pot->D;
    pot->DP = 0;
    pot->d_frag = nullptr;
    pot->d_mean_pad = nullptr;
    if (D > 128 || pot->dtype != PBBI_F64) return PBBI_OK;  // path unavailable; callers check
    const int DP = D <= 32 ? 32 : (D <= 64 ? 64 : (D <= 96 ? 96 : 128));
    const int NT = DP / 16, KS = DP / 4;
    std::vector<double> frag((size_t)DP * DP, 0.0), mu((size_t)DP, 0.0);
    for (int s = 0; s < KS; ++s)
        for (int t = 0; t < NT; ++t)
            for (int l = 0; l < 64; ++l) {
                const int i = 16 * t + (l & 15), k = 4 * s + (l >> 4);
                const double val = (i < D && k < D) ? P[(size_t)i * D + k] : 0.0;
                frag[(((size_t)s * (NT / 2) + t / 2) * 64 + l) * 2 + (t & 1)] = val;
            }
    pot->zero_mean = true;
    for (int d = 0; d < D; ++d) {
        mu[d] = mean ? mean[d] : 0.0;
        if (mu[d] != 0.0) pot->zero_mean = false;
    }
    PBBI_HIP(hipMalloc(&pot->d_frag, frag.size() * sizeof(double)));
    PBBI_HIP(hipMalloc(&pot->d_mean_pad, mu.size() * sizeof(double)));
    PBBI_HIP(hipMemcpy(pot->d_frag, frag.data(), frag.size() * sizeof(double), hipMemcpyHostToDevice));
    PBBI_HIP(hipMemcpy(pot->d_mean_pad, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice));
    pot->DP = DP;
    return PBBI_OK;
}

int dense_hmc_iter(const IterArgs& a) {
    if (int rc = check(a.pot)) return rc;
    if (pbbi_dyn(a) && (a.method != PBBI_LEAPFROG || a.L < 1))
        return pbbi_fail(PBBI_ERR_UNSUPPORTED, "per-chain trajectory lengths on the dense kernel: Leapfrog with L >= 1");
    if (int rc = check_ld(a.pot, a.ldn_in > a.ldn_out ? a.ldn_in : a.ldn_out)) return rc;
    if (a.N == 0) return PBBI_OK;
    DensePrm prm{};
#ifdef PBBI_STAMPS
    prm.stamps = g_stamp_buf;
#endif
    prm.frag = (const double*)a.pot->d_frag;
    prm.mu = (const double*)a.pot->d_mean_pad;
    prm.q_in = (const double*)a.q_in;
    prm.p_in = (const double*)a.p_in;
    prm.u_in = (const double*)a.u_in;
    prm.mass = (const double*)a.mass;
    prm.q_out = (double*)a.q_out;
    prm.p_out = (double*)a.p_out;
    prm.v_out = nullptr;
    prm.ratio_out = (double*)a.ratio_out;
    prm.reject_out = a.reject_out;
    prm.N = a.N; prm.ldn_in = a.ldn_in; prm.ldn_out = a.ldn_out;
    prm.h = a.h; prm.cst = a.pot->cst; prm.kT = a.kT;
    prm.L = a.L; prm.D = a.pot->D; prm.flags = a.flags; prm.rng = a.rng; prm.mode = 0;
    prm.seed = a.seed; prm.iter = a.iter; prm.chain0 = a.chain0;
    prm.steps_in = a.steps_in; prm.steps_out = a.steps_out;
    int carry = 0;
    if (a.carry && a.carry_g && a.carry_sel && dense_carry_applies(a)) {
        carry = a.carry;
        prm.carry_g = (double*)a.carry_g;
        prm.carry_sel = a.carry_sel;
        prm.carry_slab_bytes = (uint32_t)((uint64_t)a.pot->DP * (uint64_t)a.N * 8u);  // slabs hold the padded rows too
    }
    if (a.fuse_S > 1) {
        if (carry == 0 || a.ldn_in != a.ldn_out || !a.rng)
            return pbbi_fail(PBBI_ERR_INVALID, "fused dense iterations belong to a carried run (internal)");
        prm.fuse_first = (carry == 1);  // the launch starts the run: its iteration 0 forms g(q_0)
        carry = 2;
        prm.fuse_S = a.fuse_S;
        prm.fuse_wrap2 = a.fuse_wrap2;
        prm.fuse_slab0 = a.fuse_slab0;
        prm.fuse_slab = (int64_t)a.pot->D * a.N;
        prm.fuse_q_base = (double*)a.fuse_q_base;
    } else if (carry != 0 && a.method == PBBI_STORMER_VERLET) {
        // carried Stormer-Verlet iterations exist in the fused form only: a call that covers one iteration is a
        // fused launch of one (slab 0 of a base that is this call's q_out; the first iteration of a run reads
        // the caller's state with its own stride)
        if (!a.rng) return pbbi_fail(PBBI_ERR_INVALID, "carried dense iterations draw in the kernel (internal)");
        prm.fuse_first = (carry == 1);
        carry = 2;
        prm.fuse_S = 1;
        prm.fuse_wrap2 = 0;
        prm.fuse_slab0 = 0;
        prm.fuse_slab = (int64_t)a.pot->D * a.N;
        prm.fuse_q_base = (double*)a.q_out;
    }
    return launch_traj(a.pot, a.method, prm, a.N, a.stream, carry);
}

// Iterations of a run that ONE launch may cover (k_dense_hmc FUSE): a run that carries its gradient, in-kernel
// draws, the state the launch starts from stored with the slabs' stride (a run's first iteration reads the
// caller's q_state: ldn == N, else that iteration gets a launch of its own).  PBBI_DENSE_FUSE=n overrides the
// chunk (1 = off).
int dense_fused_iterations(const IterArgs& a) {
    static const int chunk = [] {
        const char* e = getenv("PBBI_DENSE_FUSE");
        const int v = e ? atoi(e) : 64;
        return v < 1 ? 1 : v;
    }();
    if (a.carry == 0 || !a.carry_g || !a.carry_sel || !a.rng || a.ldn_in != a.ldn_out || !dense_carry_applies(a))
        return 1;
    return chunk;
}

// (both slabs are addressed from ONE descriptor with unsigned 32-bit byte offsets: row offset + lane offset + slab
//  select stay below 2^32 while the two slabs together do, with a margin for the lane term)
// May the iterations of a run on these arguments carry the gradient (k_dense_hmc, CARRY)?  Fixed-length
// Leapfrog / Stormer-Verlet on the two-wave kernel at D <= 128 (padded D included: DP = 32, 64, 96 or 128), both slabs -- DP rows
// each -- addressable with 32-bit offsets.
bool dense_carry_applies(const IterArgs& a) {
    static const bool off = (getenv("PBBI_NO_CARRY") != nullptr);  // A/B switch
    return !off && (a.method == PBBI_LEAPFROG || a.method == PBBI_STORMER_VERLET) && a.L >= 1 && !pbbi_dyn(a) && a.pot->DP != 0 &&
           a.N > 0 && (uint64_t)a.pot->DP * (uint64_t)a.N * 16u < PBBI_CARRY_MAX_BYTES &&
           getenv("PBBI_DENSE_V1") == nullptr;
}

int dense_integrate(const IntegrateArgs& a) {
    if (int rc = check(a.pot)) return rc;
    if (int rc = check_ld(a.pot, a.ldn)) return rc;
    if (a.N == 0) return PBBI_OK;
    DensePrm prm{};
    prm.frag = (const double*)a.pot->d_frag;
    prm.mu = (const double*)a.pot->d_mean_pad;
    prm.q_in = (const double*)a.q;
    prm.p_in = (const double*)a.p;
    prm.mass = (const double*)a.mass;
    prm.q_out = (double*)a.q;
    prm.p_out = (double*)a.p;
    prm.v_out = (double*)a.v_out;
    prm.N = a.N; prm.ldn_in = a.ldn; prm.ldn_out = a.ldn;
    prm.h = a.h; prm.cst = a.pot->cst; prm.kT = 1.0;
    prm.L = a.L; prm.D = a.pot->D; prm.mode = 1;
    return launch_traj(a.pot, a.method, prm, a.N, a.stream);
}

static int dense_eval_launch(const EvalArgs& a, int mode) {
    if (int rc = check(a.pot)) return rc;
    if (int rc = check_ld(a.pot, a.ldn)) return rc;
    if (a.N == 0) return PBBI_OK;
    const pbbi_potential* pot = a.pot;
    DenseEvalPrm prm{(const double*)pot->d_frag, (const double*)pot->d_mean_pad,
                     (const double*)a.q, (const double*)a.p, (const double*)a.mass,
                     (double*)a.U_out, (double*)a.grad_out, (double*)a.w_out,
                     a.N, a.ldn, pot->cst, pot->D, mode};
    const size_t lds = lds_bytes(pot->DP);
    const dim3 grid(grid_size(pot, a.N)), block(BLOCK);
    const bool full = (pot->D == pot->DP);
#define LAUNCH(NT_, F_)                                                               \
    {                                                                                 \
        if (int rc = set_lds(k_dense_eval<NT_, F_>, lds)) return rc;                  \
        hipLaunchKernelGGL((k_dense_eval<NT_, F_>), grid, block, lds, a.stream, prm); \
    }
#define CASE(NT_)                          \
    if (pot->DP == 16 * NT_) {             \
        if (full) LAUNCH(NT_, true)        \
        else LAUNCH(NT_, false)            \
    }
    CASE(2) CASE(4) CASE(6) CASE(8)
#undef CASE
#undef LAUNCH
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int dense_eval(const EvalArgs& a) { return dense_eval_launch(a, 0); }
int dense_energy(const EvalArgs& a) { return dense_eval_launch(a, a.ratio_finish ? 2 : 1); }
