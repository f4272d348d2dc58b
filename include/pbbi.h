/*
 * pbbi.h -- C ABI of libpbbi.so, the MI355X (gfx950) ensemble-HMC hot path.
 *
 * Drop-in boundary.  The reference (Anton-Le/PhysicsBasedBayesianInference) has
 * no FFI layer: its boundary is the Python class API of src/{ensemble,
 * integrator,HMC,potential}.py.  Each entry point below replaces the arithmetic
 * behind one of those methods (cited as file:line relative to the reference
 * root) and is what a ctypes binding in those files would call -- see
 * INTEGRATION.md for the stub a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no exceptions cross the ABI.
 *   - Every function returns an int status (PBBI_OK = 0, negative = error
 *     class); pbbi_last_error() returns a thread-local message for the last
 *     failure on the calling thread.
 *   - State arrays are DEVICE pointers owned by the caller (e.g.
 *     torch.Tensor.data_ptr()); the library never frees or retains them past
 *     the call.  Layout: (D, N) "ensemble-major" -- element (d, n) at
 *     [d*ldn + n], chain index n fastest, ldn >= N.  This is the reference's
 *     np.zeros((numDimensions, numParticles)) C-order layout
 *     (src/ensemble.py:40-41).  Element type is the potential handle's dtype.
 *   - `mass` is a device array of N elements or NULL (= all ones, the
 *     reference default src/ensemble.py:42; NULL also selects the division-free
 *     fast path, which is exact for unit mass).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Calls are asynchronous with respect to the host.
 *   - Potential handles are immutable after creation; the library is
 *     re-entrant; one handle may be used from several streams.
 *   - There is NO CPU fallback: every entry point launches HIP kernels on the
 *     handle's device and fails with PBBI_ERR_HIP if that is impossible.
 *
 * RNG contract (device "philox" mode; pbbi_hmc_run, pbbi_philox_*)
 *   Philox-4x32-10, key = (seed_lo32, seed_hi32),
 *   counter = (chain_lo32, block, iteration_lo32, stream | chain_hi24 << 8),
 *   where `chain` is the GLOBAL chain index (chain0 + n), so results do not
 *   depend on how the ensemble is sharded over GPUs.
 *   Standard normals: one block serves the four dims d, d+4, d+8, d+12 of a group of 16
 *     (block = ((dim>>4)<<2) | (dim&3), slot = (dim>>2)&3) through two SINGLE-PRECISION
 *     Box-Muller transforms, (x0,x1) -> slots 0,1 and (x2,x3) -> slots 2,3:
 *       u1 = a*2^-32 + 2^-33 in (0,1],  u2 = (b>>8)*2^-24,  r = sqrtf(-2 ln u1),
 *       z_even = r*cos(2 pi u2),  z_odd = r*sin(2 pi u2),  widened to the array dtype.
 *     The device evaluates log2/sqrt/sin/cos on its transcendental unit; draws are
 *     reproducible bit for bit on the device (pbbi_philox_normal returns exactly what
 *     pbbi_hmc_run draws), exact N(0,1) on a 2^-24-relative grid with tails to 6.7 sigma.
 *   Flag PBBI_DRAW_F64 (stream bit PBBI_STREAM_DRAW_F64 of pbbi_philox_normal): DOUBLE-PRECISION Box-Muller,
 *     the counterpart of the reference's float64 draws (src/ensemble.py:72-74,88-91).  Two blocks per four
 *     dims: slots 0,1 from block blk, slots 2,3 from block blk | 0x80000000.  With w1 = x1:x0, w2 = x3:x2:
 *       u1 = ((w1 >> 12) + 0.5) * 2^-52 in (0,1);   k2 = w2 >> 11,  angle 2 pi k2 2^-53 = (pi/2)(n + y),
 *       n = (k2 + 2^50) >> 51,  y = (k2 - n 2^51) 2^-51 in [-1/2, 1/2)   (integer arithmetic, exact);
 *       r = sqrt(-2 ln u1), ln by fdlibm's e_log.c scheme (s = f/(2+f), Lg1..Lg7, Horner with fma);
 *       sin / cos((pi/2) y) by Taylor polynomials in y^2 (9 / 10 terms, fma), rotated by n & 3;
 *       z_even = r cos(angle),  z_odd = r sin(angle).
 *     Only +, -, *, /, sqrt, fma in a fixed order: the device's variates and the oracle's are THE SAME BITS
 *     (csrc/pbbi_rng.h, oracle/pbbi_oracle.c); exact N(0,1) on a 2^-52 grid with tails to 8.57 sigma.
 *   Metropolis uniform of a chain: block = 0xFFFFFFFF, stream = PBBI_STREAM_UNIFORM,
 *     u = ((x1:x0)>>11) * 2^-53 in [0,1).
 *   The integer part is bit-identical to oracle/pbbi_oracle.c; the single-precision
 *   transcendental part agrees with the host mirror to ~1e-6 absolute.
 */
#ifndef PBBI_H
#define PBBI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBBI_VERSION 103 /* major*100 + minor */

enum { PBBI_OK = 0, PBBI_ERR_INVALID = -1, PBBI_ERR_UNSUPPORTED = -2, PBBI_ERR_HIP = -3 };
enum { PBBI_F64 = 0, PBBI_F32 = 1 };
/* method: the two Integrator subclasses, src/integrator.py:94,126 */
enum { PBBI_LEAPFROG = 0, PBBI_STORMER_VERLET = 1 };
/* flags */
enum {
    /* reproduce src/HMC.py:176: a rejected chain's stored momentum is its OLD POSITION.
     * Without the flag the stored momentum of a rejected chain is the drawn momentum. */
    PBBI_COMPAT_P_FROM_OLDQ = 1,
    /* Opt-in throughput form: integrate in kick-drift-kick form with fused
     * multiply-adds (state q and the half-step velocity; 2 instead of 7 fp64 instructions per
     * element-step for the updates, no acceleration array).  Algebraically the reference's
     * velocity-Verlet (src/integrator.py:105-120); results agree to ~1e-13 relative instead of bit
     * for bit, accept masks as before.  Honoured by the two-lane Rosenbrock kernel (config C3) and
     * by the multi-lane / multi-wave harmonic, diagonal-Gaussian and Rosenbrock kernels (D <= 256); the MFMA kernels always
     * integrate this way, the other chain-per-lane kernels ignore it.  Stormer-Verlet under the flag
     * is the same recurrence without the closing half kick and with one more drift (honoured by the
     * separable kernel up to D = 256 and the in-wave Rosenbrock kernel up to D = 128). */
    PBBI_KDK_FMA = 2,
    /* Accept test at the temperature of the momentum draw: ratio = exp((oldH - newH) / kT) instead of
     * the reference's exp(oldH - newH) (src/HMC.py:115 has no beta although src/ensemble.py:88 draws p at
     * kB*T; the two agree only for T = 1/kB).  With the flag the chains sample exp(-U(q)/kT): the
     * canonical ensemble at that temperature.  kT is pbbi_hmc_run's argument; uploaded-draw iterations
     * state it through pbbi_hmc_iter_kt.  Without the flag (default) the reference's test is kept. */
    PBBI_BETA_ACCEPT = 4,
    /* Per-chain trajectory lengths (pbbi_hmc_iter_dyn / pbbi_hmc_run_dyn; SURVEY 8f row 2 -- planned in
     * the reference's WeekPlan.md:16-17, not a reference feature).  PBBI_PER_CHAIN_STEPS: chain n takes
     * its OWN number of leapfrog steps L_n in [1, L] -- steps_in[n] (uploaded draws) or
     * 1 + floor(u * L) with u from PBBI_STREAM_STEPS (in-kernel draws); lengths drawn independently of
     * the state keep the HMC kernel valid, and a fixed length's resonances go away chain by chain.
     * PBBI_UTURN_STOP: a chain stops after the first step j at which (q_j - q_0) . p_j < 0 (the
     * no-U-turn criterion), at most L (or L_n) steps; steps_out[n] records where.  Stopping at the
     * U-turn alone is NOT a reversible kernel: the flag is for measuring trajectory lengths during
     * warm-up (HMC.adaptTrajectoryLength), not for the recorded run.  Lanes of finished chains are
     * masked out; a wave leaves the loop when its last chain has. */
    PBBI_PER_CHAIN_STEPS = 8,
    PBBI_UTURN_STOP = 16,
    /* Momenta drawn in DOUBLE precision (pbbi_hmc_run / pbbi_hmc_run_dyn, every kernel family): the
     * counterpart of the reference's float64 normals (src/ensemble.py:72-74,88-91).  See "RNG contract":
     * bit-identical to oracle/pbbi_oracle.c, 52-bit radius / 53-bit angle, tails to 8.57 sigma.  Costs
     * ~2.5x the vector instructions of the default single-precision draw (the C2 headline: see DESIGN.md). */
    PBBI_DRAW_F64 = 32
};
enum { PBBI_STREAM_MOMENTUM = 0, PBBI_STREAM_POSITION = 1, PBBI_STREAM_UNIFORM = 2, PBBI_STREAM_STEPS = 3,
       PBBI_STREAM_SWAP = 4, /* replica-exchange uniforms (pbbi_replica_exchange) */
       /* OR-ed into pbbi_philox_normal's rng_stream: the draw PBBI_DRAW_F64 selects in pbbi_hmc_run */
       PBBI_STREAM_DRAW_F64 = 0x100 };

typedef struct pbbi_potential pbbi_potential; /* opaque */

typedef struct {
    char name[128];
    char arch[64];
    int compute_units;
    int64_t hbm_bytes;
    int lds_bytes_per_block;
    int clock_khz;
} pbbi_devinfo;

/* ---- library ------------------------------------------------------------- */
int pbbi_version(void);
const char* pbbi_last_error(void);
int pbbi_device_count(int* count_out);
int pbbi_device_info(int device, pbbi_devinfo* out);

/* ---- potentials ----------------------------------------------------------
 * "potential" in the reference is any callable q(D,) -> scalar with a gradient
 * callable (D,) -> (D,) (src/integrator.py:73, src/HMC.py:102).  A GPU kernel
 * cannot call Python, so the closed forms the reference exercises are handles.
 * All parameter arrays are HOST pointers (float64), copied at creation.
 *   harmonic   U = 0.5*dot(k, q**2)                     src/potential.py:18-27
 *   gauss_diag U = 0.5*dot(prec*(q-mu), q-mu) + cst
 *   gauss_dense U = 0.5*dot(x, P x) + cst, x = q-mu, grad = P x with P symmetric
 *              (= -multivariate_normal.logpdf, src/tests/test_HMC.py:49,125).
 *              `precision` is D x D row-major and MUST be symmetric.
 *              (fp64 HMC iterations: D <= 128 on the register-resident MFMA kernel with P in LDS, 128 < D <= 256 on
 *              the same kernel with P streamed through LDS, beyond -- and fp32 -- one fused GEMM per leapfrog step.)
 *   rosenbrock U = sum_{i<D-1} [b (q_{i+1}-q_i^2)^2 + (a-q_i)^2] / s
 *              (defined by this build).
 * mean may be NULL (= 0).
 */
int pbbi_potential_create_harmonic(int D, const double* springConsts, int dtype, int device,
                                   pbbi_potential** out);
int pbbi_potential_create_gauss_diag(int D, const double* mean, const double* prec, double cst,
                                     int dtype, int device, pbbi_potential** out);
int pbbi_potential_create_gauss_dense(int D, const double* mean, const double* precision,
                                      double cst, int dtype, int device, pbbi_potential** out);
int pbbi_potential_create_rosenbrock(int D, double a, double b, double s, int dtype, int device,
                                     pbbi_potential** out);
/* User-defined potential -- the counterpart of the reference's arbitrary Python callables
 * (`potential=` / `gradient=` of src/HMC.py:35-60, src/integrator.py:36-59).  The two functions
 * are stated in C++ (contract in physicsbasedbayesianinference_amd/custom.py and
 * csrc/pbbi_custom.h), compiled by hipcc for gfx950 into a plugin shared object whose kernels
 * have them inlined; `plugin_path` names that file.  `params` (host, float64, may be NULL when
 * n_params == 0) is copied to the device in the handle's dtype and passed to both functions:
 * model constants, or the data set of a Bayesian model. */
int pbbi_potential_create_custom(const char* plugin_path, int D, const double* params, int n_params,
                                 int dtype, int device, pbbi_potential** out);
int pbbi_potential_destroy(pbbi_potential* pot);
int pbbi_potential_dim(const pbbi_potential* pot);
int pbbi_potential_dtype(const pbbi_potential* pot);
int pbbi_potential_device(const pbbi_potential* pot);

/* potential(q[:, n]) and gradient(q[:, n]) for every chain.
 * U_out: N elements or NULL; grad_out: (D, N) with stride ldn or NULL.
 * Replaces the user callables the reference invokes per chain
 * (src/integrator.py:73, src/HMC.py:102,111,114). */
int pbbi_potential_eval(const pbbi_potential* pot, const void* q, int64_t N, int64_t ldn,
                        void* U_out, void* grad_out, void* stream);

/* ---- integrators ---------------------------------------------------------
 * Leapfrog.integrate() src/integrator.py:95-123 / StormerVerlet.integrate()
 * src/integrator.py:127-165, over the whole ensemble, in place on q and p.
 * L = numSteps = int(finalTime/stepSize), computed by the host
 * (src/integrator.py:51).  v_out: optional (D, N) receiving Integrator.v.
 */
int pbbi_integrate(const pbbi_potential* pot, int method, void* q, void* p, const void* mass,
                   void* v_out, int64_t N, int64_t ldn, double h, int L, void* stream);
int pbbi_leapfrog(const pbbi_potential* pot, void* q, void* p, const void* mass, int64_t N,
                  int64_t ldn, double h, int L, void* stream);
int pbbi_stormer_verlet(const pbbi_potential* pot, void* q, void* p, const void* mass, int64_t N,
                        int64_t ldn, double h, int L, void* stream);

/* ---- energies ------------------------------------------------------------
 * H = 0.5*dot(p,p)/mass + potential(q) per chain          src/HMC.py:100-102
 * pbbi_energy: H_out (N) and/or weight_out (N) = exp(-H)  (HMC.getWeights :86-104)
 * pbbi_weights_ratio: exp(oldH - newH)                    (HMC.getWeightsRatio :106-116)
 */
int pbbi_energy(const pbbi_potential* pot, const void* q, const void* p, const void* mass,
                int64_t N, int64_t ldn, void* H_out, void* weight_out, void* stream);
int pbbi_weights_ratio(const pbbi_potential* pot, const void* newQ, const void* newP,
                       const void* oldQ, const void* oldP, const void* mass, int64_t N,
                       int64_t ldn, void* ratio_out, void* stream);

/* ---- fused HMC iteration -------------------------------------------------
 * One pass of the body of HMC.getSamples' loop, src/HMC.py:154-179, fused:
 * integrate -> energies -> ratio -> reject mask -> select -> store.
 *   q_in  (D,N) chain state;  p_in (D,N) freshly drawn momentum;  u_in (N) uniforms
 *   q_out (D,N) <- samples_hmc[:,:,i]   (may alias q_in)
 *   p_out (D,N) <- momentum_hmc[:,:,i]  (may alias p_in; may be NULL)
 *   ratio_out (N) optional; reject_out (N bytes, 1 = rejected) optional.
 * Parity mode: the host uploads the NumPy RandomState stream into p_in / u_in.
 */
int pbbi_hmc_iter(const pbbi_potential* pot, int method, const void* q_in, const void* p_in,
                  const void* u_in, const void* mass, void* q_out, void* p_out, void* ratio_out,
                  uint8_t* reject_out, int64_t N, int64_t ldn, double h, int L, int flags,
                  void* stream);

/* pbbi_hmc_iter for momenta drawn at kT = boltzmannConst*temperature (src/ensemble.py:88): identical
 * to it unless flags has PBBI_BETA_ACCEPT, which needs kT for its accept test. */
int pbbi_hmc_iter_kt(const pbbi_potential* pot, int method, const void* q_in, const void* p_in,
                     const void* u_in, const void* mass, void* q_out, void* p_out, void* ratio_out,
                     uint8_t* reject_out, int64_t N, int64_t ldn, double h, int L, int flags, double kT,
                     void* stream);

/* S iterations of the same loop with momentum and uniforms drawn in-kernel
 * (RNG contract above): iteration i uses draw index iter0+i, chain n uses the
 * global index chain0+n, p = sqrt(mass*kT) * z  (src/ensemble.py:88-91 with
 * kT = boltzmannConst*temperature).
 *   q_state (D,N; stride ldn): in = current positions, out = positions after S iterations
 *   samples_out (S, D, N) dense slabs <- samples_hmc   (device layout is S-major;
 *               the reference's (D,N,S) is a permuted view of it); NULL = burn-in: the S
 *               iterations only advance q_state (momenta_out must then be NULL too)
 *   momenta_out (S, D, N) or NULL;  reject_out (S, N) bytes or NULL;
 *   ratio_out (S, N) or NULL.
 * iter0 + S must not exceed 2^32: the Philox counter carries 32 iteration bits, a larger index
 * would repeat the draws of iteration (index mod 2^32) and is refused (PBBI_ERR_INVALID).  Runs
 * that must not share draws (a warm-up and the sampling that follows) use different seeds.
 * The run is the unit of work: consecutive iterations may share one launch, and the dense-Gaussian paths
 * keep the gradient of the chain's current position between iterations (the loop of src/HMC.py:150-179
 * evaluates it again at the top of every iteration, src/integrator.py:105-108).  Neither changes a
 * result: a run of S iterations equals S runs of one iteration bit for bit.
 */
int pbbi_hmc_run(const pbbi_potential* pot, int method, void* q_state, const void* mass,
                 void* samples_out, void* momenta_out, uint8_t* reject_out, void* ratio_out,
                 int64_t N, int64_t ldn, double h, int L, int S, int flags, uint64_t seed,
                 uint64_t iter0, uint64_t chain0, double kT, void* stream);

/* What pbbi_hmc_run would do with these arguments, in words, written to out (NUL-terminated, truncated to
 * out_len): the kernel family, whether the run carries the gradient between iterations and -- if not -- why
 * (e.g. the dense D = 128 path carries below N = 2^32 / (16 D) = 2 097 152 chains per call only: beyond, the
 * two carried slabs pass the 32-bit buffer offsets), and how many iterations one launch covers.  Nothing is
 * launched. */
int pbbi_describe_run(const pbbi_potential* pot, int method, int64_t N, int64_t ldn, int L, int S, int flags,
                      char* out, int out_len);

/* pbbi_hmc_iter_kt / pbbi_hmc_run with per-chain trajectory lengths (flags PBBI_PER_CHAIN_STEPS,
 * PBBI_UTURN_STOP above).  steps_in (N int32, device; iter only): the chains' step counts, each in
 * [0, L]; NULL = L for every chain.  steps_out (N, or (S, N) for run; may be NULL): the steps each
 * chain took.  Leapfrog only.  Served by the chain-per-lane kernels (harmonic, diagonal Gaussian,
 * Rosenbrock: fp64, D <= 32, reference operation order whatever PBBI_KDK_FMA says, bit-exact with the
 * oracle) and by the dense MFMA kernels (fp64, D <= 256, both flags: the 16-chain tile keeps stepping while
 * one of its chains is live; the U-turn quantity is formed from the kernel's kick-drift-kick values, so a
 * chain whose (q - q_0) . v passes zero within rounding may stop one step apart from a reference-order
 * run); with PBBI_PER_CHAIN_STEPS alone (no U-turn stop) and PBBI_KDK_FMA also by the multi-lane kernels --
 * harmonic / diagonal Gaussian 32 < D <= 256, Rosenbrock 32 < D <= 128, kick-drift-kick form, ~1e-13 from the
 * reference order like every PBBI_KDK_FMA path; other paths return PBBI_ERR_UNSUPPORTED. */
int pbbi_hmc_iter_dyn(const pbbi_potential* pot, int method, const void* q_in, const void* p_in,
                      const void* u_in, const void* mass, const int32_t* steps_in, void* q_out, void* p_out,
                      void* ratio_out, uint8_t* reject_out, int32_t* steps_out, int64_t N, int64_t ldn,
                      double h, int L, int flags, double kT, void* stream);
int pbbi_hmc_run_dyn(const pbbi_potential* pot, int method, void* q_state, const void* mass,
                     void* samples_out, void* momenta_out, uint8_t* reject_out, void* ratio_out,
                     int32_t* steps_out, int64_t N, int64_t ldn, double h, int L, int S, int flags,
                     uint64_t seed, uint64_t iter0, uint64_t chain0, double kT, void* stream);

/* ---- a reversible per-chain dynamic trajectory length: the self-tuning no-U-turn sampler ------------
 * ("no u-turn sampling" is planned in the reference, references/PhysicsBasedHMC_SoHPC2022_WeekPlan.md:16-17;
 * PBBI_UTURN_STOP alone only measures, see above.)  GIST (Bou-Rabee, Carpenter & Marsden 2024): with
 * tau(q, p) = the number of leapfrog steps until (q_j - q_0) . p_j < 0 first holds, at most Lmax,
 *     tau_f = tau(q, p);    L uniform on 1..tau_f  (1 + floor(u tau_f), u from PBBI_STREAM_STEPS);
 *     (q', p') = L steps from (q, p);    tau_b = tau(q', -p');
 *     accept with probability min(1, exp((H - H')/kT_or_1) * tau_f / tau_b * [L <= tau_b]).
 * Every chain adapts its length to where it is, and the chain still leaves exp(-H) invariant: the length is a
 * Gibbs draw whose density enters the Metropolis ratio at both ends.  One iteration is three masked
 * trajectories (forward search, proposal, backward search: the per-chain-length kernels of
 * pbbi_hmc_iter_dyn, so the same potentials / sizes are served: elementwise D <= 32, dense fp64 D <= 256, other
 * handles return PBBI_ERR_UNSUPPORTED) plus an accept kernel; momenta are drawn by pbbi_philox_normal's
 * kernel for the same counters as pbbi_hmc_run (PBBI_DRAW_F64 honoured), the Metropolis uniform is the
 * chain's PBBI_STREAM_UNIFORM draw.  Arguments as pbbi_hmc_run (samples_out required); tau_out (S, 3, N)
 * int32 or NULL receives tau_f, L, tau_b; ratio_out the full acceptance ratio.  flags: PBBI_COMPAT_P_FROM_OLDQ,
 * PBBI_BETA_ACCEPT, PBBI_DRAW_F64.  Restated in oracle/pbbi_oracle.c::oracle_hmc_iter_gist. */
int pbbi_hmc_run_gist(const pbbi_potential* pot, void* q_state, const void* mass, void* samples_out,
                      void* momenta_out, uint8_t* reject_out, void* ratio_out, int32_t* tau_out, int64_t N,
                      int64_t ldn, double h, int Lmax, int S, int flags, uint64_t seed, uint64_t iter0,
                      uint64_t chain0, double kT, void* stream);

/* ---- tempering: replica exchange between temperature rungs ----------------------------
 * The reference draws momenta at kB*T (src/ensemble.py:88) and plans canonical / micro-canonical ensembles
 * (references/PhysicsBasedHMC_SoHPC2022_WeekPlan.md:25-27; Ensemble.setWeights, src/ensemble.py:52-61, is
 * commented out).  A temperature LADDER: the ensemble's N = R*Nr chains form R rungs -- rung r = chains
 * [r*Nr, (r+1)*Nr), sampled at kT_r by pbbi_hmc_run(..., PBBI_BETA_ACCEPT, kT_r) on its block of the state --
 * and this call is the exchange step between them, on the device, one launch after the energy evaluation:
 * for r = parity, parity + 2, ... chain n of rung r and chain n of rung r+1 swap POSITIONS with probability
 *     min(1, exp((beta_r - beta_{r+1}) (U(q_r) - U(q_{r+1})))),   beta_r = 1 / kT_r,
 * which leaves prod_r exp(-beta_r U(q_r)) invariant; hot rungs cross barriers, swaps carry the crossings down
 * to the rung at kT = 1.  betas: R doubles on the DEVICE.  The uniform of a pair is the lower chain's
 * (global index chain0 + r*Nr + n) draw of PBBI_STREAM_SWAP for `iter`, u53 like the Metropolis uniform.
 * swapped_out: ((R-1), Nr) bytes or NULL (rows of the other parity are left alone). */
int pbbi_replica_exchange(const pbbi_potential* pot, void* q, int64_t Nr, int R, int64_t ldn, const double* betas,
                          int parity, uint64_t seed, uint64_t iter, uint64_t chain0, uint8_t* swapped_out,
                          void* stream);

/* ---- RNG (device Philox stream) -------------------------------------------
 * out[d*ldn+n] = scale * z(dim d, chain chain0+n); scale_per_chain (N) overrides
 * scale when non-NULL.  Device-mode counterpart of Ensemble.setPosition /
 * setMomentum (src/ensemble.py:63-93).  dtype of out/scale_per_chain = `dtype`. */
int pbbi_philox_normal(uint64_t seed, int rng_stream, uint64_t iter, uint64_t chain0, int D,
                       int64_t N, int64_t ldn, double scale, const void* scale_per_chain,
                       int dtype, int device, void* out, void* stream);
int pbbi_philox_uniform(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, int dtype,
                        int device, void* out, void* stream);
/* out[n] = 1 + floor(u * L) capped at L, u = the PBBI_STREAM_STEPS uniform of chain chain0+n (block
 * 0xFFFFFFFF, same 53-bit construction as the Metropolis uniform): what PBBI_PER_CHAIN_STEPS draws. */
int pbbi_philox_steps(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, int L, int device,
                      int32_t* out, void* stream);

/* ---- layout helper --------------------------------------------------------
 * (S, D, N) device slabs -> the reference's (D, N, S) S-fastest array
 * (src/HMC.py:136-145,178-179).  Both device pointers, same dtype. */
int pbbi_transpose_sdn_to_dns(const void* src_sdn, void* dst_dns, int S, int D, int64_t N,
                              int dtype, int device, void* stream);

/* ---- streaming statistics (the sample sink) -----------------------
 * Per-dimension mean and (biased) variance over all S*N draws of (S, D, N) device slabs,
 * without moving the samples off the GPU: mean_out[d], var_out[d] (D elements each, dtype of
 * the slabs; accumulated in fp64, two deterministic stages, no atomics).  The reference only
 * returns the (D, N, S) array (src/HMC.py:136-183); at C2 scale that is 6.7 GB per 100 draws. */
int pbbi_sample_moments(const void* samples_sdn, int S, int D, int64_t N, int dtype, int device,
                        void* mean_out, void* var_out, void* stream);

/* Per-CHAIN mean and unbiased variance over the S draws of each (dim, chain): chain_mean_out and
 * chain_var_out are (D, N) device arrays (stride N) in the slabs' dtype, accumulated in fp64
 * (Welford).  The ingredients of the Gelman-Rubin statistic across the ensemble's chains
 * (HMC.rhat in the Python layer); S >= 2. */
int pbbi_chain_moments(const void* samples_sdn, int S, int D, int64_t N, int dtype, int device,
                       void* chain_mean_out, void* chain_var_out, void* stream);

/* Ensemble-averaged autocovariance of the chains' draws at lags 0..T (T <= PBBI_MAX_LAG):
 *   acov_out[t*D + d] = mean_n (1/S) sum_{s < S-t} (x[s,d,n] - m[d,n]) (x[s+t,d,n] - m[d,n]),
 * m = chain_mean (D, N) from pbbi_chain_moments.  acov_out: (T+1, D) DOUBLES on the device (fp64
 * accumulation, deterministic two-stage reduction over the chains).  The ingredient of the effective
 * sample size of the ensemble (HMC.ess in the Python layer). */
#define PBBI_MAX_LAG 32
int pbbi_chain_autocov(const void* samples_sdn, const void* chain_mean, int S, int D, int64_t N, int T,
                       int dtype, int device, double* acov_out, void* stream);

/* Covariance matrix of all S*N draws: cov_out[i*D + j] = mean((x_i - mean_i)(x_j - mean_j)), D x D
 * DOUBLES on the device, row-major, symmetric; `mean` (D doubles on the device) is subtracted before
 * the products (give the output of pbbi_sample_moments converted to double, or any shift: the result
 * is then the second moment about that point). */
int pbbi_sample_covariance(const void* samples_sdn, int S, int D, int64_t N, int dtype, int device,
                           const double* mean, double* cov_out, void* stream);

/* ---- ensemble weights ----------------------------------------------
 * Normalised canonical weights of an ensemble from its per-chain Hamiltonians (pbbi_energy),
 *     w_n = exp(-beta (H_n - H_min)) / sum_m exp(-beta (H_m - H_min)),
 * the normalised form of HMC.getWeights (src/HMC.py:86-104; the reference's commented-out
 * Ensemble.setWeights, src/ensemble.py:52-61) -- computed on the device in three steps so that a
 * SHARDED ensemble can all-reduce the two scalars in between (MIN of *min_out, SUM of *sum_out; both
 * are device doubles):
 *   pbbi_reduce_min        *min_out = min_n x[n]  (NaNs are skipped; +inf for N == 0)
 *   pbbi_canonical_weights w_out[n] = exp(-beta (H[n] - *hmin)),  *sum_out = sum_n w_out[n]
 *   pbbi_scale_inverse     w[n] /= *sum
 * Reductions are two-stage and deterministic (no atomics).  x / H / w are of `dtype`. */
int pbbi_reduce_min(const void* x, int64_t N, int dtype, int device, double* min_out, void* stream);
int pbbi_canonical_weights(const void* H, int64_t N, double beta, const double* hmin, int dtype,
                           int device, void* w_out, double* sum_out, void* stream);
int pbbi_scale_inverse(void* w, int64_t N, const double* sum, int dtype, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PBBI_H */
