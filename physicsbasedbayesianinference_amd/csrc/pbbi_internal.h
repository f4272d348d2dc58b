// pbbi_internal.h -- shared between the translation units of libpbbi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "pbbi.h"

enum { KIND_HARMONIC = 0, KIND_GAUSS_DIAG = 1, KIND_GAUSS_DENSE = 2, KIND_ROSENBROCK = 3, KIND_CUSTOM = 4 };

// layout version of the structs a user-potential plugin (pbbi_custom.h) shares with libpbbi.so
#define PBBI_PLUGIN_ABI 4  /* 4: plugins honour IterArgs::fuse_* (several iterations of a run per call) */

struct IterArgs;
struct IntegrateArgs;
struct EvalArgs;

struct pbbi_potential {
    int kind;
    int D;
    int dtype;
    int device;
    double cst, a, b, s;
    void* d_mean;  // D elements (dtype) or nullptr
    void* d_prec;  // D elements (harmonic / diag) or D*D row-major (dense)
    // dense Gaussian, register-resident MFMA path (D <= 128):
    int DP;        // D padded to a multiple of 16 (0 = path not available)
    void* d_frag;  // DP*DP elements: precision in MFMA A-fragment order
    void* d_mean_pad;  // DP elements, zero padded
    bool zero_mean;    // every mean entry is exactly 0 (x = q, no subtraction needed)
    // dense Gaussian, streaming GEMM path (D > 128, or fp32): kernels_big.hip
    int DPAD_big;      // D padded to a multiple of 128 (0 = path not built)
    void* d_big_PT;    // DPAD x DPAD, P transposed ([k][i]), zero padded, handle dtype
    void* d_big_mu;    // DPAD, zero padded
    // user-defined potential (KIND_CUSTOM): kernels live in a plugin .so built by custom.py
    void* plugin;      // dlopen handle
    void* d_params;    // n_params elements (dtype) handed to the user's functions, or nullptr
    int n_params;
    int (*plugin_hmc_iter)(const IterArgs*);
    int (*plugin_integrate)(const IntegrateArgs*);
    int (*plugin_eval)(const EvalArgs*, int mode);
    // dense Gaussian, the same kernel with P streamed through the LDS (128 < D <= 256, fp64): kernels_dstream.hip
    int DPS;           // D padded to 192 or 256 (0 = path not available)
    void* d_sfrag;     // DPS*DPS elements: precision in the order the stream consumes it
    void* d_smean;     // DPS elements, zero padded
};

// ---- error plumbing ---------------------------------------------------------
void pbbi_set_error(const std::string& msg);
int pbbi_fail(int code, const std::string& msg);

#define PBBI_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return pbbi_fail(PBBI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct DeviceGuard {  // make the handle's device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// PBBI_BETA_ACCEPT: factor on (oldH - newH) in the accept test.  1.0 (exact: the product then has the
// bits of the difference) reproduces src/HMC.py:115; 1/kT is the test that matches the momentum draw
// of src/ensemble.py:88.
__host__ __device__ inline double pbbi_accept_beta(int flags, double kT) {
    return (flags & PBBI_BETA_ACCEPT) ? 1.0 / kT : 1.0;
}

int pbbi_num_cus(int device);  // compute units of a device (cached), pbbi_api.hip
// ---- launch descriptors passed between api and kernel TUs ---------------------
struct IterArgs {
    const pbbi_potential* pot;
    int method;
    const void* q_in;
    const void* p_in;  // nullptr in RNG mode
    const void* u_in;  // nullptr in RNG mode
    const void* mass;
    void* q_out;
    void* p_out;
    void* ratio_out;
    uint8_t* reject_out;
    int64_t N, ldn_in, ldn_out;
    double h;
    int L;
    int flags;
    // RNG mode
    int rng;
    uint64_t seed, iter, chain0;
    double kT;
    hipStream_t stream;
    // scratch arena lent by pbbi_hmc_run (all its iterations run on one stream, one after the other):
    // paths that need device scratch carve it from here instead of allocating every iteration, and
    // report the bytes they asked for through scratch_used (how the first iteration sizes the arena)
    void* scratch;
    size_t scratch_bytes;
    size_t* scratch_used;
    // Fused run (pbbi_hmc_run on a path that keeps the chain on chip across iterations, see
    // pbbi_fused_iterations): fuse_S > 1 makes this ONE call cover draw indices iter .. iter+fuse_S-1.
    // Iteration k of the call writes position slab fuse_slab0 + k of fuse_q_base (D*N elements each,
    // stride N; modulo 2 when fuse_wrap2: a burn-in's two scratch slabs), p_out + k*D*N,
    // ratio_out + k*N, reject_out + k*N; q_in / ldn_in feed iteration 0 only.
    int fuse_S;
    int fuse_wrap2;
    int64_t fuse_slab0;
    void* fuse_q_base;
    // per-chain trajectory lengths (PBBI_PER_CHAIN_STEPS / PBBI_UTURN_STOP, pbbi_hmc_*_dyn)
    const int32_t* steps_in;  // uploaded mode; nullptr = L
    int32_t* steps_out;       // optional
    // gradient carried across the iterations of a run (dense MFMA kernel, kernels_dense.hip CARRY):
    // carry = 1 first iteration (forms and stores g(q_0)), 2 later ones; carry_g = [2][D][N] doubles,
    // carry_sel = [N] bytes, zeroed before iteration 0
    int carry;
    void* carry_g;
    uint8_t* carry_sel;
    // routing decisions a run takes ONCE for all its iterations (pbbi_hmc_run): PBBI_ROUTE_* bits
    int route_hint;
};
#define PBBI_ROUTE_NO_DENSE_STREAM 1  /* 128 < D <= 256 dense Gaussian: stay on the GEMM path (kernels_big.hip) */
inline bool pbbi_dyn(const IterArgs& a) { return (a.flags & (PBBI_PER_CHAIN_STEPS | PBBI_UTURN_STOP)) != 0; }

struct IntegrateArgs {
    const pbbi_potential* pot;
    int method;
    void* q;
    void* p;
    const void* mass;
    void* v_out;
    int64_t N, ldn;
    double h;
    int L;
    hipStream_t stream;
};

struct EvalArgs {
    const pbbi_potential* pot;
    const void* q;
    const void* p;     // energy only
    const void* mass;  // energy only
    int64_t N, ldn;
    void* U_out;     // potential (eval) or H (energy)
    void* grad_out;  // eval only
    void* w_out;     // energy only: exp(-H)
    int ratio_finish;  // energy only: U_out[n] = exp(U_out[n] - H)   (src/HMC.py:115)
    hipStream_t stream;
};

// Device scratch for one call: from the arena an IterArgs lends, else stream-ordered allocations.
#include <vector>
struct Scratch {
    hipStream_t st;
    char* base = nullptr;
    size_t cap = 0, off = 0, total = 0;
    size_t* used = nullptr;
    std::vector<void*> owned;
    explicit Scratch(hipStream_t s) : st(s) {}
    explicit Scratch(const IterArgs& a)
        : st(a.stream), base((char*)a.scratch), cap(a.scratch ? a.scratch_bytes : 0), used(a.scratch_used) {}
    void* get(size_t bytes) {
        bytes = ((bytes ? bytes : 16) + 255) / 256 * 256;
        total += bytes;
        if (base && off + bytes <= cap) {
            void* p = base + off;
            off += bytes;
            return p;
        }
        void* p = nullptr;
        if (hipMallocAsync(&p, bytes, st) != hipSuccess) return nullptr;
        owned.push_back(p);
        return p;
    }
    ~Scratch() {
        for (void* p : owned) (void)hipFreeAsync(p, st);
        if (used) *used = total;
    }
};

// chain-per-lane kernels (harmonic / diagonal Gaussian / Rosenbrock), kernels_lane.hip
int lane_hmc_iter(const IterArgs& a);
int lane_integrate(const IntegrateArgs& a);
int lane_eval(const EvalArgs& a);
int lane_energy(const EvalArgs& a);
// Rosenbrock 16 < D <= 32, Leapfrog: two lanes per chain (config C3), kernels_lane2.hip
bool lane2_applies(const IterArgs& a);
int lane2_hmc_iter(const IterArgs& a);
// how many consecutive iterations of pbbi_hmc_run one call of route_hmc may cover for these arguments
// (1 = the path has no fused form); lane_fused_iterations: kernels_lane.hip
int lane_fused_iterations(const IterArgs& a);
const char* lane_route_name(const IterArgs& a);  // the kernel family lane_hmc_iter picks (pbbi_describe_run)
// per-chain trajectory lengths (k_lane_dyn_hmc: elementwise potentials, fp64, D <= 32, Leapfrog)
int lane_dyn_hmc_iter(const IterArgs& a);
// one GIST iteration fused into one launch (k_lane_gist_hmc; pbbi_hmc_run_gist): elementwise potentials, fp64, D <= 32
bool lane_gist_applies(const pbbi_potential* pot);
int lane_gist_iter(const IterArgs& a);
// harmonic / diagonal Gaussian, 16 < D <= 256, PBBI_KDK_FMA: 16-dim parts in the waves of a workgroup, kernels_sepn.hip
bool sepn_applies(const IterArgs& a);
int sepn_hmc_iter(const IterArgs& a);
// ... and in the reference's operation order (bit-exact), PBBI_KDK_FMA not set
bool sepx_applies(const IterArgs& a);
int sepx_hmc_iter(const IterArgs& a);
// Rosenbrock, 32 < D <= 256, PBBI_KDK_FMA, same layout with boundary exchange through LDS, kernels_rosn.hip
bool rosn_applies(const IterArgs& a);
int rosn_hmc_iter(const IterArgs& a);
// Rosenbrock, 32 < D <= 64 (128), PBBI_KDK_FMA: 4 (8) lanes of one wave per chain, kernels_rosg.hip
bool rosg_applies(const IterArgs& a);
int rosg_hmc_iter(const IterArgs& a);
// ... and in the reference's operation order (bit-exact), 32 < D <= 128, PBBI_KDK_FMA not set
bool rosgx_applies(const IterArgs& a);
int rosgx_hmc_iter(const IterArgs& a);
// the same potentials for D > 64 and for fp32: chain state in a device workspace, kernels_stream.hip
int stream_hmc_iter(const IterArgs& a);
int stream_integrate(const IntegrateArgs& a);
int stream_eval(const EvalArgs& a);
int stream_energy(const EvalArgs& a);
// dense-precision Gaussian, MFMA register-resident (D <= 128), kernels_dense.hip
#define PBBI_CARRY_MAX_BYTES (((uint64_t)1 << 32) - ((uint64_t)1 << 20))  /* carried dense slabs: 2 * DP * N * 8 below this */
int dense_hmc_iter(const IterArgs& a);
bool dense_carry_applies(const IterArgs& a);  // may a run on these arguments carry the gradient?
int dense_fused_iterations(const IterArgs& a);
bool big_carry_applies(const IterArgs& a);     // the same for the GEMM path (kernels_big.hip)
void big_carry_bytes(const pbbi_potential* pot, int64_t N, size_t* g_bytes, size_t* sel_bytes);  // iterations one launch may cover (carried runs)
int dense_integrate(const IntegrateArgs& a);
int dense_eval(const EvalArgs& a);
int dense_energy(const EvalArgs& a);
int dense_build_fragments(pbbi_potential* pot, const double* precision_host, const double* mean_host);
// dense-precision Gaussian, 128 < D <= 256, fp64: the register-resident kernel with P streamed, kernels_dstream.hip
int dense_stream_build(pbbi_potential* pot, const double* precision_host, const double* mean_host);
bool dense_stream_applies(const IterArgs& a);
bool dense_stream_carry_applies(const IterArgs& a);
int dense_stream_fused_iterations(const IterArgs& a);
int dense_stream_hmc_iter(const IterArgs& a);
bool dense_stream_integrate_applies(const IntegrateArgs& a);
int dense_stream_integrate(const IntegrateArgs& a);
// dense-precision Gaussian, streaming MFMA GEMM per step (D > 128, fp64 / fp32), kernels_big.hip
int big_hmc_iter(const IterArgs& a);
int big_integrate(const IntegrateArgs& a);
int big_eval(const EvalArgs& a);
int big_energy(const EvalArgs& a);
int big_build(pbbi_potential* pot, const double* precision_host, const double* mean_host);
