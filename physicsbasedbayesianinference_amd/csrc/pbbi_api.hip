// pbbi_api.hip -- the extern "C" surface declared in include/pbbi.h: handle management,
// argument checking and dispatch to the kernel translation units.  No arithmetic of the
// hot path lives here and nothing falls back to the host.
#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include <type_traits>

#include "pbbi_internal.h"
#include "pbbi_rng.h"

// ------------------------------------------------------------------- errors
static thread_local std::string g_last_error;

void pbbi_set_error(const std::string& msg) { g_last_error = msg; }
int pbbi_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int pbbi_num_cus(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0)
        cus = 256;
    if (device >= 0 && device < 64) cached[device] = cus;
    return cus;
}

namespace {

inline size_t elem_size(int dtype) { return dtype == PBBI_F64 ? 8 : 4; }

// host double -> device array of dtype
int upload(const double* host, size_t n, int dtype, void** dev_out) {
    *dev_out = nullptr;
    if (n == 0) return PBBI_OK;
    PBBI_HIP(hipMalloc(dev_out, n * elem_size(dtype)));
    if (dtype == PBBI_F64) {
        PBBI_HIP(hipMemcpy(*dev_out, host, n * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = (float)host[i];
        PBBI_HIP(hipMemcpy(*dev_out, tmp.data(), n * 4, hipMemcpyHostToDevice));
    }
    return PBBI_OK;
}

// The workspace-based paths (kernels_big / kernels_stream / user plugins) take their scratch from
// the device's default stream-ordered pool (hipMallocAsync).  With the default release threshold
// (0) the pool hands its memory back to the OS at every synchronisation and the next call's
// allocations are real ones again; keep freed blocks in the pool instead (once per device).
void keep_pool_memory(int device) {
    static bool done[64];
    if (device < 0 || device >= 64 || done[device]) return;
    done[device] = true;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, device) != hipSuccess) return;
    uint64_t threshold = UINT64_MAX;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
}

int new_handle(int kind, int D, int dtype, int device, pbbi_potential** out) {
    if (!out) return pbbi_fail(PBBI_ERR_INVALID, "out pointer is NULL");
    *out = nullptr;
    if (D < 1) return pbbi_fail(PBBI_ERR_INVALID, "D must be >= 1");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    int count = 0;
    PBBI_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return pbbi_fail(PBBI_ERR_INVALID, "device " + std::to_string(device) + " out of range (" +
                                               std::to_string(count) + " HIP devices visible)");
    keep_pool_memory(device);
    pbbi_potential* p = new (std::nothrow) pbbi_potential();
    if (!p) return pbbi_fail(PBBI_ERR_INVALID, "out of host memory");
    p->kind = kind; p->D = D; p->dtype = dtype; p->device = device;
    p->cst = 0.0; p->a = 1.0; p->b = 100.0; p->s = 20.0;
    p->d_mean = p->d_prec = p->d_frag = p->d_mean_pad = p->d_big_PT = p->d_big_mu = nullptr;
    p->DP = 0;
    p->DPAD_big = 0;
    p->DPS = 0;
    p->d_sfrag = p->d_smean = nullptr;
    p->plugin = p->d_params = nullptr;
    p->n_params = 0;
    p->plugin_hmc_iter = nullptr; p->plugin_integrate = nullptr; p->plugin_eval = nullptr;
    p->zero_mean = true;
    *out = p;
    return PBBI_OK;
}

int finish_or_destroy(int rc, pbbi_potential** out) {
    if (rc != PBBI_OK && out && *out) {
        std::string keep = g_last_error;
        pbbi_potential_destroy(*out);
        *out = nullptr;
        g_last_error = keep;
    }
    return rc;
}

// D-element parameter vectors are stored zero-padded to a multiple of 64 entries: the register
// kernels read rows d >= D of a padded chain with precision 0 / mean 0 instead of guarding them
inline size_t padded64(int D) { return ((size_t)D + 63) / 64 * 64; }

int upload_padded(const double* host, int D, int dtype, void** dev_out) {
    std::vector<double> tmp(padded64(D), 0.0);
    if (host) std::memcpy(tmp.data(), host, sizeof(double) * D);
    return upload(tmp.data(), tmp.size(), dtype, dev_out);
}

int upload_mean(pbbi_potential* p, const double* mean) { return upload_padded(mean, p->D, p->dtype, &p->d_mean); }

int check_common(const pbbi_potential* pot, int64_t N, int64_t ldn) {
    if (!pot) return pbbi_fail(PBBI_ERR_INVALID, "potential handle is NULL");
    if (N < 0) return pbbi_fail(PBBI_ERR_INVALID, "N must be >= 0");
    if (ldn < N) return pbbi_fail(PBBI_ERR_INVALID, "leading stride ldn must be >= N");
    return PBBI_OK;
}

inline bool is_dense(const pbbi_potential* pot) {  // register-resident MFMA path
    return pot->kind == KIND_GAUSS_DENSE && pot->DP != 0;
}
inline bool is_big(const pbbi_potential* pot) {  // streaming GEMM path
    return pot->kind == KIND_GAUSS_DENSE && pot->DP == 0;
}

// ---- routing: which kernel family serves a handle -----------------------------------------------
int plugin_rc(int rc) {
    if (rc == 0) return PBBI_OK;
    if (rc < 0) return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the chain workspace");
    return pbbi_fail(PBBI_ERR_HIP, std::string("user-potential kernel launch: ") +
                                       hipGetErrorString((hipError_t)rc));
}
int route_hmc(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (pbbi_dyn(a) && (pot->kind == KIND_CUSTOM || (is_big(pot) && !dense_stream_applies(a))))
        return pbbi_fail(PBBI_ERR_UNSUPPORTED, "per-chain trajectory lengths are served by the chain-per-lane "
                                               "kernels (D <= 32) and the dense kernels (fp64, D <= 256) only");
    if (pot->kind == KIND_CUSTOM) return a.N ? plugin_rc(pot->plugin_hmc_iter(&a)) : PBBI_OK;
    if (is_big(pot) && dense_stream_applies(a)) return dense_stream_hmc_iter(a);
    return is_big(pot) ? big_hmc_iter(a) : is_dense(pot) ? dense_hmc_iter(a) : lane_hmc_iter(a);
}
// consecutive iterations of pbbi_hmc_run that ONE route_hmc call may cover (IterArgs::fuse_*)
int route_fused_iterations(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (is_big(pot) && pot->kind != KIND_CUSTOM) return dense_stream_applies(a) ? dense_stream_fused_iterations(a) : 1;
    if (pot->kind == KIND_CUSTOM) {
        // plugins (PBBI_PLUGIN_ABI >= 4) take the iterations of a run several at a time: their register
        // kernels keep the chain and its potential energy on chip, the workspace kernels unroll the call
        static const int fuse = getenv("PBBI_FUSE_ITERS") ? atoi(getenv("PBBI_FUSE_ITERS")) : 16;
        return (fuse > 1 && a.rng && a.N > 0 && !pbbi_dyn(a)) ? fuse : 1;
    }
    if (is_dense(pot)) return dense_fused_iterations(a);
    return lane_fused_iterations(a);
}
int route_integrate(const IntegrateArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (pot->kind == KIND_CUSTOM) return a.N ? plugin_rc(pot->plugin_integrate(&a)) : PBBI_OK;
    if (is_big(pot) && dense_stream_integrate_applies(a)) return dense_stream_integrate(a);
    return is_big(pot) ? big_integrate(a) : is_dense(pot) ? dense_integrate(a) : lane_integrate(a);
}
int route_eval(const EvalArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (pot->kind == KIND_CUSTOM) return a.N ? plugin_rc(pot->plugin_eval(&a, 0)) : PBBI_OK;
    return is_big(pot) ? big_eval(a) : is_dense(pot) ? dense_eval(a) : lane_eval(a);
}
int route_energy(const EvalArgs& a) {
    const pbbi_potential* pot = a.pot;
    if (pot->kind == KIND_CUSTOM)
        return a.N ? plugin_rc(pot->plugin_eval(&a, a.ratio_finish ? 2 : 1)) : PBBI_OK;
    return is_big(pot) ? big_energy(a) : is_dense(pot) ? dense_energy(a) : lane_energy(a);
}

// ---- small utility kernels ---------------------------------------------------
template <typename T>
__global__ void k_philox_normal(uint64_t seed, int stream, uint64_t iter, uint64_t chain0, int D,
                                int64_t N, int64_t ldn, double scale, const T* scale_n, T* out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double sc = scale_n ? (double)scale_n[n] : scale;
    for (int d = 0; d < D; ++d)
        out[(int64_t)d * ldn + n] = (T)(rng_normal(seed, (uint32_t)stream & 0xFFu, iter, chain0 + n, d,
                                                   (stream & PBBI_STREAM_DRAW_F64) != 0) * sc);
}

template <typename T>
__global__ void k_philox_uniform(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, T* out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = (T)rng_uniform(seed, iter, chain0 + n);
}

__global__ void k_fill_i32(int32_t* out, int64_t N, int value) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = value;
}

__global__ void k_philox_steps(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, int L, int32_t* out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = L > 0 ? rng_steps(seed, iter, chain0 + n, L) : 0;
}

// ---- GIST (self-tuning no-U-turn) iteration pieces, pbbi_hmc_run_gist -------------------------------
// L = 1 + floor(u * tau_f), capped at tau_f: uniform on 1..tau_f, u = the chain's PBBI_STREAM_STEPS uniform
__global__ void k_gist_length(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, const int32_t* tau_f,
                              int32_t* L_out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const PhiloxOut x = rng_block(seed, /*PBBI_STREAM_STEPS*/ 3u, iter, chain0 + (uint64_t)n, 0xFFFFFFFFu);
    const int t = tau_f[n];
    const int L = 1 + (int)(u53(x.x0, x.x1) * (double)t);
    L_out[n] = L > t ? t : L;
}

template <typename T>
__global__ void k_negate(const T* in, T* out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = -in[i];
}

template <typename T>
__global__ void k_fill_zero(T* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = T(0);
}

template <typename T>
__global__ void k_pstd(const T* mass, double kT, int64_t N, T* out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = (T)sqrt((double)mass[n] * kT);   // src/ensemble.py:88
}

// accept with min(1, exp(beta (H - H')) tau_f / tau_b [L <= tau_b]); rejected chains keep their position and
// report the old position (compat, src/HMC.py:176) or the drawn momentum as momentum
template <typename T>
__global__ void k_gist_accept(const T* q_prev, int64_t ld_prev, const T* p_draw, const T* qB, const T* pB,
                              const T* ratioB, const int32_t* tau_f, const int32_t* Ls, const int32_t* tau_b,
                              uint64_t seed, uint64_t iter, uint64_t chain0, int D, int64_t N, int flags, T* q_out,
                              T* p_out, T* ratio_out, uint8_t* reject_out, int32_t* tau_out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double hr = (double)ratioB[n];
    const int tf = tau_f[n], L = Ls[n], tb = tau_b[n];
    const double ratio = (L <= tb) ? hr * ((double)tf / (double)tb) : 0.0;
    const double u = rng_uniform(seed, iter, chain0 + (uint64_t)n);
    const bool reject = (ratio == ratio) && (u > (ratio < 1.0 ? ratio : 1.0));
    const bool compat = (flags & PBBI_COMPAT_P_FROM_OLDQ) != 0;
    for (int d = 0; d < D; ++d) {
        const T qo = q_prev[(int64_t)d * ld_prev + n];
        q_out[(int64_t)d * N + n] = reject ? qo : qB[(int64_t)d * N + n];
        if (p_out) p_out[(int64_t)d * N + n] = reject ? (compat ? qo : p_draw[(int64_t)d * N + n]) : pB[(int64_t)d * N + n];
    }
    if (ratio_out) ratio_out[n] = (T)ratio;
    if (reject_out) reject_out[n] = reject ? 1 : 0;
    if (tau_out) { tau_out[n] = tf; tau_out[N + n] = L; tau_out[2 * N + n] = tb; }
}

// ---- replica exchange between the temperature rungs of an ensemble (pbbi_replica_exchange) ----------------
// rung r = chains [r*Nr, (r+1)*Nr).  Pair (r, r+1), r = parity, parity+2, ...: chain n of the one with chain n of
// the other swap POSITIONS with probability min(1, exp((beta_r - beta_{r+1}) (U_r - U_{r+1}))) -- the Metropolis
// test that leaves prod_r exp(-beta_r U(q_r)) invariant.  One thread per (pair, n); its uniform is the lower
// chain's draw of PBBI_STREAM_SWAP.
template <typename T>
__global__ void k_replica_exchange(T* q, int64_t ldn, const T* U, const double* betas, int D, int64_t Nr, int R,
                                   int parity, uint64_t seed, uint64_t iter, uint64_t chain0, uint8_t* swapped) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = parity + 2 * (int)blockIdx.y;
    if (n >= Nr || r + 1 >= R) return;
    const int64_t lo = (int64_t)r * Nr + n, hi = lo + Nr;
    const double a = (betas[r] - betas[r + 1]) * ((double)U[lo] - (double)U[hi]);
    const PhiloxOut x = rng_block(seed, /*PBBI_STREAM_SWAP*/ 4u, iter, chain0 + (uint64_t)lo, 0xFFFFFFFFu);
    const double u = u53(x.x0, x.x1);
    const bool swap = u < exp(a);   // NaN energies: no swap
    if (swapped) swapped[(int64_t)r * Nr + n] = swap ? 1 : 0;
    if (swap)
        for (int d = 0; d < D; ++d) {
            const T t = q[(int64_t)d * ldn + lo];
            q[(int64_t)d * ldn + lo] = q[(int64_t)d * ldn + hi];
            q[(int64_t)d * ldn + hi] = t;
        }
}

// (S, D*N) -> (D*N, S) tiled transpose through LDS: both sides coalesced.
template <typename T>
__global__ void k_transpose(const T* __restrict__ src, T* __restrict__ dst, int S, int64_t M) {
    __shared__ T tile[32][33];
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    const int s0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int s = s0 + j;
        const int64_t m = m0 + threadIdx.x;
        if (s < S && m < M) tile[j][threadIdx.x] = src[(int64_t)s * M + m];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t m = m0 + j;
        const int s = s0 + threadIdx.x;
        if (s < S && m < M) dst[m * S + s] = tile[threadIdx.x][j];
    }
}

// moments stage 1: block (d, chunk) sums x and x^2 over its slice of the S*N draws of row d
template <typename T>
__global__ void k_moments_partial(const T* __restrict__ x, int S, int D, int64_t N, int chunks,
                                  double* __restrict__ part /* [D][chunks][2] */) {
    const int d = blockIdx.y, ch = blockIdx.x;
    const int64_t total = (int64_t)S * N;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t lo = (int64_t)ch * per, hi = lo + per < total ? lo + per : total;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int64_t sidx = i / N, n = i - sidx * N;
        const double v = (double)x[(sidx * D + d) * N + n];
        s1 += v;
        s2 += v * v;
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            r1[threadIdx.x] += r1[threadIdx.x + w];
            r2[threadIdx.x] += r2[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[((size_t)d * chunks + ch) * 2 + 0] = r1[0];
        part[((size_t)d * chunks + ch) * 2 + 1] = r2[0];
    }
}

template <typename T>
__global__ void k_moments_final(const double* __restrict__ part, int D, int chunks, double count,
                                T* mean_out, T* var_out) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < chunks; ++c) {
        s1 += part[((size_t)d * chunks + c) * 2 + 0];
        s2 += part[((size_t)d * chunks + c) * 2 + 1];
    }
    const double mean = s1 / count;
    if (mean_out) mean_out[d] = (T)mean;
    if (var_out) var_out[d] = (T)(s2 / count - mean * mean);
}

// per-(dim, chain) mean and unbiased variance over the S slabs; lanes run along the chain axis
template <typename T>
__global__ void k_chain_moments(const T* __restrict__ x, int S, int D, int64_t N, T* mean_out,
                                T* var_out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (n >= N) return;
    double mean = 0.0, m2 = 0.0;
    for (int s = 0; s < S; ++s) {  // Welford
        const double v = (double)x[((int64_t)s * D + d) * N + n];
        const double delta = v - mean;
        mean += delta / (double)(s + 1);
        m2 += delta * (v - mean);
    }
    if (mean_out) mean_out[(int64_t)d * N + n] = (T)mean;
    if (var_out) var_out[(int64_t)d * N + n] = (T)(m2 / (double)(S - 1));
}

// lagged products of one (dim, chain) series through a register window: acc[t] += x_s * x_{s-t}
template <typename T>
__global__ void __launch_bounds__(256) k_chain_autocov(const T* __restrict__ x, const T* __restrict__ mean,
                                                       int S, int D, int64_t N, double* __restrict__ part) {
    constexpr int TL = PBBI_MAX_LAG + 1;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    double acc[TL], w[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) { acc[t] = 0.0; w[t] = 0.0; }
    if (n < N) {
        const double m = (double)mean[(int64_t)d * N + n];
        for (int s = 0; s < S; ++s) {
            const double v = (double)x[((int64_t)s * D + d) * N + n] - m;
#pragma unroll
            for (int t = TL - 1; t > 0; --t) w[t] = w[t - 1];
            w[0] = v;
#pragma unroll
            for (int t = 0; t < TL; ++t) acc[t] = fma(v, w[t], acc[t]);
        }
    }
    // sum over the block's chains, lag by lag (fixed order: deterministic)
    __shared__ double r[256];
    for (int t = 0; t < TL; ++t) {
        r[threadIdx.x] = acc[t];
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) r[threadIdx.x] += r[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[((size_t)blockIdx.x * D + d) * TL + t] = r[0];
        __syncthreads();
    }
}

__global__ void k_autocov_final(const double* __restrict__ part, int n_blocks, int D, int T, double scale,
                                double* __restrict__ out) {
    constexpr int TL = PBBI_MAX_LAG + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // t*D + d
    if (i >= (T + 1) * D) return;
    const int t = i / D, d = i - t * D;
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += part[((size_t)b * D + d) * TL + t];
    out[i] = s * scale;
}

// one 16x16 tile of sum_m (x_i - mean_i)(x_j - mean_j) over a chunk of the S*N draws (tiles on and above
// the diagonal only); partial tiles are summed in a fixed order by k_cov_final
template <typename T>
__global__ void __launch_bounds__(256) k_cov_partial(const T* __restrict__ x, const double* __restrict__ mean,
                                                     int S, int D, int64_t N, int chunks,
                                                     double* __restrict__ part /* [chunks][D][D] */) {
    const int ti = blockIdx.y, tj = blockIdx.z;
    if (tj < ti) return;
    const int li = threadIdx.x >> 4, lj = threadIdx.x & 15;
    const int64_t M = (int64_t)S * N;
    const int64_t per = ((M + chunks - 1) / chunks + 63) / 64 * 64;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
    __shared__ double xi[16][65], xj[16][65];
    double acc = 0.0;
    for (int64_t m0 = lo; m0 < hi; m0 += 64) {
        for (int k = threadIdx.x; k < 16 * 64; k += 256) {  // rows of the two tiles, 64 draws each
            const int row = k >> 6, c = k & 63;
            const int64_t m = m0 + c;
            double a = 0.0, b = 0.0;
            if (m < hi) {
                const int64_t s = m / N, n = m - s * N;
                const int di = ti * 16 + row, dj = tj * 16 + row;
                if (di < D) a = (double)x[(s * D + di) * N + n] - mean[di];
                if (dj < D) b = (double)x[(s * D + dj) * N + n] - mean[dj];
            }
            xi[row][c] = a;
            xj[row][c] = b;
        }
        __syncthreads();
#pragma unroll 16
        for (int c = 0; c < 64; ++c) acc = fma(xi[li][c], xj[lj][c], acc);
        __syncthreads();
    }
    const int i = ti * 16 + li, j = tj * 16 + lj;
    if (i < D && j < D) part[((size_t)blockIdx.x * D + i) * D + j] = acc;
}

__global__ void k_cov_final(const double* __restrict__ part, int chunks, int D, double inv_count,
                            double* __restrict__ cov) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= D * D) return;
    int i = idx / D, j = idx - i * D;
    if ((j >> 4) < (i >> 4)) { const int t = i; i = j; j = t; }  // below-diagonal tiles mirror the upper ones
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += part[((size_t)c * D + i) * D + j];
    cov[idx] = s * inv_count;
}

// ensemble weights: stage 1 of a deterministic two-stage reduction (MODE 0: min of x, NaNs skipped;
// MODE 1: w = exp(-beta (H - hmin)) stored, partial sums of w)
template <typename T, int MODE>
__global__ void k_weights_partial(const T* __restrict__ x, int64_t N, double beta, const double* hmin,
                                  T* w_out, double* __restrict__ part) {
    const double h0 = MODE == 1 ? *hmin : 0.0;
    double acc = MODE == 0 ? INFINITY : 0.0;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)x[n];
        if constexpr (MODE == 0) {
            if (v < acc) acc = v;  // false for NaN
        } else {
            const double w = exp(-beta * (v - h0));
            w_out[n] = (T)w;
            acc += (double)(T)w;   // the sum of what is stored
        }
    }
    __shared__ double r[256];
    r[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double o = r[threadIdx.x + s];
            if constexpr (MODE == 0) { if (o < r[threadIdx.x]) r[threadIdx.x] = o; }
            else r[threadIdx.x] += o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = r[0];
}

template <int MODE>
__global__ void k_weights_final(const double* __restrict__ part, int n_part, double* out) {
    double acc = MODE == 0 ? INFINITY : 0.0;
    for (int i = 0; i < n_part; ++i) {  // fixed order
        if constexpr (MODE == 0) { if (part[i] < acc) acc = part[i]; }
        else acc += part[i];
    }
    *out = acc;
}

template <typename T>
__global__ void k_scale_inverse(T* w, int64_t N, const double* sum) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) w[n] = (T)((double)w[n] / *sum);
}

template <int MODE>
int weights_reduce(const void* x, int64_t N, double beta, const double* hmin, int dtype, int device,
                   void* w_out, double* out, hipStream_t st) {
    if (N < 0) return pbbi_fail(PBBI_ERR_INVALID, "bad N");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    if (!out || (N > 0 && !x) || (MODE == 1 && (!hmin || (N > 0 && !w_out))))
        return pbbi_fail(PBBI_ERR_INVALID, "a pointer is NULL");
    DeviceGuard guard(device);
    const int n_part = (int)((N + 255) / 256 < 256 ? (N + 255) / 256 : 256);
    double* part = nullptr;
    PBBI_HIP(hipMallocAsync((void**)&part, sizeof(double) * (n_part > 0 ? n_part : 1), st));
    if (n_part > 0) {
        if (dtype == PBBI_F64)
            hipLaunchKernelGGL((k_weights_partial<double, MODE>), dim3(n_part), dim3(256), 0, st,
                               (const double*)x, N, beta, hmin, (double*)w_out, part);
        else
            hipLaunchKernelGGL((k_weights_partial<float, MODE>), dim3(n_part), dim3(256), 0, st,
                               (const float*)x, N, beta, hmin, (float*)w_out, part);
    }
    hipLaunchKernelGGL((k_weights_final<MODE>), dim3(1), dim3(1), 0, st, (const double*)part, n_part, out);
    PBBI_HIP(hipGetLastError());
    PBBI_HIP(hipFreeAsync(part, st));
    return PBBI_OK;
}

}  // namespace

// ======================================================================= library
// ---- one GIST run (pbbi_hmc_run_gist below): per iteration three masked trajectories and an accept kernel
template <typename T>
static int gist_run(const pbbi_potential* pot, void* q_state, const void* mass, void* samples_out, void* momenta_out,
                    uint8_t* reject_out, void* ratio_out, int32_t* tau_out, int64_t N, int64_t ldn, double h,
                    int Lmax, int S, int flags, uint64_t seed, uint64_t iter0, uint64_t chain0, double kT,
                    hipStream_t st) {
    const int D = pot->D;
    const size_t slab = (size_t)D * (size_t)N;
    if constexpr (std::is_same<T, double>::value) {
        if (lane_gist_applies(pot)) {   // elementwise potential, D <= 32: ONE launch per iteration (k_lane_gist_hmc)
            Scratch ws0(st);
            double* q0 = (double*)ws0.get(slab * sizeof(double));
            if (!q0) return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the GIST workspace");
            PBBI_HIP(hipMemcpy2DAsync(q0, (size_t)N * 8, q_state, (size_t)ldn * 8, (size_t)N * 8, (size_t)D,
                                      hipMemcpyDeviceToDevice, st));
            for (int i = 0; i < S; ++i) {
                IterArgs a{};
                a.pot = pot; a.method = PBBI_LEAPFROG; a.mass = mass; a.N = N; a.h = h; a.L = Lmax; a.rng = 1;
                a.kT = kT; a.stream = st; a.ldn_in = N; a.ldn_out = N; a.flags = flags;
                a.seed = seed; a.iter = iter0 + (uint64_t)i; a.chain0 = chain0;
                a.q_in = (i == 0) ? q0 : (double*)samples_out + (size_t)(i - 1) * slab;
                a.q_out = (double*)samples_out + (size_t)i * slab;
                a.p_out = momenta_out ? (double*)momenta_out + (size_t)i * slab : nullptr;
                a.ratio_out = ratio_out ? (double*)ratio_out + (size_t)i * N : nullptr;
                a.reject_out = reject_out ? reject_out + (size_t)i * N : nullptr;
                a.steps_out = tau_out ? tau_out + (size_t)i * 3 * N : nullptr;
                if (int rc = lane_gist_iter(a)) return rc;
            }
            PBBI_HIP(hipMemcpy2DAsync(q_state, (size_t)ldn * 8, (const double*)samples_out + (size_t)(S - 1) * slab,
                                      (size_t)N * 8, (size_t)N * 8, (size_t)D, hipMemcpyDeviceToDevice, st));
            return PBBI_OK;
        }
    }
    Scratch ws(st);
    T* p_draw = (T*)ws.get(slab * sizeof(T));
    T* scrQ = (T*)ws.get(slab * sizeof(T));
    T* scrP = (T*)ws.get(slab * sizeof(T));
    T* qB = (T*)ws.get(slab * sizeof(T));
    T* pB = (T*)ws.get(slab * sizeof(T));
    T* negp = (T*)ws.get(slab * sizeof(T));
    T* zeros = (T*)ws.get((size_t)N * sizeof(T));
    T* ratioB = (T*)ws.get((size_t)N * sizeof(T));
    T* pstd = mass ? (T*)ws.get((size_t)N * sizeof(T)) : nullptr;
    int32_t* tf = (int32_t*)ws.get((size_t)N * 3 * sizeof(int32_t));
    if (!p_draw || !scrQ || !scrP || !qB || !pB || !negp || !zeros || !ratioB || (mass && !pstd) || !tf)
        return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the GIST workspace");
    T* q0 = (T*)ws.get(slab * sizeof(T));
    if (!q0) return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the GIST workspace");
    PBBI_HIP(hipMemcpy2DAsync(q0, (size_t)N * sizeof(T), q_state, (size_t)ldn * sizeof(T), (size_t)N * sizeof(T),
                              (size_t)D, hipMemcpyDeviceToDevice, st));
    int32_t *Ls = tf + N, *tb = tf + 2 * N;
    const dim3 b(256), gN((unsigned)((N + 255) / 256)), gDN((unsigned)((slab + 255) / 256));
    hipLaunchKernelGGL(k_fill_zero<T>, gN, b, 0, st, zeros, N);
    if (mass) hipLaunchKernelGGL(k_pstd<T>, gN, b, 0, st, (const T*)mass, kT, N, pstd);
    const bool f64 = (flags & PBBI_DRAW_F64) != 0;
    const int base = flags & (PBBI_BETA_ACCEPT);   // the integrations run in the reference's operation order
    int rc = PBBI_OK;
    for (int i = 0; i < S && rc == PBBI_OK; ++i) {
        const uint64_t it = iter0 + (uint64_t)i;
        // iteration i starts from the state iteration i-1 recorded (dense slabs; q0: the caller's state, repacked)
        const T* q_prev = (i == 0) ? (const T*)q0 : (const T*)samples_out + (size_t)(i - 1) * slab;
        hipLaunchKernelGGL(k_philox_normal<T>, gN, b, 0, st, seed, PBBI_STREAM_MOMENTUM | (f64 ? PBBI_STREAM_DRAW_F64 : 0),
                           it, chain0, D, N, N, sqrt(kT), (const T*)pstd, p_draw);
        IterArgs a{};
        a.pot = pot; a.method = PBBI_LEAPFROG; a.mass = mass; a.N = N; a.h = h; a.L = Lmax; a.rng = 0; a.kT = kT;
        a.stream = st; a.u_in = zeros; a.ldn_in = N; a.ldn_out = N;
        // (1) forward U-turn count tau_f = tau(q, p); the end state is not used
        a.q_in = q_prev; a.p_in = p_draw; a.q_out = scrQ; a.p_out = scrP;
        a.flags = base | PBBI_UTURN_STOP; a.steps_in = nullptr; a.steps_out = tf;
        if ((rc = route_hmc(a)) != PBBI_OK) break;
        // (2) L uniform on 1..tau_f
        hipLaunchKernelGGL(k_gist_length, gN, b, 0, st, seed, it, chain0, N, (const int32_t*)tf, Ls);
        // (3) the proposal: L_n steps from (q, p); ratioB = exp(beta (H - H')); u = 0 accepts every chain
        a.q_out = qB; a.p_out = pB; a.ratio_out = ratioB; a.flags = base | PBBI_PER_CHAIN_STEPS;
        a.steps_in = Ls; a.steps_out = nullptr;
        if ((rc = route_hmc(a)) != PBBI_OK) break;
        // (4) backward U-turn count tau_b = tau(q', -p')
        hipLaunchKernelGGL(k_negate<T>, gDN, b, 0, st, (const T*)pB, negp, (int64_t)slab);
        a.q_in = qB; a.p_in = negp; a.q_out = scrQ; a.p_out = scrP; a.ratio_out = nullptr;
        a.flags = base | PBBI_UTURN_STOP; a.steps_in = nullptr; a.steps_out = tb;
        if ((rc = route_hmc(a)) != PBBI_OK) break;
        // (5) the accept test and the record
        hipLaunchKernelGGL(k_gist_accept<T>, gN, b, 0, st, q_prev, N, (const T*)p_draw, (const T*)qB,
                           (const T*)pB, (const T*)ratioB, (const int32_t*)tf, (const int32_t*)Ls, (const int32_t*)tb,
                           seed, it, chain0, D, N, flags, (T*)samples_out + (size_t)i * slab,
                           momenta_out ? (T*)momenta_out + (size_t)i * slab : nullptr,
                           ratio_out ? (T*)ratio_out + (size_t)i * N : nullptr,
                           reject_out ? reject_out + (size_t)i * N : nullptr,
                           tau_out ? tau_out + (size_t)i * 3 * N : nullptr);
    }
    if (rc != PBBI_OK) return rc;
    PBBI_HIP(hipGetLastError());
    PBBI_HIP(hipMemcpy2DAsync(q_state, (size_t)ldn * sizeof(T), (const T*)samples_out + (size_t)(S - 1) * slab,
                              (size_t)N * sizeof(T), (size_t)N * sizeof(T), (size_t)D, hipMemcpyDeviceToDevice, st));
    return PBBI_OK;
}


extern "C" {

int pbbi_version(void) { return PBBI_VERSION; }

const char* pbbi_last_error(void) { return g_last_error.c_str(); }

int pbbi_device_count(int* count_out) {
    if (!count_out) return pbbi_fail(PBBI_ERR_INVALID, "count_out is NULL");
    *count_out = 0;
    PBBI_HIP(hipGetDeviceCount(count_out));
    return PBBI_OK;
}

int pbbi_device_info(int device, pbbi_devinfo* out) {
    if (!out) return pbbi_fail(PBBI_ERR_INVALID, "out is NULL");
    hipDeviceProp_t prop;
    PBBI_HIP(hipGetDeviceProperties(&prop, device));
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->name, prop.name, sizeof(out->name) - 1);
    std::strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    out->compute_units = prop.multiProcessorCount;
    out->hbm_bytes = (int64_t)prop.totalGlobalMem;
    out->lds_bytes_per_block = (int)prop.sharedMemPerBlock;
    out->clock_khz = prop.clockRate;
    return PBBI_OK;
}

// ==================================================================== potentials
int pbbi_potential_create_harmonic(int D, const double* springConsts, int dtype, int device,
                                   pbbi_potential** out) {
    if (!springConsts) return pbbi_fail(PBBI_ERR_INVALID, "springConsts is NULL");
    if (int rc = new_handle(KIND_HARMONIC, D, dtype, device, out)) return rc;
    DeviceGuard guard(device);
    int rc = upload_padded(springConsts, D, dtype, &(*out)->d_prec);
    if (rc == PBBI_OK) rc = upload_mean(*out, nullptr);
    return finish_or_destroy(rc, out);
}

int pbbi_potential_create_gauss_diag(int D, const double* mean, const double* prec, double cst,
                                     int dtype, int device, pbbi_potential** out) {
    if (!prec) return pbbi_fail(PBBI_ERR_INVALID, "prec is NULL");
    if (int rc = new_handle(KIND_GAUSS_DIAG, D, dtype, device, out)) return rc;
    DeviceGuard guard(device);
    (*out)->cst = cst;
    int rc = upload_padded(prec, D, dtype, &(*out)->d_prec);
    if (rc == PBBI_OK) rc = upload_mean(*out, mean);
    return finish_or_destroy(rc, out);
}

int pbbi_potential_create_gauss_dense(int D, const double* mean, const double* precision,
                                      double cst, int dtype, int device, pbbi_potential** out) {
    if (!precision) return pbbi_fail(PBBI_ERR_INVALID, "precision is NULL");
    if (int rc = new_handle(KIND_GAUSS_DENSE, D, dtype, device, out)) return rc;
    DeviceGuard guard(device);
    (*out)->cst = cst;
    int rc = upload(precision, (size_t)D * D, dtype, &(*out)->d_prec);
    if (rc == PBBI_OK) rc = upload_mean(*out, mean);
    // D <= 128 in fp64: register-resident MFMA kernels; otherwise the streaming GEMM path
    if (rc == PBBI_OK) rc = dense_build_fragments(*out, precision, mean);
    if (rc == PBBI_OK && (*out)->DP == 0) rc = big_build(*out, precision, mean);
    // 128 < D <= 256 in fp64: HMC iterations on the register-resident kernel with P streamed (kernels_dstream.hip);
    // the GEMM path above keeps integrate(), the evaluations and the per-chain-length modes of the same handle
    if (rc == PBBI_OK && (*out)->DP == 0) rc = dense_stream_build(*out, precision, mean);
    return finish_or_destroy(rc, out);
}

int pbbi_potential_create_rosenbrock(int D, double a, double b, double s, int dtype, int device,
                                     pbbi_potential** out) {
    if (!(s != 0.0)) return pbbi_fail(PBBI_ERR_INVALID, "Rosenbrock scale s must be non-zero");
    if (int rc = new_handle(KIND_ROSENBROCK, D, dtype, device, out)) return rc;
    (*out)->a = a; (*out)->b = b; (*out)->s = s;
    return PBBI_OK;
}

int pbbi_potential_create_custom(const char* plugin_path, int D, const double* params, int n_params,
                                 int dtype, int device, pbbi_potential** out) {
    if (!plugin_path) return pbbi_fail(PBBI_ERR_INVALID, "plugin_path is NULL");
    if (n_params < 0 || (n_params > 0 && !params))
        return pbbi_fail(PBBI_ERR_INVALID, "params is NULL / n_params < 0");
    if (int rc = new_handle(KIND_CUSTOM, D, dtype, device, out)) return rc;
    pbbi_potential* p = *out;
    DeviceGuard guard(device);
    p->plugin = dlopen(plugin_path, RTLD_NOW | RTLD_LOCAL);
    if (!p->plugin) {
        const char* why = dlerror();
        return finish_or_destroy(pbbi_fail(PBBI_ERR_INVALID, std::string("cannot load the potential plugin: ") +
                                                                 (why ? why : plugin_path)), out);
    }
    auto abi = (int (*)(void))dlsym(p->plugin, "pbbi_plugin_abi");
    auto pdt = (int (*)(void))dlsym(p->plugin, "pbbi_plugin_dtype");
    p->plugin_hmc_iter = (int (*)(const IterArgs*))dlsym(p->plugin, "pbbi_plugin_hmc_iter");
    p->plugin_integrate = (int (*)(const IntegrateArgs*))dlsym(p->plugin, "pbbi_plugin_integrate");
    p->plugin_eval = (int (*)(const EvalArgs*, int))dlsym(p->plugin, "pbbi_plugin_eval");
    int rc = PBBI_OK;
    if (!abi || !pdt || !p->plugin_hmc_iter || !p->plugin_integrate || !p->plugin_eval)
        rc = pbbi_fail(PBBI_ERR_INVALID, "plugin does not export the pbbi_plugin_* entry points");
    else if (abi() != PBBI_PLUGIN_ABI)
        rc = pbbi_fail(PBBI_ERR_INVALID, "plugin was built against another libpbbi (ABI " +
                                             std::to_string(abi()) + ", expected " +
                                             std::to_string(PBBI_PLUGIN_ABI) + "): rebuild it");
    else if (pdt() != dtype)
        rc = pbbi_fail(PBBI_ERR_INVALID, "plugin was built for the other dtype");
    if (rc == PBBI_OK) {
        p->n_params = n_params;
        rc = upload(params, (size_t)n_params, dtype, &p->d_params);
    }
    return finish_or_destroy(rc, out);
}

int pbbi_potential_destroy(pbbi_potential* pot) {
    if (!pot) return PBBI_OK;
    DeviceGuard guard(pot->device);
    for (void* p : {pot->d_mean, pot->d_prec, pot->d_frag, pot->d_mean_pad, pot->d_big_PT, pot->d_big_mu,
                    pot->d_sfrag, pot->d_smean, pot->d_params})
        if (p) (void)hipFree(p);
    if (pot->plugin) (void)dlclose(pot->plugin);
    delete pot;
    return PBBI_OK;
}

int pbbi_potential_dim(const pbbi_potential* pot) { return pot ? pot->D : PBBI_ERR_INVALID; }
int pbbi_potential_dtype(const pbbi_potential* pot) { return pot ? pot->dtype : PBBI_ERR_INVALID; }
int pbbi_potential_device(const pbbi_potential* pot) { return pot ? pot->device : PBBI_ERR_INVALID; }

int pbbi_potential_eval(const pbbi_potential* pot, const void* q, int64_t N, int64_t ldn,
                        void* U_out, void* grad_out, void* stream) {
    if (int rc = check_common(pot, N, ldn)) return rc;
    if (!q && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "q is NULL");
    DeviceGuard guard(pot->device);
    EvalArgs a{pot, q, nullptr, nullptr, N, ldn, U_out, grad_out, nullptr, 0, (hipStream_t)stream};
    return route_eval(a);
}

// ==================================================================== integrators
int pbbi_integrate(const pbbi_potential* pot, int method, void* q, void* p, const void* mass,
                   void* v_out, int64_t N, int64_t ldn, double h, int L, void* stream) {
    if (int rc = check_common(pot, N, ldn)) return rc;
    if (method != PBBI_LEAPFROG && method != PBBI_STORMER_VERLET)
        return pbbi_fail(PBBI_ERR_INVALID, "Invalid integration method selected.");  // src/HMC.py:71
    if (L < 0) return pbbi_fail(PBBI_ERR_INVALID, "numSteps must be >= 0");
    if ((!q || !p) && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "q / p is NULL");
    DeviceGuard guard(pot->device);
    IntegrateArgs a{pot, method, q, p, mass, v_out, N, ldn, h, L, (hipStream_t)stream};
    return route_integrate(a);
}

int pbbi_leapfrog(const pbbi_potential* pot, void* q, void* p, const void* mass, int64_t N,
                  int64_t ldn, double h, int L, void* stream) {
    return pbbi_integrate(pot, PBBI_LEAPFROG, q, p, mass, nullptr, N, ldn, h, L, stream);
}

int pbbi_stormer_verlet(const pbbi_potential* pot, void* q, void* p, const void* mass, int64_t N,
                        int64_t ldn, double h, int L, void* stream) {
    return pbbi_integrate(pot, PBBI_STORMER_VERLET, q, p, mass, nullptr, N, ldn, h, L, stream);
}

// ======================================================================= energies
int pbbi_energy(const pbbi_potential* pot, const void* q, const void* p, const void* mass,
                int64_t N, int64_t ldn, void* H_out, void* weight_out, void* stream) {
    if (int rc = check_common(pot, N, ldn)) return rc;
    if ((!q || !p) && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "q / p is NULL");
    DeviceGuard guard(pot->device);
    EvalArgs a{pot, q, p, mass, N, ldn, H_out, nullptr, weight_out, 0, (hipStream_t)stream};
    return route_energy(a);
}

int pbbi_weights_ratio(const pbbi_potential* pot, const void* newQ, const void* newP,
                       const void* oldQ, const void* oldP, const void* mass, int64_t N,
                       int64_t ldn, void* ratio_out, void* stream) {
    if (int rc = check_common(pot, N, ldn)) return rc;
    if (!ratio_out && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "ratio_out is NULL");
    if ((!newQ || !newP || !oldQ || !oldP) && N > 0)
        return pbbi_fail(PBBI_ERR_INVALID, "a state pointer is NULL");
    DeviceGuard guard(pot->device);
    // pass 1: ratio_out <- oldH ; pass 2: ratio_out <- exp(ratio_out - newH)   (src/HMC.py:109-115)
    EvalArgs a{pot, oldQ, oldP, mass, N, ldn, ratio_out, nullptr, nullptr, 0, (hipStream_t)stream};
    if (int rc = route_energy(a)) return rc;
    EvalArgs b{pot, newQ, newP, mass, N, ldn, ratio_out, nullptr, nullptr, 1, (hipStream_t)stream};
    return route_energy(b);
}

// ================================================================ HMC iteration(s)
static int hmc_check(const pbbi_potential* pot, int method, int64_t N, int64_t ldn, int L) {
    if (int rc = check_common(pot, N, ldn)) return rc;
    if (method != PBBI_LEAPFROG && method != PBBI_STORMER_VERLET)
        return pbbi_fail(PBBI_ERR_INVALID, "Invalid integration method selected.");
    if (L < 0) return pbbi_fail(PBBI_ERR_INVALID, "numSteps must be >= 0");
    return PBBI_OK;
}

int pbbi_hmc_iter(const pbbi_potential* pot, int method, const void* q_in, const void* p_in,
                  const void* u_in, const void* mass, void* q_out, void* p_out, void* ratio_out,
                  uint8_t* reject_out, int64_t N, int64_t ldn, double h, int L, int flags,
                  void* stream) {
    if (int rc = hmc_check(pot, method, N, ldn, L)) return rc;
    if ((!q_in || !p_in || !u_in || !q_out) && N > 0)
        return pbbi_fail(PBBI_ERR_INVALID, "q_in / p_in / u_in / q_out must be non-NULL");
    DeviceGuard guard(pot->device);
    IterArgs a{};
    a.pot = pot; a.method = method; a.q_in = q_in; a.p_in = p_in; a.u_in = u_in; a.mass = mass;
    a.q_out = q_out; a.p_out = p_out; a.ratio_out = ratio_out; a.reject_out = reject_out;
    a.N = N; a.ldn_in = ldn; a.ldn_out = ldn; a.h = h; a.L = L; a.flags = flags;
    a.rng = 0; a.kT = 1.0; a.stream = (hipStream_t)stream;
    return route_hmc(a);
}

int pbbi_hmc_iter_dyn(const pbbi_potential* pot, int method, const void* q_in, const void* p_in,
                      const void* u_in, const void* mass, const int32_t* steps_in, void* q_out, void* p_out,
                      void* ratio_out, uint8_t* reject_out, int32_t* steps_out, int64_t N, int64_t ldn,
                      double h, int L, int flags, double kT, void* stream) {
    if (int rc = hmc_check(pot, method, N, ldn, L)) return rc;
    if ((!q_in || !p_in || !u_in || !q_out) && N > 0)
        return pbbi_fail(PBBI_ERR_INVALID, "q_in / p_in / u_in / q_out must be non-NULL");
    if (!(kT > 0.0)) return pbbi_fail(PBBI_ERR_INVALID, "kT must be > 0");
    if (method != PBBI_LEAPFROG) return pbbi_fail(PBBI_ERR_UNSUPPORTED, "per-chain trajectory lengths: Leapfrog only");
    DeviceGuard guard(pot->device);
    IterArgs a{};
    a.pot = pot; a.method = method; a.q_in = q_in; a.p_in = p_in; a.u_in = u_in; a.mass = mass;
    a.q_out = q_out; a.p_out = p_out; a.ratio_out = ratio_out; a.reject_out = reject_out;
    a.N = N; a.ldn_in = ldn; a.ldn_out = ldn; a.h = h; a.L = L; a.flags = flags;
    a.rng = 0; a.kT = kT; a.stream = (hipStream_t)stream;
    a.steps_in = steps_in; a.steps_out = steps_out;
    if (!pbbi_dyn(a)) {  // plain iteration: every chain takes L steps
        if (steps_out && N > 0) {
            const dim3 grid((unsigned)((N + 255) / 256)), block(256);
            hipLaunchKernelGGL(k_fill_i32, grid, block, 0, a.stream, steps_out, N, L);
        }
        a.steps_in = nullptr; a.steps_out = nullptr;
    }
    return route_hmc(a);
}

int pbbi_hmc_iter_kt(const pbbi_potential* pot, int method, const void* q_in, const void* p_in,
                     const void* u_in, const void* mass, void* q_out, void* p_out, void* ratio_out,
                     uint8_t* reject_out, int64_t N, int64_t ldn, double h, int L, int flags, double kT,
                     void* stream) {
    if (int rc = hmc_check(pot, method, N, ldn, L)) return rc;
    if ((!q_in || !p_in || !u_in || !q_out) && N > 0)
        return pbbi_fail(PBBI_ERR_INVALID, "q_in / p_in / u_in / q_out must be non-NULL");
    if (!(kT > 0.0)) return pbbi_fail(PBBI_ERR_INVALID, "kT must be > 0");
    DeviceGuard guard(pot->device);
    IterArgs a{};
    a.pot = pot; a.method = method; a.q_in = q_in; a.p_in = p_in; a.u_in = u_in; a.mass = mass;
    a.q_out = q_out; a.p_out = p_out; a.ratio_out = ratio_out; a.reject_out = reject_out;
    a.N = N; a.ldn_in = ldn; a.ldn_out = ldn; a.h = h; a.L = L; a.flags = flags;
    a.rng = 0; a.kT = kT; a.stream = (hipStream_t)stream;
    return route_hmc(a);
}

int pbbi_hmc_run(const pbbi_potential* pot, int method, void* q_state, const void* mass,
                 void* samples_out, void* momenta_out, uint8_t* reject_out, void* ratio_out,
                 int64_t N, int64_t ldn, double h, int L, int S, int flags, uint64_t seed,
                 uint64_t iter0, uint64_t chain0, double kT, void* stream) {
    if (flags & (PBBI_PER_CHAIN_STEPS | PBBI_UTURN_STOP))
        return pbbi_fail(PBBI_ERR_INVALID, "per-chain trajectory lengths go through pbbi_hmc_run_dyn");
    return pbbi_hmc_run_dyn(pot, method, q_state, mass, samples_out, momenta_out, reject_out, ratio_out, nullptr,
                            N, ldn, h, L, S, flags, seed, iter0, chain0, kT, stream);
}

int pbbi_hmc_run_dyn(const pbbi_potential* pot, int method, void* q_state, const void* mass,
                     void* samples_out, void* momenta_out, uint8_t* reject_out, void* ratio_out,
                     int32_t* steps_out, int64_t N, int64_t ldn, double h, int L, int S, int flags,
                     uint64_t seed, uint64_t iter0, uint64_t chain0, double kT, void* stream) {
    if (int rc = hmc_check(pot, method, N, ldn, L)) return rc;
    const bool dyn = (flags & (PBBI_PER_CHAIN_STEPS | PBBI_UTURN_STOP)) != 0;
    if (dyn && method != PBBI_LEAPFROG)
        return pbbi_fail(PBBI_ERR_UNSUPPORTED, "per-chain trajectory lengths: Leapfrog only");
    if (S < 0) return pbbi_fail(PBBI_ERR_INVALID, "S must be >= 0");
    if (!q_state && N > 0 && S > 0) return pbbi_fail(PBBI_ERR_INVALID, "q_state must be non-NULL");
    if (!samples_out && momenta_out)
        return pbbi_fail(PBBI_ERR_INVALID, "momenta_out without samples_out (burn-in records nothing)");
    if (!(kT >= 0.0)) return pbbi_fail(PBBI_ERR_INVALID, "kT must be >= 0");
    if ((flags & PBBI_BETA_ACCEPT) && !(kT > 0.0))
        return pbbi_fail(PBBI_ERR_INVALID, "PBBI_BETA_ACCEPT needs kT > 0");
    // the Philox counter carries the low 32 bits of the iteration index (include/pbbi.h): a run that
    // crosses 2^32 would silently repeat the draws of iterations 0, 1, ...
    if (iter0 > UINT32_MAX || iter0 + (uint64_t)S > (uint64_t)UINT32_MAX + 1)
        return pbbi_fail(PBBI_ERR_INVALID, "iter0 + S must be <= 2^32 (the Philox counter holds 32 iteration bits)");
    if (S == 0 || N == 0) return PBBI_OK;
    DeviceGuard guard(pot->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t es = elem_size(pot->dtype);
    const size_t slab = (size_t)pot->D * (size_t)N;  // elements per (D, N) sample slab
    // samples_out == NULL: burn-in.  Nothing is recorded; the state ping-pongs between two scratch
    // slabs (a third of the traffic of a recorded iteration is the momentum slab, which is skipped).
    char* pong = nullptr;
    if (!samples_out && hipMallocAsync((void**)&pong, 2 * slab * es, st) != hipSuccess)
        return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the burn-in state");
    auto slab_of = [&](int i) -> char* {
        return samples_out ? (char*)samples_out + (size_t)i * slab * es : pong + (size_t)(i & 1) * slab * es;
    };
    // paths that need device scratch (GEMM, workspace-streaming, user plugins) report what they
    // took in iteration 0; iterations 1.. carve the same from ONE arena instead of allocating again
    size_t need = 0;
    void* arena = nullptr;
    int rc = PBBI_OK;
    // dense MFMA kernel: the gradient at the chain's position is carried from one iteration to the next
    // (two (D, N) slabs + one byte per chain, kernels_dense.hip CARRY) -- L mat-vecs per iteration, not L + 1
    char* carry = nullptr;
    // (the dense path's slabs hold the PADDED rows: DP x N each)
    size_t carry_bytes = 2 * (size_t)(pot->DP ? pot->DP : pot->D) * (size_t)N * sizeof(double), sel_bytes = (size_t)N;
    bool use_carry = false;
    // 128 < D <= 256: streamed dense kernel or GEMM path -- decided here, once, for every iteration of the run
    // (iteration 0 reads the caller's stride, the others the slabs'; the carried formats of the two paths differ)
    int route_hint = 0;
    if (pot->kind != KIND_CUSTOM && is_big(pot)) {
        IterArgs probe{};
        probe.pot = pot; probe.method = method; probe.N = N; probe.L = L; probe.flags = flags;
        probe.ldn_in = ldn; probe.ldn_out = N;
        if (!dense_stream_applies(probe)) route_hint |= PBBI_ROUTE_NO_DENSE_STREAM;
    }
    if (S >= 2 && pot->kind != KIND_CUSTOM && (is_dense(pot) || is_big(pot))) {
        IterArgs probe{};
        probe.pot = pot; probe.method = method; probe.N = N; probe.L = L; probe.flags = flags;
        probe.ldn_in = ldn; probe.ldn_out = N; probe.route_hint = route_hint;
        if (is_big(pot) && dense_stream_applies(probe)) {  // the dense kernel's format: two padded slabs + a byte per chain
            use_carry = dense_stream_carry_applies(probe);
            carry_bytes = 2 * (size_t)pot->DPS * (size_t)N * sizeof(double);
        } else if (is_big(pot)) {  // the GEMM path keeps the x.g partial sums next to the accept bytes
            use_carry = big_carry_applies(probe);
            big_carry_bytes(pot, N, &carry_bytes, &sel_bytes);
        } else {
            use_carry = dense_carry_applies(probe);
        }
    }
    if (use_carry) {
        if (hipMallocAsync((void**)&carry, carry_bytes + sel_bytes, st) != hipSuccess) {
            carry = nullptr;
            use_carry = false;  // (without the buffers every iteration forms its own gradient)
            (void)hipGetLastError();
        } else if (hipMemsetAsync(carry + carry_bytes, 0, sel_bytes, st) != hipSuccess) {
            (void)hipFreeAsync(carry, st);
            if (pong) (void)hipFreeAsync(pong, st);
            return pbbi_fail(PBBI_ERR_HIP, "hipMemsetAsync of the carry selector failed");
        }
    }
    for (int i = 0; i < S && rc == PBBI_OK;) {
        IterArgs a{};
        a.pot = pot; a.method = method; a.mass = mass;
        // iteration i reads the state left by iteration i-1: the previous slab
        a.q_in = (i == 0) ? (const char*)q_state : slab_of(i - 1);
        a.ldn_in = (i == 0) ? ldn : N;
        a.q_out = slab_of(i);
        a.p_out = momenta_out ? (char*)momenta_out + (size_t)i * slab * es : nullptr;
        a.ldn_out = N;
        a.ratio_out = ratio_out ? (char*)ratio_out + (size_t)i * N * es : nullptr;
        a.reject_out = reject_out ? reject_out + (size_t)i * N : nullptr;
        a.N = N; a.h = h; a.L = L; a.flags = flags;
        a.rng = 1; a.seed = seed; a.iter = iter0 + (uint64_t)i; a.chain0 = chain0; a.kT = kT;
        a.stream = st;
        a.scratch = arena; a.scratch_bytes = arena ? need : 0; a.scratch_used = (i == 0) ? &need : nullptr;
        a.route_hint = route_hint;
        if (use_carry) {
            a.carry = (i == 0) ? 1 : 2;
            a.carry_g = carry;
            a.carry_sel = (uint8_t*)(carry + carry_bytes);
        }
        if (steps_out && dyn) a.steps_out = steps_out + (size_t)i * N;
        // kernels that keep the chain on chip take several iterations per launch
        int chunk = route_fused_iterations(a);
        if (chunk > 1) {  // launches of (nearly) equal length instead of full chunks and a short tail
            const int rem = S - i, launches = (rem + chunk - 1) / chunk;
            chunk = (rem + launches - 1) / launches;
        }
        if (chunk > S - i) chunk = S - i;
        if (steps_out && !dyn) {  // fixed length: every chain of the iterations this call covers took L steps
            const int64_t cells = (int64_t)N * (chunk > 1 ? chunk : 1);
            hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st,
                               steps_out + (size_t)i * N, cells, L);
        }
        if (chunk > 1) {
            a.fuse_S = chunk;
            a.fuse_wrap2 = samples_out ? 0 : 1;
            a.fuse_slab0 = i;
            a.fuse_q_base = samples_out ? samples_out : (void*)pong;
        }
        rc = route_hmc(a);
        if (rc == PBBI_OK && i == 0 && need > 0 && S > 1 && hipMallocAsync(&arena, need, st) != hipSuccess)
            arena = nullptr;  // keep allocating per iteration
        i += chunk > 1 ? chunk : 1;
    }
    if (arena) (void)hipFreeAsync(arena, st);
    if (carry) (void)hipFreeAsync(carry, st);
    // leave the chain state in q_state (strided D2D copy of the last slab)
    if (rc == PBBI_OK &&
        hipMemcpy2DAsync(q_state, (size_t)ldn * es, slab_of(S - 1), (size_t)N * es, (size_t)N * es,
                         (size_t)pot->D, hipMemcpyDeviceToDevice, st) != hipSuccess)
        rc = pbbi_fail(PBBI_ERR_HIP, "hipMemcpy2DAsync of the final state failed");
    if (pong) (void)hipFreeAsync(pong, st);
    return rc;
}

int pbbi_hmc_run_gist(const pbbi_potential* pot, void* q_state, const void* mass, void* samples_out,
                      void* momenta_out, uint8_t* reject_out, void* ratio_out, int32_t* tau_out, int64_t N,
                      int64_t ldn, double h, int Lmax, int S, int flags, uint64_t seed, uint64_t iter0,
                      uint64_t chain0, double kT, void* stream) {
    if (int rc = hmc_check(pot, PBBI_LEAPFROG, N, ldn, Lmax)) return rc;
    if (Lmax < 1) return pbbi_fail(PBBI_ERR_INVALID, "GIST: the U-turn search needs Lmax >= 1");
    if (S < 0) return pbbi_fail(PBBI_ERR_INVALID, "S must be >= 0");
    if (!(kT > 0.0)) return pbbi_fail(PBBI_ERR_INVALID, "kT must be > 0");
    if (flags & (PBBI_PER_CHAIN_STEPS | PBBI_UTURN_STOP | PBBI_KDK_FMA))
        return pbbi_fail(PBBI_ERR_INVALID, "GIST: flags may hold PBBI_COMPAT_P_FROM_OLDQ, PBBI_BETA_ACCEPT, PBBI_DRAW_F64");
    if (iter0 > UINT32_MAX || iter0 + (uint64_t)S > (uint64_t)UINT32_MAX + 1)
        return pbbi_fail(PBBI_ERR_INVALID, "iter0 + S must be <= 2^32 (the Philox counter holds 32 iteration bits)");
    if (S == 0 || N == 0) return PBBI_OK;
    if (!q_state || !samples_out) return pbbi_fail(PBBI_ERR_INVALID, "q_state and samples_out must be non-NULL");
    DeviceGuard guard(pot->device);
    if (pot->dtype == PBBI_F64)
        return gist_run<double>(pot, q_state, mass, samples_out, momenta_out, reject_out, ratio_out, tau_out, N, ldn, h,
                                Lmax, S, flags, seed, iter0, chain0, kT, (hipStream_t)stream);
    return gist_run<float>(pot, q_state, mass, samples_out, momenta_out, reject_out, ratio_out, tau_out, N, ldn, h, Lmax,
                           S, flags, seed, iter0, chain0, kT, (hipStream_t)stream);
}

int pbbi_replica_exchange(const pbbi_potential* pot, void* q, int64_t Nr, int R, int64_t ldn, const double* betas,
                          int parity, uint64_t seed, uint64_t iter, uint64_t chain0, uint8_t* swapped_out,
                          void* stream) {
    if (!pot) return pbbi_fail(PBBI_ERR_INVALID, "potential handle is NULL");
    if (R < 1 || Nr < 0) return pbbi_fail(PBBI_ERR_INVALID, "need R >= 1 rungs of Nr >= 0 chains");
    const int64_t N = (int64_t)R * Nr;
    if (int rc = check_common(pot, N, ldn)) return rc;
    if (parity != 0 && parity != 1) return pbbi_fail(PBBI_ERR_INVALID, "parity must be 0 or 1");
    if (iter > UINT32_MAX) return pbbi_fail(PBBI_ERR_INVALID, "iter must be < 2^32");
    if (N == 0 || R < 2 || parity + 1 >= R) return PBBI_OK;
    if (!q || !betas) return pbbi_fail(PBBI_ERR_INVALID, "q / betas is NULL");
    DeviceGuard guard(pot->device);
    hipStream_t st = (hipStream_t)stream;
    Scratch ws(st);
    void* U = ws.get((size_t)N * elem_size(pot->dtype));
    if (!U) return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the potential energies");
    EvalArgs e{pot, q, nullptr, nullptr, N, ldn, U, nullptr, nullptr, 0, st};
    if (int rc = route_eval(e)) return rc;
    const int pairs = (R - parity) / 2;
    const dim3 grid((unsigned)((Nr + 255) / 256), (unsigned)pairs), block(256);
    if (pot->dtype == PBBI_F64)
        hipLaunchKernelGGL(k_replica_exchange<double>, grid, block, 0, st, (double*)q, ldn, (const double*)U, betas,
                           pot->D, Nr, R, parity, seed, iter, chain0, swapped_out);
    else
        hipLaunchKernelGGL(k_replica_exchange<float>, grid, block, 0, st, (float*)q, ldn, (const float*)U, betas,
                           pot->D, Nr, R, parity, seed, iter, chain0, swapped_out);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// what pbbi_hmc_run would do with these arguments, in words (include/pbbi.h)
int pbbi_describe_run(const pbbi_potential* pot, int method, int64_t N, int64_t ldn, int L, int S, int flags,
                      char* out, int out_len) {
    if (int rc = hmc_check(pot, method, N, ldn, L)) return rc;
    if (!out || out_len < 1) return pbbi_fail(PBBI_ERR_INVALID, "out buffer missing");
    IterArgs a{};
    a.pot = pot; a.method = method; a.N = N; a.ldn_in = ldn; a.ldn_out = N; a.L = L; a.flags = flags; a.rng = 1;
    a.kT = 1.0;
    // (the fused-launch queries look at the carry buffers only for presence)
    a.carry = 2; a.carry_g = (void*)(uintptr_t)16; a.carry_sel = (uint8_t*)(uintptr_t)16;
    std::string d;
    int fuse = 1;
    if (pot->kind == KIND_CUSTOM) {
        d = "user-potential plugin kernels (one chain per lane; registers up to D = 16 / 32, workspace beyond)";
        fuse = route_fused_iterations(a);
    } else if (is_big(pot) && dense_stream_applies(a)) {
        d = "k_dense_hmc (streamed P): register-resident MFMA kernel, 16 chains per wave, one wave per SIMD, the "
            "precision matrix streamed through an LDS ring (rows padded to " + std::to_string(pot->DPS) + ")";
        const bool carry = S >= 2 && dense_stream_carry_applies(a);
        d += carry ? "; gradient carried between iterations: yes (L mat-vecs per iteration)"
                   : "; gradient carried between iterations: no (L + 1 mat-vecs per iteration)";
        if (!carry) a.carry = 0;
        fuse = carry ? dense_stream_fused_iterations(a) : 1;
    } else if (is_big(pot)) {
        d = "kernels_big: one fused MFMA GEMM per leapfrog step over the whole ensemble";
        d += big_carry_applies(a) && S >= 2 ? "; gradient carried between iterations: yes (L GEMMs per iteration)"
                                            : "; gradient carried between iterations: no (L + 1 GEMMs per iteration)";
    } else if (is_dense(pot)) {
        d = "k_dense_hmc: register-resident MFMA kernel, 16 chains per wave";
        const bool carry = S >= 2 && dense_carry_applies(a);
        if (carry) {
            d += "; gradient carried between iterations: yes (L mat-vecs per iteration)";
        } else {
            d += "; gradient carried between iterations: no (L + 1 mat-vecs per iteration) -- ";
            const uint64_t lim = (PBBI_CARRY_MAX_BYTES - 1) / ((uint64_t)pot->DP * 16u);
            if (S < 2) d += "a run of one iteration";
            else if (L < 1 || pbbi_dyn(a)) d += "runs of a fixed trajectory length L >= 1 only";
            else if ((uint64_t)N > lim)
                d += "the two carried-gradient slabs (D*N*16 bytes) must stay below 2^32 for 32-bit buffer offsets: "
                     "at D = " + std::to_string(pot->D) + " (rows padded to " + std::to_string(pot->DP) + ") that is N <= " + std::to_string(lim) +
                     " chains per call; shard the ensemble or split the call";
            else d += "switched off (PBBI_NO_CARRY / PBBI_DENSE_V1)";
        }
        if (!carry) a.carry = 0;
        fuse = carry ? dense_fused_iterations(a) : 1;
    } else {
        d = lane_route_name(a);
        fuse = lane_fused_iterations(a);
    }
    d += "; iterations per launch: up to " + std::to_string(fuse < 1 ? 1 : fuse);
    if (flags & PBBI_DRAW_F64) d += "; momentum draw: double precision";
    snprintf(out, (size_t)out_len, "%s", d.c_str());
    return PBBI_OK;
}

// ============================================================================ RNG
int pbbi_philox_normal(uint64_t seed, int rng_stream, uint64_t iter, uint64_t chain0, int D,
                       int64_t N, int64_t ldn, double scale, const void* scale_per_chain,
                       int dtype, int device, void* out, void* stream) {
    if (D < 1 || N < 0 || ldn < N) return pbbi_fail(PBBI_ERR_INVALID, "bad D / N / ldn");
    if (iter > UINT32_MAX) return pbbi_fail(PBBI_ERR_INVALID, "iter must be < 2^32");
    if (!out && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "out is NULL");
    if (N == 0) return PBBI_OK;
    DeviceGuard guard(device);
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_philox_normal<double>, grid, block, 0, (hipStream_t)stream, seed,
                           rng_stream, iter, chain0, D, N, ldn, scale,
                           (const double*)scale_per_chain, (double*)out);
    else if (dtype == PBBI_F32)
        hipLaunchKernelGGL(k_philox_normal<float>, grid, block, 0, (hipStream_t)stream, seed,
                           rng_stream, iter, chain0, D, N, ldn, scale, (const float*)scale_per_chain,
                           (float*)out);
    else
        return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int pbbi_philox_steps(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, int L, int device,
                      int32_t* out, void* stream) {
    if (N < 0 || L < 0) return pbbi_fail(PBBI_ERR_INVALID, "bad N / L");
    if (iter > UINT32_MAX) return pbbi_fail(PBBI_ERR_INVALID, "iter must be < 2^32");
    if (!out && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "out is NULL");
    if (N == 0) return PBBI_OK;
    DeviceGuard guard(device);
    hipLaunchKernelGGL(k_philox_steps, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       iter, chain0, N, L, out);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int pbbi_philox_uniform(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, int dtype,
                        int device, void* out, void* stream) {
    if (N < 0) return pbbi_fail(PBBI_ERR_INVALID, "bad N");
    if (iter > UINT32_MAX) return pbbi_fail(PBBI_ERR_INVALID, "iter must be < 2^32");
    if (!out && N > 0) return pbbi_fail(PBBI_ERR_INVALID, "out is NULL");
    if (N == 0) return PBBI_OK;
    DeviceGuard guard(device);
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_philox_uniform<double>, grid, block, 0, (hipStream_t)stream, seed, iter,
                           chain0, N, (double*)out);
    else if (dtype == PBBI_F32)
        hipLaunchKernelGGL(k_philox_uniform<float>, grid, block, 0, (hipStream_t)stream, seed, iter,
                           chain0, N, (float*)out);
    else
        return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// ========================================================================= layout
int pbbi_transpose_sdn_to_dns(const void* src_sdn, void* dst_dns, int S, int D, int64_t N,
                              int dtype, int device, void* stream) {
    if (S < 0 || D < 1 || N < 0) return pbbi_fail(PBBI_ERR_INVALID, "bad S / D / N");
    if (S == 0 || N == 0) return PBBI_OK;
    if (!src_sdn || !dst_dns) return pbbi_fail(PBBI_ERR_INVALID, "src / dst is NULL");
    DeviceGuard guard(device);
    const int64_t M = (int64_t)D * N;
    const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((S + 31) / 32)), block(32, 8);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_transpose<double>, grid, block, 0, (hipStream_t)stream,
                           (const double*)src_sdn, (double*)dst_dns, S, M);
    else if (dtype == PBBI_F32)
        hipLaunchKernelGGL(k_transpose<float>, grid, block, 0, (hipStream_t)stream,
                           (const float*)src_sdn, (float*)dst_dns, S, M);
    else
        return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

// =================================================================== statistics
int pbbi_sample_moments(const void* samples_sdn, int S, int D, int64_t N, int dtype, int device,
                        void* mean_out, void* var_out, void* stream) {
    if (S < 1 || D < 1 || N < 1) return pbbi_fail(PBBI_ERR_INVALID, "S, D, N must be >= 1");
    if (!samples_sdn) return pbbi_fail(PBBI_ERR_INVALID, "samples is NULL");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    DeviceGuard guard(device);
    const int chunks = 64;
    double* part = nullptr;
    PBBI_HIP(hipMallocAsync((void**)&part, (size_t)D * chunks * 2 * sizeof(double), (hipStream_t)stream));
    const dim3 grid(chunks, D), block(256);
    if (dtype == PBBI_F64) {
        hipLaunchKernelGGL(k_moments_partial<double>, grid, block, 0, (hipStream_t)stream,
                           (const double*)samples_sdn, S, D, N, chunks, part);
        hipLaunchKernelGGL(k_moments_final<double>, dim3((D + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                           (const double*)part, D, chunks, (double)S * (double)N, (double*)mean_out,
                           (double*)var_out);
    } else {
        hipLaunchKernelGGL(k_moments_partial<float>, grid, block, 0, (hipStream_t)stream,
                           (const float*)samples_sdn, S, D, N, chunks, part);
        hipLaunchKernelGGL(k_moments_final<float>, dim3((D + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                           (const double*)part, D, chunks, (double)S * (double)N, (float*)mean_out,
                           (float*)var_out);
    }
    PBBI_HIP(hipGetLastError());
    PBBI_HIP(hipFreeAsync(part, (hipStream_t)stream));
    return PBBI_OK;
}

int pbbi_chain_moments(const void* samples_sdn, int S, int D, int64_t N, int dtype, int device,
                       void* chain_mean_out, void* chain_var_out, void* stream) {
    if (S < 2 || D < 1 || N < 1) return pbbi_fail(PBBI_ERR_INVALID, "need S >= 2, D >= 1, N >= 1");
    if (!samples_sdn) return pbbi_fail(PBBI_ERR_INVALID, "samples is NULL");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    DeviceGuard guard(device);
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)D), block(256);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_chain_moments<double>, grid, block, 0, (hipStream_t)stream,
                           (const double*)samples_sdn, S, D, N, (double*)chain_mean_out,
                           (double*)chain_var_out);
    else
        hipLaunchKernelGGL(k_chain_moments<float>, grid, block, 0, (hipStream_t)stream,
                           (const float*)samples_sdn, S, D, N, (float*)chain_mean_out,
                           (float*)chain_var_out);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int pbbi_chain_autocov(const void* samples_sdn, const void* chain_mean, int S, int D, int64_t N, int T,
                       int dtype, int device, double* acov_out, void* stream) {
    if (S < 2 || D < 1 || N < 1) return pbbi_fail(PBBI_ERR_INVALID, "need S >= 2, D >= 1, N >= 1");
    if (T < 0 || T > PBBI_MAX_LAG) return pbbi_fail(PBBI_ERR_INVALID, "T must be in [0, PBBI_MAX_LAG]");
    if (!samples_sdn || !chain_mean || !acov_out) return pbbi_fail(PBBI_ERR_INVALID, "a pointer is NULL");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    DeviceGuard guard(device);
    hipStream_t st = (hipStream_t)stream;
    const int n_blocks = (int)((N + 255) / 256);
    double* part = nullptr;
    PBBI_HIP(hipMallocAsync((void**)&part, sizeof(double) * (size_t)n_blocks * D * (PBBI_MAX_LAG + 1), st));
    const dim3 grid((unsigned)n_blocks, (unsigned)D), block(256);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_chain_autocov<double>, grid, block, 0, st, (const double*)samples_sdn,
                           (const double*)chain_mean, S, D, N, part);
    else
        hipLaunchKernelGGL(k_chain_autocov<float>, grid, block, 0, st, (const float*)samples_sdn,
                           (const float*)chain_mean, S, D, N, part);
    const int outs = (T + 1) * D;
    hipLaunchKernelGGL(k_autocov_final, dim3((outs + 255) / 256), dim3(256), 0, st, (const double*)part,
                       n_blocks, D, T, 1.0 / ((double)S * (double)N), acov_out);
    PBBI_HIP(hipGetLastError());
    PBBI_HIP(hipFreeAsync(part, st));
    return PBBI_OK;
}

int pbbi_sample_covariance(const void* samples_sdn, int S, int D, int64_t N, int dtype, int device,
                           const double* mean, double* cov_out, void* stream) {
    if (S < 1 || D < 1 || N < 1) return pbbi_fail(PBBI_ERR_INVALID, "S, D, N must be >= 1");
    if (!samples_sdn || !mean || !cov_out) return pbbi_fail(PBBI_ERR_INVALID, "a pointer is NULL");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    DeviceGuard guard(device);
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (D + 15) / 16;
    const int64_t M = (int64_t)S * N;
    int chunks = (int)(M / 4096 > 0 ? M / 4096 : 1);
    const int max_chunks = tiles * tiles >= 64 ? 64 : 1024 / (tiles * tiles);
    if (chunks > max_chunks) chunks = max_chunks;
    double* part = nullptr;
    PBBI_HIP(hipMallocAsync((void**)&part, sizeof(double) * (size_t)chunks * D * D, st));
    const dim3 grid((unsigned)chunks, (unsigned)tiles, (unsigned)tiles), block(256);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_cov_partial<double>, grid, block, 0, st, (const double*)samples_sdn, mean, S, D,
                           N, chunks, part);
    else
        hipLaunchKernelGGL(k_cov_partial<float>, grid, block, 0, st, (const float*)samples_sdn, mean, S, D, N,
                           chunks, part);
    hipLaunchKernelGGL(k_cov_final, dim3((D * D + 255) / 256), dim3(256), 0, st, (const double*)part, chunks,
                       D, 1.0 / (double)M, cov_out);
    PBBI_HIP(hipGetLastError());
    PBBI_HIP(hipFreeAsync(part, st));
    return PBBI_OK;
}

// ============================================================== ensemble weights
int pbbi_reduce_min(const void* x, int64_t N, int dtype, int device, double* min_out, void* stream) {
    return weights_reduce<0>(x, N, 0.0, nullptr, dtype, device, nullptr, min_out, (hipStream_t)stream);
}

int pbbi_canonical_weights(const void* H, int64_t N, double beta, const double* hmin, int dtype,
                           int device, void* w_out, double* sum_out, void* stream) {
    return weights_reduce<1>(H, N, beta, hmin, dtype, device, w_out, sum_out, (hipStream_t)stream);
}

int pbbi_scale_inverse(void* w, int64_t N, const double* sum, int dtype, int device, void* stream) {
    if (N < 0) return pbbi_fail(PBBI_ERR_INVALID, "bad N");
    if (dtype != PBBI_F64 && dtype != PBBI_F32) return pbbi_fail(PBBI_ERR_INVALID, "unknown dtype");
    if (!sum || (N > 0 && !w)) return pbbi_fail(PBBI_ERR_INVALID, "a pointer is NULL");
    if (N == 0) return PBBI_OK;
    DeviceGuard guard(device);
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (dtype == PBBI_F64)
        hipLaunchKernelGGL(k_scale_inverse<double>, grid, block, 0, (hipStream_t)stream, (double*)w, N, sum);
    else
        hipLaunchKernelGGL(k_scale_inverse<float>, grid, block, 0, (hipStream_t)stream, (float*)w, N, sum);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

}  // extern "C"
