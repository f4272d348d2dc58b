// kernels_big.hip -- dense-precision Gaussian with D > 128 (fp64 and fp32), gfx950.
//
// BASELINE config 5 (D = 4096, fp32, 8192 chains): the precision matrix (64 MiB) no longer fits
// LDS and the state of a chain (3 x 4096 values) no longer fits registers, so the trajectory is
// a sequence of L+1 tiled MFMA GEMMs  G = P (Q - mu)  over the whole ensemble, each with the
// leapfrog bookkeeping FUSED INTO ITS EPILOGUE (kick, drift into the other q buffer, partial
// x.g sums for the potential energy).  State streams through HBM once per step:
//     read q_j (GEMM operand + epilogue), read/write vh, write q_{j+1}
// and the GEMM is MFMA-bound: 2 D^2 flop vs ~5 D w bytes per step*chain.
//
//   * A operand: P^T stored [k][i] (uploaded transposed and zero-padded to a multiple of 128),
//     B operand: X = q - mu stored [k][n]  (the (D, N) state layout as it is).  Both tiles are
//     therefore "k-major" with a contiguous inner dimension: global -> LDS is a straight,
//     coalesced tile copy and the MFMA operand reads are lane-consecutive (conflict-free).
//   * fp32: v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 flop/clk/SIMD = the f32 vector
//     peak), 128 x 128 x 32 block tile, 4 waves as 2 x 2, each wave 2 x 2 tiles of 32 x 32.
//     fp64: v_mfma_f64_16x16x4_f64, 128 x 128 x 16 block tile, each wave 4 x 4 tiles of 16 x 16.
//   * Kick-drift-kick leapfrog (see kernels_dense.hip): epilogue of the GEMM on q_j does
//         vh += -(g/m) * hk;   q_{j+1} = q_j + vh*h   (written to the other q buffer)
//     first and last kicks are half kicks; the last epilogue does not drift.
//   * Energies: per-chain sums over D are two-stage and deterministic (per-tile partials, then
//     one ordered sum), never atomics: the accept decision must not depend on arrival order.
//   * Block order: consecutive blockIdx sweep the P^T panels for ONE 128-chain panel of X, so
//     that X panel (K x 128) stays hot while it is reused D/128 times; P^T (64 MiB at D=4096
//     fp32) is re-read by every X panel and stays resident in the 256 MiB Infinity Cache.
//
// Stormer-Verlet runs on the same epilogue: no closing half kick, one more drift, and one
// evaluation GEMM for U of the final position.
//
// Inside pbbi_hmc_run the gradient of the point an iteration starts from is carried over from the
// previous iteration (run_hmc): L GEMMs per Leapfrog iteration, the first half kick is an elementwise pass.
// fp32, zero mean, whole aligned tiles (config C5's every block) run on k_big_gemm_wide (256 x 128 tiles).
#include <cstdlib>
#include <vector>

#include "pbbi_internal.h"
#include "pbbi_rng.h"

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;

// Block tile 128 x 128 for both types.  fp32: 4 waves (2 x 2), 64 x 64 per wave (64 accumulator
// registers).  fp64: the same wave tile needs 128 accumulator registers and ran at one wave per SIMD
// (328 registers: 0.27-0.47 of the fp64 MFMA peak at D = 256...1024); with 8 waves (2 x 4), 64 x 32
// per wave, it fits 128 registers: two workgroups per CU, four waves per SIMD.
template <typename T> struct Cfg;
#ifndef PBBI_BIG_F32_BK
#define PBBI_BIG_F32_BK 32
#endif
#ifndef PBBI_BIG_F32_MINW
#define PBBI_BIG_F32_MINW 2
#endif
template <> struct Cfg<float> {
    static constexpr int BK = PBBI_BIG_F32_BK, NTHR = 256, WAVES_N = 2, WTN = 64, MIN_WAVES = PBBI_BIG_F32_MINW;
};
template <> struct Cfg<double> {
    static constexpr int BK = 16, NTHR = 512, WAVES_N = 4, WTN = 32, MIN_WAVES = 4;
};

enum { EPI_EVAL = 0, EPI_KDK = 1 };

template <typename T>
struct GemmPrm {
    const T* PT;      // DPAD x DPAD, [k][i], zero padded
    const T* mu;      // DPAD, zero padded
    const T* q;       // (D, N) operand, leading stride ldq
    T* q_next;        // (D, N) drift target (ldw) or nullptr (no drift)
    T* vh;            // (D, N) half-step velocity, updated in place (ldw); EPI_KDK
    const T* minv;    // N: 1/m per chain, or nullptr (= 1)
    T* grad_out;      // (D, N) with stride ldg, or nullptr: the gradient itself (EPI_EVAL's output; EPI_KDK: kept for
                      // the next iteration of a run, see run_hmc)
    T* xg_part;       // [DPAD/BM][N] partial sums of x*g over the tile's rows, or nullptr
    int64_t N, ldq, ldw, ldg;
    int D, DPAD;
    T hk, h;
    int no_dma;  // PBBI_BIG_NO_DMA=1: A/B switch for profiling (register-staged tiles everywhere)
};

// ---- per-wave MFMA micro-kernels ---------------------------------------------------------------
// As/Bs are [BK][BM] / [BK][BN] tiles in LDS.  wm/wn: this wave's 64 x 64 corner of the block.
struct MmaF32 {
    v16f32 acc[2][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    __device__ __forceinline__ void tile(const float* As, const float* Bs, int wm, int wn, int lane) {
        const int r = lane & 31, kh = lane >> 5;
        // Software-pipelined by hand: the operands of K-step s+1 are read from LDS BEFORE the four
        // MFMAs of K-step s issue, and the sched_barrier pins that order (left alone, hipcc emits
        // read, s_waitcnt lgkmcnt(0), 4 MFMAs per step: every step then waits out an LDS round trip
        // that only the previous step's last MFMA covers).
        const float* ap = As + kh * BM + wm + r;
        const float* bp = Bs + kh * BN + wn + r;
        float a[2][2], b[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[0][t] = ap[32 * t];  // A[i = r][k = kh]
            b[0][t] = bp[32 * t];  // B[k = kh][n = r]
        }
#pragma unroll
        for (int s = 0; s < Cfg<float>::BK / 2; ++s) {
            const int c = s & 1, nx = c ^ 1;
            if (s + 1 < Cfg<float>::BK / 2) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[nx][t] = ap[2 * (s + 1) * BM + 32 * t];
                    b[nx][t] = bp[2 * (s + 1) * BN + 32 * t];
                }
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][ta], b[c][tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // visit(i, n, tb, value): C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    template <typename F>
    __device__ __forceinline__ void each(int wm, int wn, int lane, F&& f) const {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    f(wm + 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), wn + 32 * tb + (lane & 31),
                      tb, acc[ta][tb][r]);
    }
};

struct MmaF64 {  // 64 x 32 per wave: 4 x 2 tiles of 16 x 16
    v4f64 acc[4][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
    }
    __device__ __forceinline__ void tile(const double* As, const double* Bs, int wm, int wn, int lane) {
        const int r = lane & 15, kq = lane >> 4;
        // operands of K-step s+1 are read before the eight MFMAs of K-step s issue (see MmaF32::tile)
        const double* ap = As + kq * BM + wm + r;
        const double* bp = Bs + kq * BN + wn + r;
        double a[2][4], b[2][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[0][t] = ap[16 * t];  // A[i = r][k = kq]
#pragma unroll
        for (int t = 0; t < 2; ++t) b[0][t] = bp[16 * t];  // B[k = kq][n = r]
#pragma unroll
        for (int s = 0; s < Cfg<double>::BK / 4; ++s) {
            const int c = s & 1, nx = c ^ 1;
            if (s + 1 < Cfg<double>::BK / 4) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[nx][t] = ap[4 * (s + 1) * BM + 16 * t];
#pragma unroll
                for (int t = 0; t < 2; ++t) b[nx][t] = bp[4 * (s + 1) * BN + 16 * t];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][ta], b[c][tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // f64 C/D map: col = lane&15, row = (lane>>4) + 4*reg
    template <typename F>
    __device__ __forceinline__ void each(int wm, int wn, int lane, F&& f) const {
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f(wm + 16 * ta + (lane >> 4) + 4 * r, wn + 16 * tb + (lane & 15), tb, acc[ta][tb][r]);
    }
};

template <typename T> struct MmaOf;
template <> struct MmaOf<float> { using type = MmaF32; };
template <> struct MmaOf<double> { using type = MmaF64; };

// ---- the GEMM with fused epilogue ----------------------------------------------------------------
template <typename T, int EPI, bool ZMEAN>
__global__ void __launch_bounds__(Cfg<T>::NTHR, Cfg<T>::MIN_WAVES) k_big_gemm(GemmPrm<T> prm) {
    constexpr int BK = Cfg<T>::BK, NTHR = Cfg<T>::NTHR, WAVES_N = Cfg<T>::WAVES_N;
    extern __shared__ __attribute__((aligned(16))) char smem_big[];
    T (*As)[BK * BM] = reinterpret_cast<T (*)[BK * BM]>(smem_big);                      // [2]
    T (*Bs)[BK * BN] = reinterpret_cast<T (*)[BK * BN]>(smem_big + 2 * BK * BM * sizeof(T));  // [2]
    T (*red)[BN] = reinterpret_cast<T (*)[BN]>(smem_big + 4 * BK * BM * sizeof(T));    // [2]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform
    const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * Cfg<T>::WTN;
    const int n_tiles_m = prm.DPAD / BM;
    const int bm = blockIdx.x % n_tiles_m, bn = blockIdx.x / n_tiles_m;  // a P^T panel's tiles are
    const int i0 = bm * BM;                                              // consecutive: spread over XCDs
    const int64_t n0 = (int64_t)bn * BN;

    // staging map: 16-byte chunk c = tid + 256*j of a BK x 128 tile -> (k = c / CPR, x = VEC*(c % CPR)),
    // VEC elements per chunk, CPR = 128/VEC chunks per tile row: every wave instruction moves
    // whole 128-B..1-KiB row segments (coalesced), one global_load_dwordx4 per chunk.
    // Every vector instruction costs matrix-pipe time (see kernels_dense.hip), so the common case
    // -- whole tile inside D x N, aligned rows -- is a FAST path with loop-invariant 32-bit chunk
    // offsets against a per-tile uniform base pointer and no selects; PMC showed 2.2 VALU
    // instructions per MFMA with the guarded form.
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CPR = BM / VEC;
    constexpr int NCH = BK * BM / VEC / NTHR;  // chunks per thread per tile (4)
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    const bool q_vec_ok = (prm.ldq % VEC == 0) && ((reinterpret_cast<uintptr_t>(prm.q) & 15) == 0);
    const bool cols_full = q_vec_ok && (n0 + BN <= prm.N);          // block-uniform
    uint32_t offA[NCH], offB[NCH];
    int kc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = tid + NTHR * j, k = c / CPR, x = (c % CPR) * VEC;
        kc[j] = k;
        offA[j] = (uint32_t)k * (uint32_t)prm.DPAD + (uint32_t)x;
        offB[j] = (uint32_t)k * (uint32_t)prm.ldq + (uint32_t)x;
    }
    const T* pa0 = prm.PT + i0;
    const T* pb0 = prm.q + n0;
    vecT ra[NCH], rb[NCH];
    auto load_tiles = [&](int k0) {
        const T* pa = pa0 + (size_t)k0 * prm.DPAD;  // uniform per tile
        const T* pb = pb0 + (int64_t)k0 * prm.ldq;
#pragma unroll
        for (int j = 0; j < NCH; ++j) ra[j] = *reinterpret_cast<const vecT*>(pa + offA[j]);
        if (cols_full && k0 + BK <= prm.D) {  // fast path (uniform branch)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                vecT v = *reinterpret_cast<const vecT*>(pb + offB[j]);
                if constexpr (!ZMEAN) {
                    const T m = prm.mu[k0 + kc[j]];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] -= m;
                }
                rb[j] = v;
            }
        } else {  // edge tiles: element loads with zero fill
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int kr = k0 + kc[j];
                const int64_t n = n0 + (int64_t)((tid + NTHR * j) % CPR) * VEC;
                const T m = (!ZMEAN && kr < prm.D) ? prm.mu[kr] : T(0);
                vecT v;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    v[e] = (kr < prm.D && n + e < prm.N) ? prm.q[(int64_t)kr * prm.ldq + n + e] - m : T(0);
                rb[j] = v;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = tid + NTHR * j;
            *reinterpret_cast<vecT*>(&As[buf][c * VEC]) = ra[j];
            *reinterpret_cast<vecT*>(&Bs[buf][c * VEC]) = rb[j];
        }
    };

    typename MmaOf<T>::type mma;
    mma.zero();
    const int nk = prm.DPAD / BK;
    // Interior blocks with a zero mean (config C5's every block) fetch their operand tiles by
    // LDS-DMA (buffer_load_dwordx4 ... lds): a tile row is 512 contiguous bytes in HBM and in LDS
    // alike, one wave instruction moves two rows, no VGPR staging, no ds_write (8 x ds_write_b128 per
    // thread and tile went through the LDS store path, which is shared by the CU and not hidden by
    // MFMAs), no vmcnt wait in the middle of the MFMA stream.  PMC before: matrix pipe 77 % busy at
    // 2.39 GHz (the clock was not the limit).
    bool dma = false;
    if constexpr (ZMEAN) {
        dma = !prm.no_dma && cols_full && prm.DPAD == prm.D && (reinterpret_cast<uintptr_t>(prm.PT) & 15) == 0 &&
              (int64_t)prm.DPAD * (prm.ldq > prm.DPAD ? prm.ldq : prm.DPAD) * (int64_t)sizeof(T) < ((int64_t)1 << 32);
    }
    if (dma) {  // block-uniform
        typedef __attribute__((address_space(3))) void lds_void;
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pa0), 0, 0xFFFFFFF0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pb0), 0, 0xFFFFFFF0u, 0x00020000);
        // one wave instruction = 1 KiB = RPI tile rows (2 of 512 B in fp32, 1 of 1 KiB in fp64)
        constexpr int LPR = BM * (int)sizeof(T) / 16;  // lanes per tile row
        constexpr int RPI = 64 / LPR;                  // rows per instruction
        constexpr uint32_t ES = (uint32_t)sizeof(T);
        const uint32_t sub = (uint32_t)(lane / LPR), x16 = 16u * (uint32_t)(lane % LPR);
        const uint32_t vA = sub * (uint32_t)prm.DPAD * ES + x16, vB = sub * (uint32_t)prm.ldq * ES + x16;
        auto dma_tiles = [&](int k0, int buf) {
#pragma unroll
            for (int j = 0; j < BK / RPI / (NTHR / 64); ++j) {  // wave w: row groups w, w + NWAVES, ...
                const int pr = wave + (NTHR / 64) * j;
                const uint32_t row = (uint32_t)(k0 + RPI * pr);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_void*)&As[buf][RPI * pr * BM], 16, vA,
                                                         row * (uint32_t)prm.DPAD * ES, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_void*)&Bs[buf][RPI * pr * BN], 16, vB,
                                                         row * (uint32_t)prm.ldq * ES, 0, 0);
            }
        };
        dma_tiles(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk) dma_tiles((t + 1) * BK, cur ^ 1);  // lands under the MFMAs
            mma.tile(As[cur], Bs[cur], wm, wn, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        load_tiles(0);
        store_tiles(0);
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk) load_tiles((t + 1) * BK);  // global loads in flight under the MFMAs
            mma.tile(As[cur], Bs[cur], wm, wn, lane);
            if (t + 1 < nk) store_tiles(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue, fp32 interior tiles: staged through LDS
    // In the MFMA C/D layout a lane owns 64 scattered elements: 256 four-byte global accesses per
    // lane (q, vh in; vh, q_next out).  The tile goes through LDS instead (the operand buffers are
    // free after the last step) and comes back row-major: thread (rg = tid/32, cg = tid%32) handles
    // columns 4cg..4cg+3 of rows rg, rg+8, ..: 16-byte accesses, 64 per thread, 512 B per row
    // segment.  Matters when K = D is small and the epilogue is a large share of the tile's time.
    // Edge tiles, unaligned strides and fp64 keep the per-element path below.
    if constexpr (sizeof(T) == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        constexpr int CS = BN + 4;  // padded row stride of the staged tile
        constexpr int RGS = NTHR / 32;  // row groups
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        bool staged = (i0 + BM <= prm.D) && (n0 + BN <= prm.N) && q_vec_ok;
#ifdef PBBI_BIG_NO_STAGED
        staged = false;
#endif
        staged = staged && (!prm.grad_out || (prm.ldg % 4 == 0 && al16(prm.grad_out)));
        if constexpr (EPI == EPI_KDK)
            staged = staged && prm.ldw % 4 == 0 && al16(prm.vh) && (!prm.q_next || al16(prm.q_next)) &&
                     (!prm.minv || al16(prm.minv));
        if (staged) {  // block-uniform
            float* Cst = reinterpret_cast<float*>(smem_big);  // [BM][CS]
            float* red2 = Cst + BM * CS;                      // [RGS][BN]
            mma.each(wm, wn, lane, [&](int il, int nl, int, T g) { Cst[il * CS + nl] = g; });
            __syncthreads();
            const int cg = tid & 31, rg = tid >> 5;
            const int64_t n = n0 + 4 * cg;
            f4 xg4 = {0.f, 0.f, 0.f, 0.f};
            f4 mi4 = {1.f, 1.f, 1.f, 1.f};
            if constexpr (EPI == EPI_KDK)
                if (prm.minv) mi4 = *reinterpret_cast<const f4*>(prm.minv + n);
#pragma unroll 4
            for (int pss = 0; pss < BM / RGS; ++pss) {
                const int il = pss * RGS + rg;
                const int i = i0 + il;
                const f4 g = *reinterpret_cast<const f4*>(&Cst[il * CS + 4 * cg]);
                const f4 qv = *reinterpret_cast<const f4*>(prm.q + (int64_t)i * prm.ldq + n);
                const f4 x = ZMEAN ? qv : qv - prm.mu[i];
                xg4 += x * g;
                if constexpr (EPI == EPI_EVAL) {
                    if (prm.grad_out) *reinterpret_cast<f4*>(prm.grad_out + (int64_t)i * prm.ldg + n) = g;
                } else {
                    const int64_t o = (int64_t)i * prm.ldw + n;
                    const f4 v = *reinterpret_cast<const f4*>(prm.vh + o) + (-(g * mi4)) * prm.hk;  // kick
                    *reinterpret_cast<f4*>(prm.vh + o) = v;
                    if (prm.q_next) *reinterpret_cast<f4*>(prm.q_next + o) = qv + v * prm.h;  // drift
                    if (prm.grad_out) *reinterpret_cast<f4*>(prm.grad_out + (int64_t)i * prm.ldg + n) = g;  // carried
                }
            }
            if (prm.xg_part) {
                *reinterpret_cast<f4*>(&red2[rg * BN + 4 * cg]) = xg4;
                __syncthreads();
                if (tid < BN) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < RGS; ++r) s += red2[r * BN + tid];
                    prm.xg_part[(size_t)bm * prm.N + n0 + tid] = s;
                }
            }
            return;
        }
    }

    // ---- epilogue, fp64 interior tiles: the same staging in two halves of 64 rows (64 KiB each, the
    // size of the operand buffers), 16-byte (two-element) accesses
    if constexpr (sizeof(T) == 8) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        constexpr int RGS = NTHR / 64;  // row groups of the read phase (64 column pairs per row)
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        bool staged = (i0 + BM <= prm.D) && (n0 + BN <= prm.N) && q_vec_ok;
        staged = staged && (!prm.grad_out || (prm.ldg % 2 == 0 && al16(prm.grad_out)));
        if constexpr (EPI == EPI_KDK)
            staged = staged && prm.ldw % 2 == 0 && al16(prm.vh) && (!prm.q_next || al16(prm.q_next)) &&
                     (!prm.minv || al16(prm.minv));
        if (staged) {  // block-uniform
            double* Cst = reinterpret_cast<double*>(smem_big);  // [64][BN]
            const int cg = tid & 63, rg = tid >> 6;
            const int64_t n = n0 + 2 * cg;
            d2 xg2 = {0.0, 0.0};
            d2 mi2 = {1.0, 1.0};
            if constexpr (EPI == EPI_KDK)
                if (prm.minv) mi2 = *reinterpret_cast<const d2*>(prm.minv + n);
#pragma unroll 1
            for (int hrow = 0; hrow < BM; hrow += 64) {
                if (hrow) __syncthreads();  // the previous half has been read
                if (wm == hrow)
                    mma.each(wm, wn, lane, [&](int il, int nl, int, T g) { Cst[(il - hrow) * BN + nl] = g; });
                __syncthreads();
#pragma unroll 4
                for (int pss = 0; pss < 64 / RGS; ++pss) {
                    const int il = pss * RGS + rg;
                    const int i = i0 + hrow + il;
                    const d2 g = *reinterpret_cast<const d2*>(&Cst[il * BN + 2 * cg]);
                    const d2 qv = *reinterpret_cast<const d2*>(prm.q + (int64_t)i * prm.ldq + n);
                    const d2 x = ZMEAN ? qv : qv - prm.mu[i];
                    xg2 += x * g;
                    if constexpr (EPI == EPI_EVAL) {
                        if (prm.grad_out) *reinterpret_cast<d2*>(prm.grad_out + (int64_t)i * prm.ldg + n) = g;
                    } else {
                        const int64_t o = (int64_t)i * prm.ldw + n;
                        const d2 v = *reinterpret_cast<const d2*>(prm.vh + o) + (-(g * mi2)) * prm.hk;  // kick
                        *reinterpret_cast<d2*>(prm.vh + o) = v;
                        if (prm.q_next) *reinterpret_cast<d2*>(prm.q_next + o) = qv + v * prm.h;  // drift
                        if (prm.grad_out) *reinterpret_cast<d2*>(prm.grad_out + (int64_t)i * prm.ldg + n) = g;  // carried
                    }
                }
            }
            if (prm.xg_part) {
                __syncthreads();  // the staged tile is dead: its space takes the partial sums [RGS][BN]
                *reinterpret_cast<d2*>(&Cst[rg * BN + 2 * cg]) = xg2;
                __syncthreads();
                if (tid < BN) {
                    double s2 = 0.0;
#pragma unroll
                    for (int r = 0; r < RGS; ++r) s2 += Cst[r * BN + tid];
                    prm.xg_part[(size_t)bm * prm.N + n0 + tid] = s2;
                }
            }
            return;
        }
    }

    // ---- epilogue, general path
    // both C/D maps give a lane one fixed column per column-tile tb and several rows:
    // accumulate x.g per tb, then combine the lanes that share the column
    constexpr int NTB = 2;  // column tiles per wave: 2 x 32 (fp32) or 2 x 16 (fp64)
    T xg[NTB];
#pragma unroll
    for (int b = 0; b < NTB; ++b) xg[b] = T(0);
    mma.each(wm, wn, lane, [&](int il, int nl, int tb, T g) {
        const int i = i0 + il;
        const int64_t n = n0 + nl;
        if (i < prm.D && n < prm.N) {
            const T qv = prm.q[(int64_t)i * prm.ldq + n];
            xg[tb] += (ZMEAN ? qv : qv - prm.mu[i]) * g;
            if constexpr (EPI == EPI_EVAL) {
                if (prm.grad_out) prm.grad_out[(int64_t)i * prm.ldg + n] = g;
            } else {
                const T mi = prm.minv ? prm.minv[n] : T(1);
                const int64_t o = (int64_t)i * prm.ldw + n;
                const T v = prm.vh[o] + (-(g * mi)) * prm.hk;  // kick
                prm.vh[o] = v;
                if (prm.q_next) prm.q_next[o] = qv + v * prm.h;  // drift into the other buffer
                if (prm.grad_out) prm.grad_out[(int64_t)i * prm.ldg + n] = g;  // carried to the next iteration
            }
        }
    });
    if (prm.xg_part) {
        // lanes that share a column: f32 map -> lanes l and l+32; f64 map -> l, l+16, l+32, l+48
#pragma unroll
        for (int b = 0; b < NTB; ++b) {
            T s = xg[b];
            if constexpr (sizeof(T) == 4) {
                s += __shfl_xor(s, 32, 64);
            } else {
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
            }
            const int nl = wn + b * (sizeof(T) == 4 ? 32 : 16) + (lane & (sizeof(T) == 4 ? 31 : 15));
            const bool writer = sizeof(T) == 4 ? (lane < 32) : (lane < 16);
            if (writer) red[wave / WAVES_N][nl] = s;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < prm.N)
            prm.xg_part[(size_t)bm * prm.N + n0 + tid] = red[0][tid] + red[1][tid];
    }
}

// ---- fp32, wide block tile -------------------------------------------------------------------------
// Interior of a zero-mean fp32 problem (config C5's every block): the same GEMM + fused epilogue on a
// 256-row block tile.  What limits k_big_gemm at 128 x 128 is the operand traffic, not the matrix pipe
// (ablation, DESIGN.md 4.4: the tile fills cost 11 %, the LDS operand reads 6 %): per MFMA a 128 x 128
// block moves 1/64 of a tile row pair from L2 into LDS and a 64 x 64 wave tile reads one operand value
// per MFMA.  Here a block is BM x BN = 256 x (128 | 256) and a wave 128 x (64 | 128): 0.75x / 0.5x the
// fill bytes and 0.75x / 0.5x the operand reads per MFMA.  Tiles arrive by LDS-DMA only (one 1-KiB wave
// instruction per 256-float row), epilogue staged through the (then free) operand buffers 32 rows per
// wave row at a time.  Anything that is not a whole aligned tile goes to k_big_gemm.
template <int WAVES_M_, int WAVES_N_, int TN_, int BK_, int MINW_, int NST_ = 2>
struct WideCfg {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = 4, TN = TN_, BK = BK_, MINW = MINW_;
    static constexpr int NST = NST_;                        // operand stages in LDS (tiles in flight: NST - 1)
    static constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32, NW = WAVES_M * WAVES_N, NTHR = 64 * NW;
    static constexpr int CS = BN + 4;                       // padded row stride of the staged tile
    static constexpr int RGS = NTHR / (BN / 4);             // row groups of the epilogue's read phase
    static constexpr size_t LDS_MAIN = (size_t)NST * BK * (BM + BN) * sizeof(float);
    static constexpr size_t LDS_EPI = ((size_t)WAVES_M * 32 * CS + (size_t)RGS * BN) * sizeof(float);
    static constexpr size_t LDS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
    static_assert(TM * 32 == 128, "one wave row = one 128-row slab of xg_part");
    static_assert(BM == 256 && (BN == 128 || BN == 256), "row = 1 KiB (A) and 512 B / 1 KiB (B)");
};

#ifndef PBBI_WIDE_ABLATE
#define PBBI_WIDE_ABLATE 0
#endif
template <class C, int EPI>
__global__ void __launch_bounds__(C::NTHR, C::MINW) k_big_gemm_wide(GemmPrm<float> prm) {
    constexpr int BMW = C::BM, BNW = C::BN, BK = C::BK, TM = C::TM, TN = C::TN, NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) char smem_big[];
    constexpr int NST = C::NST;
    float* const As = reinterpret_cast<float*>(smem_big);            // [NST][BK][BMW]
    float* const Bs = As + NST * BK * BMW;                           // [NST][BK][BNW]
    // (readfirstlane: the wave index is uniform, and an LDS-DMA destination the compiler cannot prove
    // uniform costs a waterfall loop per instruction)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wmi = wave / C::WAVES_N, wm = wmi * (TM * 32), wn = (wave % C::WAVES_N) * (TN * 32);
    const int n_tiles_m = prm.D / BMW;
    const int bm = blockIdx.x % n_tiles_m, bn = blockIdx.x / n_tiles_m;
    const int i0 = bm * BMW;
    const int64_t n0 = (int64_t)bn * BNW;

    typedef __attribute__((address_space(3))) void lds_void;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.PT + i0), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.q + n0), 0, 0xFFFFFFF0u, 0x00020000);
    constexpr int LPR_B = BNW / 4, RPI_B = 64 / LPR_B;   // lanes per B row, B rows per wave instruction
    const uint32_t vA = 16u * (uint32_t)lane;
    const uint32_t vB = (uint32_t)(lane / LPR_B) * (uint32_t)prm.ldq * 4u + 16u * (uint32_t)(lane % LPR_B);
    auto dma_tiles = [&](int k0, int buf) {
#pragma unroll
        for (int j = 0; j < BK / NW; ++j) {  // A: one row per instruction
            const int row = wave + NW * j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_void*)(As + (buf * BK + row) * BMW), 16, vA,
                                                     (uint32_t)(k0 + row) * (uint32_t)prm.DPAD * 4u, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BK / RPI_B / NW; ++j) {
            const int row = RPI_B * (wave + NW * j);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_void*)(Bs + (buf * BK + row) * BNW), 16, vB,
                                                     (uint32_t)(k0 + row) * (uint32_t)prm.ldq * 4u, 0, 0);
        }
    };

    v16f32 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int r32 = lane & 31, kh = lane >> 5;
    const int nk = prm.D / BK;
    // NST - 1 tiles in flight; a wave waits for its own share of the oldest one (its DMA instructions
    // retire in order: vmcnt <= the instructions of the younger tiles) and the barrier publishes all shares
    constexpr int DMA_PER_TILE = BK / NW + BK / RPI_B / NW;
#pragma unroll
    for (int pre = 0; pre < NST - 1; ++pre) dma_tiles(pre * BK, pre);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DMA_PER_TILE) : "memory");
    __syncthreads();
    int cur = 0, nxt = NST - 1;  // buffer of tile t, buffer tile t + NST - 1 goes to
    for (int t = 0; t < nk; ++t) {
#if !(PBBI_WIDE_ABLATE & 1)   // (bit 0: no tile fills after the prologue -- timing experiment, wrong results)
        if (t + NST - 1 < nk) dma_tiles((t + NST - 1) * BK, nxt);  // lands under the MFMAs
#endif
        // operands of K-step s+1 are read before the TM*TN MFMAs of K-step s issue (see MmaF32::tile)
        const float* ap = As + (cur * BK + kh) * BMW + wm + r32;
        const float* bp = Bs + (cur * BK + kh) * BNW + wn + r32;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int x = 0; x < TM; ++x) a[0][x] = ap[32 * x];
#pragma unroll
        for (int x = 0; x < TN; ++x) b[0][x] = bp[32 * x];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int c = s & 1, nx = c ^ 1;
#if PBBI_WIDE_ABLATE & 2   // (bit 1: operands read once per tile -- timing experiment, wrong results)
            if (s == 0) {
#else
            if (s + 1 < BK / 2) {
#endif
#pragma unroll
                for (int x = 0; x < TM; ++x) a[nx][x] = ap[2 * (s + 1) * BMW + 32 * x];
#pragma unroll
                for (int x = 0; x < TN; ++x) b[nx][x] = bp[2 * (s + 1) * BNW + 32 * x];
            }
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][ta], b[c][tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NST > 2 && t + NST - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }

#if PBBI_WIDE_ABLATE & 4   // (bit 2: no epilogue -- timing experiment, wrong results; one store keeps the MFMAs alive)
    {
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[a][b][r];
        if (keep == 12345.678f) prm.vh[tid] = keep;
        return;
    }
#endif
    // ---- epilogue: TM passes; in pass ta every wave stages its row tile ta (32 rows x TN*32 columns),
    // then the block reads the WAVES_M*32 staged rows back row-major, 16 bytes per access
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int CS = C::CS, RGS = C::RGS, CPRW = BNW / 4;
    float* const Cst = reinterpret_cast<float*>(smem_big);   // [WAVES_M*32][CS]
    float* const red2 = Cst + C::WAVES_M * 32 * CS;           // [RGS][BNW]
    const int cg = tid % CPRW, rg = tid / CPRW;
    const int64_t n = n0 + 4 * cg;
    f4 xg4[C::WAVES_M];
#pragma unroll
    for (int w = 0; w < C::WAVES_M; ++w) xg4[w] = f4{0.f, 0.f, 0.f, 0.f};
    f4 mi4 = {1.f, 1.f, 1.f, 1.f};
    if constexpr (EPI == EPI_KDK)
        if (prm.minv) mi4 = *reinterpret_cast<const f4*>(prm.minv + n);
#pragma unroll
    for (int ta = 0; ta < TM; ++ta) {
        if (ta) __syncthreads();  // the previous pass has been read
#pragma unroll
        for (int tb = 0; tb < TN; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r)  // C/D map of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                Cst[(wmi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * CS + wn + 32 * tb + r32] = acc[ta][tb][r];
        __syncthreads();
#pragma unroll
        for (int w = 0; w < C::WAVES_M; ++w) {
#pragma unroll 4
            for (int pss = 0; pss < 32 / RGS; ++pss) {
                const int rr = pss * RGS + rg;                   // row of wave row w's staged tile
                const int i = i0 + w * (TM * 32) + ta * 32 + rr;
                const f4 g = *reinterpret_cast<const f4*>(&Cst[(w * 32 + rr) * CS + 4 * cg]);
                const f4 qv = *reinterpret_cast<const f4*>(prm.q + (int64_t)i * prm.ldq + n);
                xg4[w] += qv * g;
                if constexpr (EPI == EPI_EVAL) {
                    if (prm.grad_out) *reinterpret_cast<f4*>(prm.grad_out + (int64_t)i * prm.ldg + n) = g;
                } else {
                    const int64_t o = (int64_t)i * prm.ldw + n;
                    const f4 v = *reinterpret_cast<const f4*>(prm.vh + o) + (-(g * mi4)) * prm.hk;  // kick
                    *reinterpret_cast<f4*>(prm.vh + o) = v;
                    if (prm.q_next) *reinterpret_cast<f4*>(prm.q_next + o) = qv + v * prm.h;  // drift
                    if (prm.grad_out) *reinterpret_cast<f4*>(prm.grad_out + (int64_t)i * prm.ldg + n) = g;  // carried
                }
            }
        }
    }
    if (prm.xg_part) {
#pragma unroll
        for (int w = 0; w < C::WAVES_M; ++w) {
            __syncthreads();
            *reinterpret_cast<f4*>(&red2[rg * BNW + 4 * cg]) = xg4[w];
            __syncthreads();
            if (tid < BNW) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < RGS; ++r) s += red2[r * BNW + tid];
                prm.xg_part[(size_t)(i0 / BM + w) * prm.N + n0 + tid] = s;
            }
        }
    }
}

#ifndef PBBI_BIG_WIDE
#define PBBI_BIG_WIDE 3   // 0: off; 1: 256 x 256, 8 waves of 128 x 64; 2: 256 x 256, 4 waves of 128 x 128; 3: 256 x 128, 4 waves of 128 x 64, two blocks per CU
#endif
#if PBBI_BIG_WIDE == 1
typedef WideCfg<2, 4, 2, 32, 1> Wide;
#elif PBBI_BIG_WIDE == 2
typedef WideCfg<2, 2, 4, 32, 1> Wide;
#elif PBBI_BIG_WIDE == 3
typedef WideCfg<2, 2, 2, 16, 2> Wide;
#elif PBBI_BIG_WIDE == 4   // 3 with three operand stages (72 KiB per block)
typedef WideCfg<2, 2, 2, 16, 2, 3> Wide;
#elif PBBI_BIG_WIDE == 5   // 3 with four stages of 8 rows (48 KiB per block)
typedef WideCfg<2, 2, 2, 8, 2, 4> Wide;
#endif

// ---- elementwise helpers ---------------------------------------------------------------------------
// column sums over row blocks: part[rb][n] = sum_{d in block rb} f(a[d][n]) with f = square.
template <typename T>
__global__ void k_big_sq_partial(const T* a, int64_t ld, int D, int64_t N, int rows_per_block,
                                 T* part) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int d0 = blockIdx.y * rows_per_block;
    const int d1 = d0 + rows_per_block < D ? d0 + rows_per_block : D;
    T s = T(0);
    for (int d = d0; d < d1; ++d) {
        const T v = a[(int64_t)d * ld + n];
        s += v * v;
    }
    part[(size_t)blockIdx.y * N + n] = s;
}

// momentum: p -> vh = p * minv (parity mode: p_in given) or Philox draw (p written to vh first).
template <typename T>
__global__ void k_big_momentum(const T* p_in, int64_t ldp, T* vh, T* p_keep, int64_t ldw, int D,
                               int64_t N, const T* mass, int rng, uint64_t seed, uint64_t iter,
                               uint64_t chain0, double kT, int draw64) {
    // one thread per (block of 4 dims sharing a Philox block, chain)
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int blk = blockIdx.y;  // Philox block index: dims 16*(blk>>2) + (blk&3) + 4*slot
    if (n >= N) return;
    const double m = mass ? (double)mass[n] : 1.0;
    const double pstd = sqrt(m * kT);
    double z[4];
    if (rng) rng_normal4d(seed, PBBI_STREAM_MOMENTUM, iter, chain0 + (uint64_t)n, (uint32_t)blk, draw64 != 0, z);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const int d = 16 * (blk >> 2) + (blk & 3) + 4 * sl;
        if (d < D) {
            const T p = rng ? (T)(z[sl] * pstd) : p_in[(int64_t)d * ldp + n];
            if (p_keep) p_keep[(int64_t)d * ldw + n] = p;  // drawn momentum (needed for pp_old / rejects)
            vh[(int64_t)d * ldw + n] = mass ? (T)(p * (T)(1.0 / m)) : p;
        }
    }
}

// decision: H_old, H_new from the partial sums; ratio; reject flag; uniform from Philox or u_in.
template <typename T>
__global__ void k_big_decide(const T* pp_old_part, const T* pp_new_part, int n_sq_parts,
                             const T* xg_old_part, const T* xg_new_part, int n_xg_parts,
                             const T* mass, const T* u_in, int rng, uint64_t seed, uint64_t iter,
                             uint64_t chain0, T cst, T beta, int64_t N, T* ratio_out, uint8_t* reject) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    T po = T(0), pn = T(0), xo = T(0), xn = T(0);
    for (int b = 0; b < n_sq_parts; ++b) {
        po += pp_old_part[(size_t)b * N + n];
        pn += pp_new_part[(size_t)b * N + n];
    }
    for (int b = 0; b < n_xg_parts; ++b) {
        xo += xg_old_part[(size_t)b * N + n];
        xn += xg_new_part[(size_t)b * N + n];
    }
    const T m = mass ? mass[n] : T(1);
    const T oldH = T(0.5) * po / m + (T(0.5) * xo + cst);
    const T newH = T(0.5) * pn / m + (T(0.5) * xn + cst);
    const T ratio = exp((oldH - newH) * beta);
    const T u = rng ? (T)rng_uniform(seed, iter, chain0 + (uint64_t)n) : u_in[n];
    const bool rej = (ratio == ratio) && (u > (ratio < T(1) ? ratio : T(1)));
    reject[n] = rej ? 1 : 0;
    if (ratio_out) ratio_out[n] = ratio;
}

// Carried gradient (run_hmc): the first half kick and drift of an iteration from the gradient the previous
// iteration left -- G_new where the chain accepted, G_keep where it did not -- instead of a GEMM:
//     vh += -(g/m) * hk;  q_next = q + vh * h        (the GEMM epilogue's arithmetic, term for term)
// and G_keep takes the gradient that is current now.
template <typename T>
__global__ void k_big_first_kick(const T* q, int64_t ldq, T* G_keep, const T* G_new, const uint8_t* acc_prev,
                                 T* vh, const T* minv, T* q_next, int64_t ldw, int D, int64_t N, T hk, T h) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const bool acc = acc_prev[n] != 0;
    const T mi = minv ? minv[n] : T(1);
    const int d0 = blockIdx.y * 16;
    for (int d = d0; d < d0 + 16 && d < D; ++d) {
        const int64_t o = (int64_t)d * N + n;
        const T g = acc ? G_new[o] : G_keep[o];
        if (acc) G_keep[o] = g;
        const int64_t w = (int64_t)d * ldw + n;
        const T v = vh[w] + (-(g * mi)) * hk;
        vh[w] = v;
        q_next[w] = q[(int64_t)d * ldq + n] + v * h;
    }
}

// after the decision: which gradient / which x.g partial sums are current for the next iteration
template <typename T>
__global__ void k_big_carry_update(const uint8_t* reject, uint8_t* acc_prev, const T* xg_old, const T* xg_new,
                                   T* xg_cur, int n_xg, int64_t N) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const bool rej = reject[n] != 0;
    acc_prev[n] = rej ? 0 : 1;
    for (int b = 0; b < n_xg; ++b) {
        const size_t o = (size_t)b * N + n;
        xg_cur[o] = rej ? xg_old[o] : xg_new[o];
    }
}

// p_new = v_L * m (into pbuf, in place) -- used before the pp_new reduction
template <typename T>
__global__ void k_big_v_to_p(T* v, int64_t ld, int D, int64_t N, const T* mass) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (n < N && d < D && mass) v[(int64_t)d * ld + n] *= mass[n];
}

// select: q_out = reject ? q_old : q_new;  p_out = reject ? (compat ? q_old : p_draw) : p_new
template <typename T>
__global__ void k_big_select(const T* q_old, int64_t ldq, const T* q_new, const T* p_new,
                             const T* p_draw, int64_t ldw, const uint8_t* reject, int compat,
                             T* q_out, T* p_out, int64_t ldo, int D, int64_t N) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (n >= N || d >= D) return;
    const bool rej = reject[n] != 0;
    const T qo = q_old[(int64_t)d * ldq + n];
    q_out[(int64_t)d * ldo + n] = rej ? qo : q_new[(int64_t)d * ldw + n];
    if (p_out)
        p_out[(int64_t)d * ldo + n] =
            rej ? (compat ? qo : p_draw[(int64_t)d * ldw + n]) : p_new[(int64_t)d * ldw + n];
}

template <typename T>
__global__ void k_big_copy(const T* src, int64_t lds, T* dst, int64_t ldd, int D, int64_t N, T scale,
                           const T* scale_n) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (n < N && d < D)
        dst[(int64_t)d * ldd + n] = src[(int64_t)d * lds + n] * (scale_n ? scale_n[n] : scale);
}

template <typename T>
__global__ void k_big_finish_eval(const T* xg_part, int n_parts, const T* pp_part, int n_sq_parts,
                                  const T* mass, T cst, int64_t N, T* U_out, T* w_out, int mode) {
    // mode 0: U_out = 0.5 xg + cst; 1: H (and w = exp(-H)); 2: U_out = exp(U_out - H)
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    T xg = T(0), pp = T(0);
    for (int b = 0; b < n_parts; ++b) xg += xg_part[(size_t)b * N + n];
    const T U = T(0.5) * xg + cst;
    if (mode == 0) {
        if (U_out) U_out[n] = U;
        return;
    }
    for (int b = 0; b < n_sq_parts; ++b) pp += pp_part[(size_t)b * N + n];
    const T H = T(0.5) * pp / (mass ? mass[n] : T(1)) + U;
    if (mode == 1) {
        if (U_out) U_out[n] = H;
        if (w_out) w_out[n] = exp(-H);
    } else {
        U_out[n] = exp(U_out[n] - H);
    }
}

template <typename T>
__global__ void k_big_inv(const T* mass, T* minv, int64_t N) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) minv[n] = T(1) / mass[n];
}

// ---- host side -------------------------------------------------------------------------------------

constexpr int SQ_ROWS = 256;  // rows per block of the p^2 column reduction

template <typename T>
int gemm(const pbbi_potential* pot, int epi, const T* q, int64_t ldq, T* q_next, T* vh, int64_t ldw,
         const T* minv, T* grad_out, int64_t ldg, T* xg_part, int64_t N, T hk, T h, hipStream_t st) {
    GemmPrm<T> prm{(const T*)pot->d_big_PT, (const T*)pot->d_big_mu, q, q_next, vh, minv, grad_out,
                   xg_part, N, ldq, ldw, ldg, pot->D, pot->DPAD_big, hk, h, 0};
    static const bool no_dma = (getenv("PBBI_BIG_NO_DMA") != nullptr);
    prm.no_dma = no_dma ? 1 : 0;
#if PBBI_BIG_WIDE
    if constexpr (sizeof(T) == 4) {
        // the wide-tile kernel: whole aligned tiles of a zero-mean problem only (PBBI_BIG_TILE=128 forces
        // the 128 x 128 kernel, for A/B measurements and tests)
        static const bool narrow = [] { const char* e = getenv("PBBI_BIG_TILE"); return e && atoi(e) == 128; }();
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        bool ok = !narrow && pot->zero_mean && pot->D == pot->DPAD_big && pot->D % Wide::BM == 0 && N % Wide::BN == 0 &&
                  ldq % 4 == 0 && al16(q) && al16(prm.PT) &&
                  (int64_t)pot->D * (ldq > pot->D ? ldq : (int64_t)pot->D) * 4 < ((int64_t)1 << 32);
        ok = ok && (!grad_out || (ldg % 4 == 0 && al16(grad_out)));
        if (epi == EPI_KDK) ok = ok && ldw % 4 == 0 && al16(vh) && (!q_next || al16(q_next)) && (!minv || al16(minv));
        if (ok) {
            const unsigned tiles_w = (unsigned)((pot->D / Wide::BM) * (N / Wide::BN));
            auto gow = [&](auto kernel) -> int {
                PBBI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide::LDS));
                hipLaunchKernelGGL(kernel, dim3(tiles_w), dim3(Wide::NTHR), Wide::LDS, st, prm);
                return PBBI_OK;
            };
            if (int rc = (epi == EPI_EVAL) ? gow(k_big_gemm_wide<Wide, EPI_EVAL>) : gow(k_big_gemm_wide<Wide, EPI_KDK>))
                return rc;
            PBBI_HIP(hipGetLastError());
            return PBBI_OK;
        }
    }
#endif
    const unsigned tiles = (unsigned)((pot->DPAD_big / BM) * ((N + BN - 1) / BN));
    size_t lds = (size_t)4 * Cfg<T>::BK * BM * sizeof(T) + 2 * BN * sizeof(T);
#ifndef PBBI_BIG_NO_STAGED
    if (sizeof(T) == 4) {  // the staged fp32 epilogue: [BM][BN+4] tile + [NTHR/32][BN] partial sums
        const size_t staged = (size_t)BM * (BN + 4) * sizeof(T) + (size_t)(Cfg<T>::NTHR / 32) * BN * sizeof(T);
        if (staged > lds) lds = staged;
    }
#endif
    auto go = [&](auto kernel) -> int {
        PBBI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(tiles), dim3(Cfg<T>::NTHR), lds, st, prm);
        return PBBI_OK;
    };
    int rc;
    if (epi == EPI_EVAL)
        rc = pot->zero_mean ? go(k_big_gemm<T, EPI_EVAL, true>) : go(k_big_gemm<T, EPI_EVAL, false>);
    else
        rc = pot->zero_mean ? go(k_big_gemm<T, EPI_KDK, true>) : go(k_big_gemm<T, EPI_KDK, false>);
    if (rc) return rc;
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

inline dim3 grid2d(int64_t N, int D) { return dim3((unsigned)((N + 255) / 256), (unsigned)D); }

template <typename T>
int run_hmc(const IterArgs& a) {
    const pbbi_potential* pot = a.pot;
    const int D = pot->D, L = a.L;
    const int64_t N = a.N;
    hipStream_t st = a.stream;
    Scratch ws(a);
    const size_t slab = (size_t)D * N * sizeof(T);
    const int n_xg = pot->DPAD_big / BM, n_sq = (D + SQ_ROWS - 1) / SQ_ROWS;
    T* vh = (T*)ws.get(slab);
    T* qa = (T*)ws.get(slab);
    T* qb = (T*)ws.get(slab);
    T* pdraw = (T*)ws.get(slab);
    T* xg_old = (T*)ws.get((size_t)n_xg * N * sizeof(T));
    T* xg_new = (T*)ws.get((size_t)n_xg * N * sizeof(T));
    T* pp_old = (T*)ws.get((size_t)n_sq * N * sizeof(T));
    T* pp_new = (T*)ws.get((size_t)n_sq * N * sizeof(T));
    uint8_t* rej = a.reject_out ? a.reject_out : (uint8_t*)ws.get((size_t)N);
    T* minv = a.mass ? (T*)ws.get((size_t)N * sizeof(T)) : nullptr;
    if (!vh || !qa || !qb || !pdraw || !xg_old || !xg_new || !pp_old || !pp_new || !rej ||
        (a.mass && !minv))
        return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the large-D workspace");
    const dim3 b1(256), g1((unsigned)((N + 255) / 256));
    if (minv) hipLaunchKernelGGL(k_big_inv<T>, g1, b1, 0, st, (const T*)a.mass, minv, N);
    // momentum (drawn or uploaded) -> pdraw, vh = p/m; pp_old partials
    const int n_blk = ((D + 15) / 16) * 4;
    hipLaunchKernelGGL(k_big_momentum<T>, dim3(g1.x, (unsigned)n_blk), b1, 0, st, (const T*)a.p_in,
                       a.ldn_in, vh, pdraw, N, D, N, (const T*)a.mass, a.rng, a.seed, a.iter,
                       a.chain0, a.kT, (a.flags & PBBI_DRAW_F64) ? 1 : 0);
    hipLaunchKernelGGL(k_big_sq_partial<T>, dim3(g1.x, (unsigned)n_sq), b1, 0, st, (const T*)pdraw,
                       (int64_t)N, D, N, SQ_ROWS, pp_old);
    const T h = (T)a.h, hh = (T)(0.5 * a.h);
    // Leapfrog: L + 1 GEMMs (half kick, L-1 kicks, half kick; L drifts).  Stormer-Verlet
    // (src/integrator.py:142-163) is the same recurrence without the closing half kick and with one
    // more drift -- d = q_n - q_{n-1} = vh*h -- so every GEMM kicks and drifts (L + 1 of each), the
    // final velocity is vh, and U of the final position takes one more (evaluation) GEMM.
    // q_0 is the caller's q_in, then ping-pong between qa and qb.
    const bool sv = (a.method == PBBI_STORMER_VERLET);
    const T* qcur = (const T*)a.q_in;
    int64_t ldcur = a.ldn_in;
    const T* xg_fin = xg_new;
    // Carried gradient (pbbi_hmc_run, IterArgs::carry): the last GEMM of an iteration is g(q_new) and the
    // next iteration starts from q_new (accepted) or from where this one started (rejected) -- either way
    // from a point whose gradient and x.g sums exist.  They are kept (two gradient slabs, one byte and the
    // partial sums per chain) and the first of the L + 1 GEMMs becomes one elementwise pass.  The decisions
    // use the very sums the GEMM epilogue would form again, so a run's samples do not change.
    const int carry = (a.carry && a.carry_g && big_carry_applies(a)) ? a.carry : 0;
    T* G_keep = carry ? (T*)a.carry_g : nullptr;
    T* G_new = carry ? G_keep + (size_t)D * N : nullptr;
    uint8_t* acc_prev = carry ? a.carry_sel : nullptr;
    T* xg_cur = carry ? (T*)(a.carry_sel + ((size_t)N + 255) / 256 * 256) : nullptr;
    if (!sv && L == 0) {  // nothing moves: one evaluation serves both Hamiltonians
        if (int rc = gemm<T>(pot, EPI_EVAL, qcur, ldcur, nullptr, nullptr, 0, nullptr, nullptr, 0,
                             xg_old, N, T(0), T(0), st))
            return rc;
        xg_fin = xg_old;
        hipLaunchKernelGGL(k_big_copy<T>, grid2d(N, D), b1, 0, st, qcur, ldcur, qa, (int64_t)N, D, N,
                           T(1), (const T*)nullptr);  // the select kernel reads stride-N workspaces
        qcur = qa;
    } else {
        for (int j = 0; j <= L; ++j) {
            const bool first = (j == 0), last = (j == L), drift = sv || !last;
            T* qnext = drift ? ((j & 1) ? qb : qa) : nullptr;
            if (first && carry == 2) {  // the gradient at q_0 exists: no GEMM (xg_old = the carried partial sums)
                hipLaunchKernelGGL(k_big_first_kick<T>, dim3(g1.x, (unsigned)((D + 15) / 16)), b1, 0, st, qcur,
                                   ldcur, G_keep, (const T*)G_new, (const uint8_t*)acc_prev, vh, (const T*)minv,
                                   qnext, (int64_t)N, D, N, hh, h);
            } else {
                // a carried run keeps g(q_0) of its first iteration and g(q_L) of every iteration
                T* gkeep = (carry && first) ? G_keep : ((carry && last) ? G_new : nullptr);
                if (int rc = gemm<T>(pot, EPI_KDK, qcur, ldcur, qnext, vh, N, minv, gkeep, N,
                                     first ? xg_old : ((last && !sv) ? xg_new : nullptr), N,
                                     (first || (last && !sv)) ? hh : h, h, st))
                    return rc;
            }
            if (drift) { qcur = qnext; ldcur = N; }
        }
        if (sv)
            if (int rc = gemm<T>(pot, EPI_EVAL, qcur, ldcur, nullptr, nullptr, 0, nullptr, nullptr, 0,
                                 xg_new, N, T(0), T(0), st))
                return rc;
    }
    hipLaunchKernelGGL(k_big_v_to_p<T>, grid2d(N, D), b1, 0, st, vh, (int64_t)N, D, N, (const T*)a.mass);
    hipLaunchKernelGGL(k_big_sq_partial<T>, dim3(g1.x, (unsigned)n_sq), b1, 0, st, (const T*)vh,
                       (int64_t)N, D, N, SQ_ROWS, pp_new);
    const T* xg_start = (carry == 2) ? (const T*)xg_cur : (const T*)xg_old;
    hipLaunchKernelGGL(k_big_decide<T>, g1, b1, 0, st, (const T*)pp_old, (const T*)pp_new, n_sq,
                       xg_start, xg_fin, n_xg, (const T*)a.mass, (const T*)a.u_in,
                       a.rng, a.seed, a.iter, a.chain0, (T)pot->cst, (T)pbbi_accept_beta(a.flags, a.kT), N,
                       (T*)a.ratio_out, rej);
    if (carry)
        hipLaunchKernelGGL(k_big_carry_update<T>, g1, b1, 0, st, (const uint8_t*)rej, acc_prev, xg_start, xg_fin,
                           xg_cur, n_xg, N);
    hipLaunchKernelGGL(k_big_select<T>, grid2d(N, D), b1, 0, st, (const T*)a.q_in, a.ldn_in, qcur,
                       (const T*)vh, (const T*)pdraw, (int64_t)N, (const uint8_t*)rej,
                       (a.flags & PBBI_COMPAT_P_FROM_OLDQ) ? 1 : 0, (T*)a.q_out, (T*)a.p_out,
                       a.ldn_out, D, N);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

template <typename T>
int run_integrate(const IntegrateArgs& a) {
    const pbbi_potential* pot = a.pot;
    const int D = pot->D, L = a.L;
    const int64_t N = a.N;
    hipStream_t st = a.stream;
    Scratch ws(st);
    const size_t slab = (size_t)D * N * sizeof(T);
    T* vh = (T*)ws.get(slab);
    T* qa = (T*)ws.get(slab);
    T* qb = (T*)ws.get(slab);
    T* minv = a.mass ? (T*)ws.get((size_t)N * sizeof(T)) : nullptr;
    if (!vh || !qa || !qb || (a.mass && !minv))
        return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the large-D workspace");
    const dim3 b1(256), g1((unsigned)((N + 255) / 256));
    if (minv) hipLaunchKernelGGL(k_big_inv<T>, g1, b1, 0, st, (const T*)a.mass, minv, N);
    // vh = p * (1/m)
    hipLaunchKernelGGL(k_big_copy<T>, grid2d(N, D), b1, 0, st, (const T*)a.p, a.ldn, vh, (int64_t)N, D,
                       N, T(1), (const T*)minv);
    const T h = (T)a.h, hh = (T)(0.5 * a.h);
    const T* qcur = (const T*)a.q;
    int64_t ldcur = a.ldn;
    const bool sv = (a.method == PBBI_STORMER_VERLET);
    for (int j = 0; j <= L && (sv || L > 0); ++j) {
        const bool first = (j == 0), last = (j == L), drift = sv || !last;
        T* qnext = drift ? ((j & 1) ? qb : qa) : nullptr;
        if (int rc = gemm<T>(pot, EPI_KDK, qcur, ldcur, qnext, vh, N, minv, nullptr, 0, nullptr, N,
                             (first || (last && !sv)) ? hh : h, h, st))
            return rc;
        if (drift) { qcur = qnext; ldcur = N; }
    }
    if (a.v_out)
        hipLaunchKernelGGL(k_big_copy<T>, grid2d(N, D), b1, 0, st, (const T*)vh, (int64_t)N,
                           (T*)a.v_out, a.ldn, D, N, T(1), (const T*)nullptr);
    if (qcur != (const T*)a.q)
        hipLaunchKernelGGL(k_big_copy<T>, grid2d(N, D), b1, 0, st, qcur, (int64_t)N, (T*)a.q, a.ldn, D,
                           N, T(1), (const T*)nullptr);
    hipLaunchKernelGGL(k_big_copy<T>, grid2d(N, D), b1, 0, st, (const T*)vh, (int64_t)N, (T*)a.p, a.ldn,
                       D, N, T(1), (const T*)a.mass);  // p = v*m
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

template <typename T>
int run_eval(const EvalArgs& a, int mode) {
    const pbbi_potential* pot = a.pot;
    const int D = pot->D;
    const int64_t N = a.N;
    hipStream_t st = a.stream;
    Scratch ws(st);
    const int n_xg = pot->DPAD_big / BM, n_sq = (D + SQ_ROWS - 1) / SQ_ROWS;
    T* xg = (T*)ws.get((size_t)n_xg * N * sizeof(T));
    T* pp = mode ? (T*)ws.get((size_t)n_sq * N * sizeof(T)) : nullptr;
    if (!xg || (mode && !pp))
        return pbbi_fail(PBBI_ERR_HIP, "hipMallocAsync failed for the large-D workspace");
    if (int rc = gemm<T>(pot, EPI_EVAL, (const T*)a.q, a.ldn, nullptr, nullptr, 0, nullptr,
                         (T*)a.grad_out, a.ldn, xg, N, T(0), T(0), st))
        return rc;
    const dim3 b1(256), g1((unsigned)((N + 255) / 256));
    if (mode)
        hipLaunchKernelGGL(k_big_sq_partial<T>, dim3(g1.x, (unsigned)n_sq), b1, 0, st, (const T*)a.p,
                           a.ldn, D, N, SQ_ROWS, pp);
    hipLaunchKernelGGL(k_big_finish_eval<T>, g1, b1, 0, st, (const T*)xg, n_xg, (const T*)pp, n_sq,
                       (const T*)a.mass, (T)pot->cst, N, (T*)a.U_out, (T*)a.w_out, mode);
    PBBI_HIP(hipGetLastError());
    return PBBI_OK;
}

int check(const pbbi_potential* pot) {
    if (!pot->d_big_PT)
        return pbbi_fail(PBBI_ERR_UNSUPPORTED, "large-D dense path was not built for this handle");
    return PBBI_OK;
}

}  // namespace

// P^T, zero padded to DPAD x DPAD (DPAD multiple of 128), and mu padded; dtype of the handle.
int big_build(pbbi_potential* pot, const double* P, const double* mean) {
    const int D = pot->D;
    const int DPAD = ((D + BM - 1) / BM) * BM;
    pot->DPAD_big = DPAD;
    const size_t es = pot->dtype == PBBI_F64 ? 8 : 4;
    std::vector<char> buf((size_t)DPAD * DPAD * es, 0), mub((size_t)DPAD * es, 0);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            const double v = P[(size_t)i * D + k];  // PT[k][i] = P[i][k]
            if (es == 8) ((double*)buf.data())[(size_t)k * DPAD + i] = v;
            else ((float*)buf.data())[(size_t)k * DPAD + i] = (float)v;
        }
    pot->zero_mean = true;
    for (int d = 0; d < D; ++d) {
        const double v = mean ? mean[d] : 0.0;
        if (v != 0.0) pot->zero_mean = false;
        if (es == 8) ((double*)mub.data())[d] = v;
        else ((float*)mub.data())[d] = (float)v;
    }
    PBBI_HIP(hipMalloc(&pot->d_big_PT, buf.size()));
    PBBI_HIP(hipMalloc(&pot->d_big_mu, mub.size()));
    PBBI_HIP(hipMemcpy(pot->d_big_PT, buf.data(), buf.size(), hipMemcpyHostToDevice));
    PBBI_HIP(hipMemcpy(pot->d_big_mu, mub.data(), mub.size(), hipMemcpyHostToDevice));
    return PBBI_OK;
}

// May the iterations of a run on these arguments carry the gradient (run_hmc)?  Plain Leapfrog, L >= 1.
bool big_carry_applies(const IterArgs& a) {
    static const bool off = (getenv("PBBI_NO_CARRY") != nullptr);  // A/B switch
    return !off && a.method == PBBI_LEAPFROG && a.L >= 1 && !pbbi_dyn(a) && a.pot->d_big_PT != nullptr && a.N > 0;
}
// bytes of IterArgs::carry_g (two gradient slabs) and of IterArgs::carry_sel (accept bytes + x.g partial sums)
void big_carry_bytes(const pbbi_potential* pot, int64_t N, size_t* g_bytes, size_t* sel_bytes) {
    const size_t es = pot->dtype == PBBI_F64 ? 8 : 4;
    *g_bytes = 2 * (size_t)pot->D * (size_t)N * es;
    *sel_bytes = ((size_t)N + 255) / 256 * 256 + (size_t)(pot->DPAD_big / BM) * (size_t)N * es;
}

int big_hmc_iter(const IterArgs& a) {
    if (int rc = check(a.pot)) return rc;
    if (a.N == 0) return PBBI_OK;
    return a.pot->dtype == PBBI_F64 ? run_hmc<double>(a) : run_hmc<float>(a);
}

int big_integrate(const IntegrateArgs& a) {
    if (int rc = check(a.pot)) return rc;
    if (a.N == 0) return PBBI_OK;
    return a.pot->dtype == PBBI_F64 ? run_integrate<double>(a) : run_integrate<float>(a);
}

int big_eval(const EvalArgs& a) {
    if (int rc = check(a.pot)) return rc;
    if (a.N == 0) return PBBI_OK;
    return a.pot->dtype == PBBI_F64 ? run_eval<double>(a, 0) : run_eval<float>(a, 0);
}

int big_energy(const EvalArgs& a) {
    if (int rc = check(a.pot)) return rc;
    if (a.N == 0) return PBBI_OK;
    const int mode = a.ratio_finish ? 2 : 1;
    return a.pot->dtype == PBBI_F64 ? run_eval<double>(a, mode) : run_eval<float>(a, mode);
}
