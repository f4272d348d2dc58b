/*
 * hoststream.c -- libpbbi_host.so: NumPy's LEGACY random stream, faster, bit for bit.
 *
 * The drop-in's default mode replays the reference's RNG consumption on NumPy's global legacy
 * RandomState (src/ensemble.py:72-74,88-91, src/HMC.py:168): standard_normal((D, N)) per iteration,
 * then uniform(size=N).  That generator is one sequential MT19937 + polar Box-Muller loop, ~11 ns
 * per normal on the GPU box's host: 93 ms per iteration of config C2, against 0.35 ms for the kernel.
 * This file produces THE SAME stream from the same state -- same 32-bit outputs, same doubles, same
 * rejection loop, the same libm log / sqrt NumPy's own C code calls -- but splits the work:
 *   1. MT19937 raw words, sequential by nature, block by block (three dependence-free phases per
 *      624-word block, which the compiler vectorises);
 *   2. the polar transform of the attempts (4 words each), IN PARALLEL over chunks of the block buffer
 *      and overlapped with 1.: one thread generates, the others take chunks in order, count each chunk's
 *      accepted attempts, chain the running count from the chunk before (a single-pass scan) and write
 *      the accepted pairs at their final positions (f*x2 first, then f*x1: legacy_gauss returns the
 *      second variate first and caches the first).
 * The state that goes back to NumPy (key, pos, has_gauss, cached gaussian) is exactly what its own
 * generator would have left, so draws before and after interleave freely with np.random calls.
 *
 * Host-side RNG plumbing, not part of the GPU hot path and not a fallback for it: when this library
 * is absent the Python layer calls np.random itself (identical results, slower).
 * Mirrors numpy/random/src/legacy/legacy-distributions.c (legacy_gauss, legacy_double) and
 * numpy/random/src/mt19937/mt19937.c (mt19937_gen, tempering) -- algorithms restated, no code copied.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MT_N 624
#define MT_M 397

/* Threads of every parallel region below.  A num_threads clause, not omp_set_num_threads: that setting
 * belongs to the calling thread, and the draws are made from a producer thread of the Python layer
 * (which would otherwise start one thread per CPU of the host: 256 on the GPU box). */
static int g_threads = 8;

typedef struct {
    uint32_t key[MT_N];
    int pos;            /* next word of key[] to hand out; MT_N = regenerate first */
    int has_gauss;
    double gauss;
} hs_state;

/* next block of 624 untempered state words from the previous one (the MT19937 recurrence) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void mt_next_block(const uint32_t* restrict old, uint32_t* restrict nw) {
    int i;
    for (i = 0; i < MT_N - MT_M; ++i) {                 /* 0 .. 226: reads old only */
        const uint32_t y = (old[i] & 0x80000000u) | (old[i + 1] & 0x7fffffffu);
        nw[i] = old[i + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 2 * (MT_N - MT_M); ++i) {                /* 227 .. 453: reads nw[0 .. 226] */
        const uint32_t y = (old[i] & 0x80000000u) | (old[i + 1] & 0x7fffffffu);
        nw[i] = nw[i - (MT_N - MT_M)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < MT_N - 1; ++i) {                         /* 454 .. 622: reads nw[227 .. 395] */
        const uint32_t y = (old[i] & 0x80000000u) | (old[i + 1] & 0x7fffffffu);
        nw[i] = nw[i - (MT_N - MT_M)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    {
        const uint32_t y = (old[MT_N - 1] & 0x80000000u) | (nw[0] & 0x7fffffffu);
        nw[MT_N - 1] = nw[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

/* ---- MT19937 jump-ahead: several generator threads for one stream -----------------------------------------
 * The recurrence is linear over GF(2): the state n words ahead is p_n(F) applied to the state, where F advances
 * the 624-word window by ONE word and p_n(x) = x^n mod phi(x), phi the characteristic polynomial of F (degree
 * 19937).  phi comes from Berlekamp-Massey on 2 x 19937 output bits, p_n from square-and-multiply, p_n(F) S
 * from Horner's rule (19937 single-word steps, a 624-word XOR for every set coefficient: ~0.3 ms).  One table
 * of p_{g R} (g = 1 .. generators - 1, R blocks per range) is built per range length and kept.  A jumped state
 * has the right bits wherever the recurrence reads them -- the low 31 bits of its first word are not part of the
 * 19937-bit state -- so it serves only as the block BEFORE a range: every word a range hands out is produced by
 * mt_next_block itself. */
#define MT_DEG 19937
#define PW 312                         /* 64-bit words of a polynomial of degree < 19968 */

static inline int pbit(const uint64_t* a, int i) { return (int)((a[i >> 6] >> (i & 63)) & 1u); }

/* dst ^= src << sh   (src: nw words; dst holds at least nw + sh / 64 + 1 words) */
static void pxor_shl(uint64_t* restrict dst, const uint64_t* restrict src, int nw, int sh) {
    const int ws = sh >> 6, bs = sh & 63;
    if (bs == 0) {
        for (int i = 0; i < nw; ++i) dst[i + ws] ^= src[i];
    } else {
        for (int i = 0; i < nw; ++i) {
            dst[i + ws] ^= src[i] << bs;
            dst[i + ws + 1] ^= src[i] >> (64 - bs);
        }
    }
}

/* a (2 * PW + 2 words, degree < 2 * MT_DEG) reduced mod phi in place: degree < MT_DEG afterwards */
static void preduce(uint64_t* a, const uint64_t* phi) {
    for (int k = 2 * MT_DEG; k >= MT_DEG; --k)
        if (pbit(a, k)) pxor_shl(a, phi, PW, k - MT_DEG);
}

static void pmulmod(const uint64_t* a, const uint64_t* b, const uint64_t* phi, uint64_t* out) {
    static uint64_t acc[2 * PW + 2];   /* (called under g_busy only) */
    memset(acc, 0, sizeof(acc));
    for (int i = 0; i < MT_DEG; ++i)
        if (pbit(b, i)) pxor_shl(acc, a, PW, i);
    preduce(acc, phi);
    memcpy(out, acc, PW * sizeof(uint64_t));
}

/* out = x^n mod phi */
static void ppowx(uint64_t n, const uint64_t* phi, uint64_t* out) {
    uint64_t r[PW + 1];
    memset(r, 0, sizeof(r));
    r[0] = 1;
    for (int b = 63; b >= 0; --b) {
        uint64_t t[PW];
        memcpy(t, r, sizeof(t));
        pmulmod(t, t, phi, r);
        r[PW] = 0;
        if ((n >> b) & 1u) {            /* times x */
            for (int i = PW; i > 0; --i) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
            r[0] <<= 1;
            if (pbit(r, MT_DEG)) for (int i = 0; i < PW; ++i) r[i] ^= phi[i];
            r[PW] = 0;
        }
    }
    memcpy(out, r, PW * sizeof(uint64_t));
}

static void mt_next_block(const uint32_t* restrict old, uint32_t* restrict nw);

/* phi(x): Berlekamp-Massey over GF(2) on the lowest bit of 2 * MT_DEG + 64 consecutive state words.  The linear
 * complexity never exceeds MT_DEG, so the connection polynomials fit PW + 2 words throughout. */
static int mt_char_poly(uint64_t* phi) {
    enum { NB = 2 * MT_DEG + 64, NBLK = NB / MT_N + 2, AW = PW + 2 };
    uint32_t* blk = (uint32_t*)malloc((size_t)(NBLK + 1) * MT_N * sizeof(uint32_t));
    uint64_t* C = (uint64_t*)calloc(4 * (size_t)AW, sizeof(uint64_t));
    if (!blk || !C) { free(blk); free(C); return -1; }
    uint64_t *B = C + AW, *H = B + AW, *T = H + AW;   /* H: bit i = s_{n-i} */
    uint32_t x = 19650218u;
    for (int i = 0; i < MT_N; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; blk[i] = x; }
    for (int b = 0; b < NBLK; ++b) mt_next_block(blk + (size_t)b * MT_N, blk + (size_t)(b + 1) * MT_N);
    const uint32_t* seq = blk + MT_N;          /* generated words only */
    C[0] = B[0] = 1;
    int L = 0, m = 1, ok = 1;
    for (int n = 0; n < NB && ok; ++n) {
        for (int i = AW - 1; i > 0; --i) H[i] = (H[i] << 1) | (H[i - 1] >> 63);   /* the history moves on */
        H[0] = (H[0] << 1) | (uint64_t)(seq[n] & 1u);
        uint64_t par = 0;
        const int lw = (L >> 6) + 1;
        for (int i = 0; i < lw; ++i) par ^= C[i] & H[i];
        if (!__builtin_parityll(par)) { ++m; continue; }
        const int nw = AW - 1 - (m >> 6);      /* words of B that x^m B can still place inside AW words */
        if (nw <= 0) { ok = 0; break; }
        if (2 * L <= n) {
            memcpy(T, C, (size_t)AW * sizeof(uint64_t));
            pxor_shl(C, B, nw, m);
            L = n + 1 - L;
            memcpy(B, T, (size_t)AW * sizeof(uint64_t));
            m = 1;
            if (L > MT_DEG) ok = 0;
        } else {
            pxor_shl(C, B, nw, m);
            ++m;
        }
    }
    ok = ok && (L == MT_DEG);
    if (ok) {   /* phi_i = c_{L-i}: the reciprocal of the connection polynomial */
        memset(phi, 0, (size_t)PW * sizeof(uint64_t));
        for (int i = 0; i <= MT_DEG; ++i)
            if (pbit(C, MT_DEG - i)) phi[i >> 6] |= (uint64_t)1 << (i & 63);
    }
    free(blk);
    free(C);
    return ok ? 0 : -2;
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void xor_block(uint32_t* restrict dst, const uint32_t* restrict src) {
    for (int i = 0; i < MT_N; ++i) dst[i] ^= src[i];
}

/* out = the state poly(F) takes `base` to (624 words; see the note on its first word above) */
static void mt_jump(const uint32_t* base, const uint64_t* poly, uint32_t* out) {
    enum { CAP = 4096 };
    uint32_t R[MT_N + CAP];             /* the window is R[h .. h + 623] (on the stack: the generator threads jump at once) */
    int h = 0;
    memset(R, 0, MT_N * sizeof(uint32_t));
    for (int i = MT_DEG - 1; i >= 0; --i) {
        const uint32_t y = (R[h] & 0x80000000u) | (R[h + 1] & 0x7fffffffu);
        R[h + MT_N] = R[h + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        ++h;
        if (h == CAP) { memmove(R, R + h, MT_N * sizeof(uint32_t)); h = 0; }
        if (pbit(poly, i)) xor_block(R + h, base);
    }
    memcpy(out, R + h, MT_N * sizeof(uint32_t));
}

static volatile int g_busy;            /* (defined with the stream buffers below) */
#define HS_GEN_MAX 8
static int g_gen_threads = 2;            /* generator threads of a large draw (1 = the sequential form); measured on the
                                          * GPU box, 16 threads, one C2 draw: 8.0 ms with 1, 7.5 with 2, 7.6-7.9 with 4-8 */
static size_t g_gen_min_blocks = 8192;   /* ... used from this many new blocks per pass on (5 M words) */
static size_t g_gen_round = 1024;        /* range lengths are rounded up to a multiple of this (one table per length) */
static uint64_t g_phi[PW];
static int g_phi_state = 0;              /* 0 not tried, 1 ready, -1 failed (sequential form from then on) */
static uint64_t g_jump_poly[HS_GEN_MAX][PW];
static size_t g_jump_R = 0;              /* blocks per range the table holds */
static int g_jump_n = 0;                 /* polynomials in it (g = 1 .. g_jump_n) */

void pbbi_host_debug_set_gen(int threads, int64_t min_blocks, int64_t round_blocks) {
    g_gen_threads = threads < 1 ? 1 : (threads > HS_GEN_MAX ? HS_GEN_MAX : threads);
    g_gen_min_blocks = min_blocks > 0 ? (size_t)min_blocks : 8192;
    g_gen_round = round_blocks > 0 ? (size_t)round_blocks : 1024;
}

/* p_{g R} for g = 1 .. gens - 1 (under g_busy); 0 when the table is ready */
static int jump_prepare(size_t R, int gens) {
    if (g_phi_state == 0) g_phi_state = (mt_char_poly(g_phi) == 0) ? 1 : -1;
    if (g_phi_state != 1) return -1;
    if (g_jump_R == R && g_jump_n >= gens - 1) return 0;
    ppowx((uint64_t)R * MT_N, g_phi, g_jump_poly[1]);
    for (int g = 2; g < gens; ++g) pmulmod(g_jump_poly[g - 1], g_jump_poly[1], g_phi, g_jump_poly[g]);
    g_jump_R = R;
    g_jump_n = gens - 1;
    return 0;
}

/* test hook: does a jump of `blocks` blocks land where the recurrence does?  0 = yes (every bit the recurrence
 * reads); -1 no table; 1 mismatch.  (tests/test_host_logic.py) */
int pbbi_host_debug_jump_check(int64_t blocks, uint32_t seed) {
    if (blocks < 1 || blocks > 4096) return -3;
    if (__sync_lock_test_and_set(&g_busy, 1)) return -4;
    int rc = 0;
    uint32_t* blk = (uint32_t*)malloc((size_t)(blocks + 1) * MT_N * sizeof(uint32_t));
    uint32_t jumped[MT_N];
    static uint64_t poly[PW];
    if (!blk) rc = -1;
    if (rc == 0 && g_phi_state == 0) g_phi_state = (mt_char_poly(g_phi) == 0) ? 1 : -1;
    if (rc == 0 && g_phi_state != 1) rc = -1;
    if (rc == 0) {
        uint32_t x = seed;
        for (int i = 0; i < MT_N; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; blk[i] = x; }
        for (int64_t b = 0; b < blocks; ++b) mt_next_block(blk + b * MT_N, blk + (b + 1) * MT_N);
        ppowx((uint64_t)blocks * MT_N, g_phi, poly);
        mt_jump(blk, poly, jumped);
        const uint32_t* ref = blk + blocks * MT_N;
        if ((jumped[0] ^ ref[0]) & 0x80000000u) rc = 1;
        for (int i = 1; i < MT_N && rc == 0; ++i) if (jumped[i] != ref[i]) rc = 1;
    }
    free(blk);
    __sync_lock_release(&g_busy);
    return rc;
}

static inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* A growable buffer of the stream's tempered words from the call's starting position on, together
 * with the untempered blocks they came from (needed to hand the state back at any word index). */
typedef struct {
    uint32_t* blocks;   /* [n_blocks][MT_N] untempered; block 0 = the state the call started with */
    size_t n_blocks, cap_blocks;
    uint32_t* words;    /* tempered words from (block 0, pos0) on */
    size_t n_words, cap_words;
    int pos0;
} hs_stream;

/* The two buffers are kept between calls (fresh 86-MB allocations per C2 iteration cost more in page
 * faults than the transform itself); one call at a time, like the global stream they feed. */
static uint32_t* g_blocks = NULL;
static size_t g_cap_blocks = 0;
static uint32_t* g_words = NULL;
static size_t g_cap_words = 0;
static char* g_scratch = NULL;
static size_t g_cap_scratch = 0;
static volatile int g_busy = 0;

static int stream_init(hs_stream* s, const hs_state* st) {
    memset(s, 0, sizeof(*s));
    if (__sync_lock_test_and_set(&g_busy, 1)) return -1;  /* concurrent use: the caller lets NumPy draw */
    if (!g_blocks) {
        g_cap_blocks = 64;
        g_blocks = (uint32_t*)malloc(g_cap_blocks * MT_N * sizeof(uint32_t));
        if (!g_blocks) { g_cap_blocks = 0; __sync_lock_release(&g_busy); return -1; }
    }
    s->blocks = g_blocks; s->cap_blocks = g_cap_blocks;
    s->words = g_words; s->cap_words = g_cap_words;
    memcpy(s->blocks, st->key, sizeof(st->key));
    s->n_blocks = 1;
    s->pos0 = st->pos;
    return 0;
}

static void stream_free(hs_stream* s) {  /* hands the (possibly grown) buffers back */
    g_blocks = s->blocks; g_cap_blocks = s->cap_blocks;
    g_words = s->words; g_cap_words = s->cap_words;
    __sync_lock_release(&g_busy);
}

/* make at least `need` tempered words available */
static int stream_ensure(hs_stream* s, size_t need) {
    if (need <= s->n_words) return 0;
    /* words available from the blocks we hold: n_blocks*624 - pos0 */
    size_t have_raw = s->n_blocks * MT_N - (size_t)s->pos0;
    if (need > have_raw) {
        const size_t more = (need - have_raw + MT_N - 1) / MT_N;
        if (s->n_blocks + more > s->cap_blocks) {
            size_t cap = s->cap_blocks;
            while (cap < s->n_blocks + more) cap *= 2;
            uint32_t* nb = (uint32_t*)realloc(s->blocks, cap * MT_N * sizeof(uint32_t));
            if (!nb) return -1;
            s->blocks = nb;
            s->cap_blocks = cap;
        }
        for (size_t b = 0; b < more; ++b) {  /* sequential by nature */
            mt_next_block(s->blocks + (s->n_blocks - 1) * MT_N, s->blocks + s->n_blocks * MT_N);
            ++s->n_blocks;
        }
        have_raw = s->n_blocks * MT_N - (size_t)s->pos0;
    }
    if (have_raw > s->cap_words) {
        uint32_t* nw = (uint32_t*)realloc(s->words, have_raw * sizeof(uint32_t));
        if (!nw) return -1;
        s->words = nw;
        s->cap_words = have_raw;
    }
    {   /* temper the new words (parallel: independent) */
        const size_t lo = s->n_words, hi = have_raw;
        const uint32_t* src = s->blocks + s->pos0;
        uint32_t* dst = s->words;
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int64_t i = (int64_t)lo; i < (int64_t)hi; ++i) dst[i] = temper(src[i]);
        s->n_words = hi;
    }
    return 0;
}

/* state after `consumed` words of the stream */
static void stream_state_after(const hs_stream* s, size_t consumed, hs_state* st) {
    const size_t g = (size_t)s->pos0 + consumed;
    size_t b = g / MT_N;
    int pos = (int)(g % MT_N);
    if (pos == 0 && b > 0) { b -= 1; pos = MT_N; }  /* the form NumPy itself leaves behind */
    memcpy(st->key, s->blocks + b * MT_N, sizeof(st->key));
    st->pos = pos;
}

static inline double words_to_double(uint32_t a, uint32_t b) {  /* mt19937 random_double */
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* one polar attempt from 4 UNTEMPERED words: returns 1 and (first, second) = (f*x2, f*x1) when accepted */
static inline int attempt(const uint32_t* w, double* first, double* second) {
    const double x1 = 2.0 * words_to_double(temper(w[0]), temper(w[1])) - 1.0;
    const double x2 = 2.0 * words_to_double(temper(w[2]), temper(w[3])) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    if (r2 >= 1.0 || r2 == 0.0) return 0;
    {
        const double f = sqrt(-2.0 * log(r2) / r2);
        *first = f * x2;   /* returned now */
        *second = f * x1;  /* cached: returned by the next call */
        return 1;
    }
}

/* The same transform over a block of attempts, in phases, so that everything except NumPy's own log() call
 * vectorises: (a) temper + uniform -> x1, x2, r2 for every attempt, (b) compaction of the accepted ones,
 * (c) log(r2) -- the libm call NumPy makes, scalar, bit-exactness pins it --, (d) f = sqrt(-2 log(r2) / r2) and the
 * two products.  Division and square root are correctly rounded in every vector width and -ffp-contract=off keeps
 * x1*x1 + x2*x2 two roundings, so the results are the bits attempt() produces (tests/test_host_logic.py:
 * test_fast_host_stream_*).  target_clones: one build serves AVX-512, AVX2 and baseline hosts. */
#define HS_SUB 1024
#if defined(__x86_64__) && defined(__GNUC__) && !defined(HS_NO_CLONES)
#define HS_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define HS_CLONES
#endif
HS_CLONES static void polar_candidates(const uint32_t* restrict w, int64_t na, double* restrict x1,
                                       double* restrict x2, double* restrict r2) {
    for (int64_t a = 0; a < na; ++a) {
        const double d1 = words_to_double(temper(w[4 * a]), temper(w[4 * a + 1]));
        const double d2 = words_to_double(temper(w[4 * a + 2]), temper(w[4 * a + 3]));
        const double a1 = 2.0 * d1 - 1.0, a2 = 2.0 * d2 - 1.0;
        x1[a] = a1;
        x2[a] = a2;
        r2[a] = a1 * a1 + a2 * a2;
    }
}
HS_CLONES static void polar_finish(int64_t kc, const double* restrict x1, const double* restrict x2,
                                   const double* restrict r2, const double* restrict lg, double* restrict out) {
    double f[HS_SUB];
    for (int64_t k = 0; k < kc; ++k) f[k] = sqrt(-2.0 * lg[k] / r2[k]);
    for (int64_t k = 0; k < kc; ++k) {
        out[2 * k] = f[k] * x2[k];      /* returned first */
        out[2 * k + 1] = f[k] * x1[k];  /* the cached one */
    }
}
/* attempts a0 .. a0 + na - 1 of a chunk (na <= HS_SUB): accepted pairs appended to out / att; returns their number */
static int64_t polar_block(const uint32_t* w, int64_t na, int64_t a0, double* out, int32_t* att) {
    double x1[HS_SUB], x2[HS_SUB], r2[HS_SUB], lg[HS_SUB];
    int64_t kc = 0;
    polar_candidates(w, na, x1, x2, r2);
    for (int64_t a = 0; a < na; ++a) {
        const double r = r2[a];
        if (r >= 1.0 || r == 0.0) continue;
        x1[kc] = x1[a];
        x2[kc] = x2[a];
        r2[kc] = r;
        att[kc] = (int32_t)(a0 + a);
        ++kc;
    }
    for (int64_t k = 0; k < kc; ++k) lg[k] = log(r2[k]);
    polar_finish(kc, x1, x2, r2, lg, out);
    return kc;
}

/* waiting for another thread's result: a few pause instructions, then give the CPU away (more threads
 * than free cores must not starve the thread everybody waits for) */
static inline void spin_pause(unsigned* spins) {
    if (++*spins < 64) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    } else {
        sched_yield();
    }
}

/* out[0 .. n) = the next n legacy standard normals of the state; the state is advanced.
 * Returns 0, or -1 when memory runs out / the stream is in use (state untouched). */
static int normal_core(hs_state* st, double* out, int64_t n, const double* scale, int64_t ncol);

int pbbi_host_standard_normal(hs_state* st, double* out, int64_t n) {
    return normal_core(st, out, n, NULL, 1);
}

/* out[i] = z_i * scale[i % ncol]: Ensemble.setMomentum's `standard_normal((D, N)) * pStd` (src/ensemble.py:
 * 88-91) in one pass, written where the caller wants it (a pinned upload buffer); one IEEE multiply per
 * element, as NumPy's broadcast does */
int pbbi_host_scaled_normal(hs_state* st, double* out, int64_t n, const double* scale, int64_t ncol) {
    if (ncol < 1) return -2;
    return normal_core(st, out, n, scale, ncol);
}

/* One pass over `nch` chunks of CH attempts starting at attempt `att0` of the stream (word pos0 + 4*att0
 * of the block buffer), pipelined: thread 0 runs the MT19937 recurrence (the only sequential part) and
 * publishes how many blocks are complete; the other threads take chunks in order, count each chunk's
 * accepted attempts, chain the running count from the chunk before (published per chunk, as in a
 * single-pass scan) and write their pairs at the final positions.  Thread 0 joins them when the blocks are
 * done.  No barrier inside; every wait is on something an earlier ticket or thread 0 produces. */
typedef struct {
    hs_stream* s;
    double* out;
    const double* scale;
    int64_t ncol, n, done, pairs, acc0, att0, nch, CH;
    size_t blocks_target;          /* n_blocks to reach */
    volatile int64_t* pref;        /* inclusive running count after chunk c; -1 = not yet */
    char* scratch;                 /* per thread: the pairs of the chunk in flight + their attempt indices */
    double last_second;            /* the variate an odd request leaves cached */
    /* the three words the threads talk through, one cache line each: the generator's progress is polled by
     * every waiting worker, and a store to a line 15 cores spin on costs a coherence round per store */
    /* generator threads of this pass: range g = blocks [range_start[g], range_start[g + 1]); gen[g].ready = the
     * block index range g has reached (its own cache line: every waiting worker polls it) */
    int gens;
    size_t range_start[HS_GEN_MAX + 1];
    char pad0[128];
    struct { volatile size_t ready; char pad[128 - sizeof(size_t)]; } gen[HS_GEN_MAX];
    volatile int64_t next_chunk;
    char pad2[128 - sizeof(int64_t)];
    volatile int64_t end_attempt;  /* attempt (within the pass) that supplies the last pair; -1 = none */
    char pad3[128 - sizeof(int64_t)];
} hs_pass;

#define HS_SCRATCH_PER_THREAD(CH) ((size_t)(CH) * (2 * sizeof(double) + sizeof(int32_t)))
#define HS_PUBLISH_EVERY 32   /* blocks per progress store (a chunk is 210 blocks) */

#ifdef HS_PROFILE   /* tools/ubench/hoststream_phases.c: where a pass spends its time, per thread */
#include <stdio.h>
#include <time.h>
static double hs_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double hs_prof[256][6];  /* generate, wait blocks, count, wait prefix, write, chunks */
#define HS_T(var) const double var = hs_now()
#define HS_ADD(slot, a, b) hs_prof[omp_get_thread_num() & 255][slot] += (b) - (a)
#else
#define HS_T(var)
#define HS_ADD(slot, a, b)
#endif

/* generator thread g of a pass: range g of the blocks; g > 0 starts from a jump (see mt_jump) */
static void pass_generate(hs_pass* ps, int g) {
    hs_stream* s = ps->s;
    HS_T(t0);
    const size_t b0 = ps->range_start[g], b1 = ps->range_start[g + 1];
    uint32_t jumped[MT_N];
    const uint32_t* prev = s->blocks + (b0 - 1) * MT_N;
    if (g > 0) {
        mt_jump(s->blocks + (ps->range_start[0] - 1) * MT_N, g_jump_poly[g], jumped);
        prev = jumped;
    }
    size_t b = b0;
    while (b < b1) {
        if (__atomic_load_n(&ps->end_attempt, __ATOMIC_RELAXED) >= 0) break;  /* the request is complete */
        for (int k = 0; k < HS_PUBLISH_EVERY && b < b1; ++k, ++b) {
            mt_next_block(prev, s->blocks + b * MT_N);
            prev = s->blocks + b * MT_N;
        }
        __atomic_store_n(&ps->gen[g].ready, b, __ATOMIC_RELEASE);
    }
    /* whoever still waits for blocks past an early stop is released by end_attempt (checked in the wait) */
    HS_T(t1);
    HS_ADD(0, t0, t1);
}

/* are the blocks [first, need) there?  (blocks below range_start[0] come from earlier passes) */
static inline int blocks_there(const hs_pass* ps, size_t first, size_t need) {
    for (int g = 0; g < ps->gens; ++g) {
        const size_t lo = ps->range_start[g], hi = ps->range_start[g + 1];
        if (hi <= first || lo >= need) continue;
        if (__atomic_load_n(&ps->gen[g].ready, __ATOMIC_ACQUIRE) < (need < hi ? need : hi)) return 0;
    }
    return 1;
}

static void pass_work(hs_pass* ps, int tid) {
    const hs_stream* s = ps->s;
    const int64_t CH = ps->CH, pairs = ps->pairs, n = ps->n, done = ps->done, ncol = ps->ncol;
    const double* scale = ps->scale;
    double* out = ps->out;
    double* tmp = (double*)(ps->scratch + (size_t)tid * HS_SCRATCH_PER_THREAD(CH));
    int32_t* att_of = (int32_t*)(tmp + CH * 2);   /* attempt index of the chunk's k-th pair */
    for (;;) {
        const int64_t c = __atomic_fetch_add(&ps->next_chunk, 1, __ATOMIC_RELAXED);
        if (c >= ps->nch) return;
        if (__atomic_load_n(&ps->end_attempt, __ATOMIC_ACQUIRE) >= 0) {  /* past the end of the request */
            __atomic_store_n(&ps->pref[c], pairs, __ATOMIC_RELEASE);
            continue;
        }
        const size_t w0 = (size_t)s->pos0 + (size_t)(ps->att0 + c * CH) * 4;
        const size_t need_blocks = (w0 + (size_t)CH * 4 + MT_N - 1) / MT_N;
        int skip = 0;
        unsigned spins = 0;
        HS_T(t0);
        while (!blocks_there(ps, w0 / MT_N, need_blocks)) {
            if (__atomic_load_n(&ps->end_attempt, __ATOMIC_ACQUIRE) >= 0) { skip = 1; break; }
            spin_pause(&spins);
        }
        if (skip) {
            __atomic_store_n(&ps->pref[c], pairs, __ATOMIC_RELEASE);
            continue;
        }
        const uint32_t* wc = s->blocks + w0;
        int64_t kc = 0;
        HS_T(t1);
        /* transform once, into the thread's own (L2-resident) buffer, HS_SUB attempts at a time */
        for (int64_t a = 0; a < CH; a += HS_SUB) {
            const int64_t na = CH - a < HS_SUB ? CH - a : HS_SUB;
            kc += polar_block(wc + a * 4, na, a, tmp + 2 * kc, att_of + kc);
        }
        HS_T(t2);
        int64_t k = ps->acc0;
        if (c > 0) {
            spins = 0;
            while ((k = __atomic_load_n(&ps->pref[c - 1], __ATOMIC_ACQUIRE)) < 0) spin_pause(&spins);
        }
        __atomic_store_n(&ps->pref[c], k + kc, __ATOMIC_RELEASE);
        HS_T(t3);
        HS_ADD(1, t0, t1); HS_ADD(2, t1, t2); HS_ADD(3, t2, t3);
        if (k >= pairs) continue;
        {   /* the chunk's pairs go to out[done + 2k ...]; the request may end inside the chunk */
            const int64_t take = (pairs - k < kc) ? pairs - k : kc;
            const int64_t o0 = done + 2 * k;
            int64_t cnt = 2 * take;
            if (o0 + cnt > n) {  /* only the very last pair of an odd request: its second variate is cached */
                cnt = n - o0;
                ps->last_second = tmp[2 * take - 1];
            }
            if (scale) {
                int64_t col = o0 % ncol;
                for (int64_t i = 0; i < cnt; ++i) {
                    out[o0 + i] = tmp[i] * scale[col];
                    if (++col == ncol) col = 0;
                }
            } else {
                memcpy(out + o0, tmp, (size_t)cnt * sizeof(double));
            }
            if (k + take == pairs) __atomic_store_n(&ps->end_attempt, c * CH + att_of[take - 1], __ATOMIC_RELEASE);
        }
        HS_T(t4);
        HS_ADD(4, t3, t4);
#ifdef HS_PROFILE
        hs_prof[omp_get_thread_num() & 255][5] += 1;
#endif
    }
}

/* The threads of a pass work on each other's cache lines (the generator's blocks, the running counts):
 * on a two-socket host they are kept on the NUMA node the caller runs on for the length of the call
 * (measured on the GPU box's 2 x EPYC 9575F, 16 threads: 4.5 ms per C2 draw on one node, 7-12 ms
 * scattered over both).  Every thread's own affinity mask is put back when the pass ends.
 * PBBI_HOST_NO_PIN=1 leaves placement to the scheduler. */
#if defined(__linux__)
#define HS_MAX_NODES 16
static int g_nodes = -1;               /* -1 = /sys not read yet */
static cpu_set_t g_node_cpus[HS_MAX_NODES];
static cpu_set_t g_primary;            /* the first logical CPU of every core */

static void read_nodes(void) {
    g_nodes = 0;
    const char* off = getenv("PBBI_HOST_NO_PIN");
    if (off && off[0] == '1') return;
    for (int nd = 0; nd < HS_MAX_NODES; ++nd) {
        char path[96], buf[4096];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", nd);
        FILE* f = fopen(path, "r");
        if (!f) break;
        const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        buf[got] = 0;
        CPU_ZERO(&g_node_cpus[nd]);
        const char* p = buf;
        while (*p) {   /* "0-63,128-191" */
            char* e;
            long a = strtol(p, &e, 10), b;
            if (e == p) break;
            b = a;
            if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
            for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &g_node_cpus[nd]);
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',') break;
        }
        g_nodes = nd + 1;
    }
    if (g_nodes < 2) g_nodes = 0;      /* one node: nothing to choose */
    /* one logical CPU per core: the first of every SMT sibling set (two threads of a pass on one core halve both) */
    CPU_ZERO(&g_primary);
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        char path[128], buf[64];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        FILE* f = fopen(path, "r");
        if (!f) { if (c > 0) break; else continue; }
        const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        buf[got] = 0;
        if (strtol(buf, NULL, 10) == c) CPU_SET(c, &g_primary);   /* the list starts with its lowest member */
    }
}

/* the CPUs of the caller's node that the caller is allowed on; 0 when there is no reason to pin */
static int pass_cpus(cpu_set_t* target, int threads) {
    if (g_nodes < 0) read_nodes();
    if (g_nodes == 0) return 0;
    const int cpu = sched_getcpu();
    cpu_set_t allowed;
    if (cpu < 0 || sched_getaffinity(0, sizeof(allowed), &allowed)) return 0;
    for (int nd = 0; nd < g_nodes; ++nd) {
        if (!CPU_ISSET(cpu, &g_node_cpus[nd])) continue;
        CPU_AND(target, &g_node_cpus[nd], &allowed);
        {
            cpu_set_t cores;
            CPU_AND(&cores, target, &g_primary);
            if (CPU_COUNT(&cores) >= threads) *target = cores;   /* enough whole cores: leave their siblings idle */
        }
        return CPU_COUNT(target) >= threads;   /* a node with fewer usable CPUs than threads: do not squeeze */
    }
    return 0;
}
#define HS_PIN_BEGIN(target, pinned) cpu_set_t hs_old; const int hs_pin = (pinned) && \
    !sched_getaffinity(0, sizeof(hs_old), &hs_old) && !sched_setaffinity(0, sizeof(cpu_set_t), (target))
#define HS_PIN_END() do { if (hs_pin) sched_setaffinity(0, sizeof(hs_old), &hs_old); } while (0)
#else
typedef int cpu_set_t;
static int pass_cpus(cpu_set_t* target, int threads) { (void)target; (void)threads; return 0; }
#define HS_PIN_BEGIN(target, pinned)
#define HS_PIN_END()
#endif

/* Test knob (tests/test_host_logic.py): how many attempts a pass asks for -- factor x expectation + extra, in
 * chunks of 2^chunk_log2.  The defaults (2 % + 1024 attempts, chunks of 32 768) make a second pass a > 100 sigma
 * event; a factor < 1 and small chunks force the shortfall passes (acc0 / att0 continuation, buffers regrown)
 * that no ordinary call reaches.  Results do not depend on the knob. */
static double g_pass_factor = 1.02;
static int64_t g_pass_extra = 1024;
static int g_pass_chunk_log2 = 15;
static int g_last_passes = 0;
void pbbi_host_debug_set_pass(double factor, int64_t extra, int chunk_log2) {
    g_pass_factor = factor > 0.0 ? factor : 1.02;
    g_pass_extra = extra >= 0 ? extra : 1024;
    g_pass_chunk_log2 = (chunk_log2 >= 4 && chunk_log2 <= 20) ? chunk_log2 : 15;
}
int pbbi_host_debug_last_passes(void) { return g_last_passes; }

static int normal_core(hs_state* st, double* out, int64_t n, const double* scale, int64_t ncol) {
    if (n <= 0) return 0;
    const int had_gauss = st->has_gauss;
    const double old_gauss = st->gauss;
    int64_t done = 0;
    hs_stream s;
    if (stream_init(&s, st)) return -1;
    if (st->has_gauss) {  /* the cached variate goes first */
        out[0] = scale ? st->gauss * scale[0] : st->gauss;
        ++done;
        st->has_gauss = 0;
        st->gauss = 0.0;
        if (done == n) { stream_free(&s); return 0; }
    }
    const int64_t pairs = (n - done + 1) / 2;  /* accepted attempts needed (the last may leave a cached variate) */
    const int64_t CH = (int64_t)1 << g_pass_chunk_log2;   /* attempts per chunk (32 768: 512 KB of words, L2-resident between the two passes) */
    g_last_passes = 0;
    int64_t att_done = 0, acc_done = 0;        /* attempts examined, pairs written */
    int64_t* pref = NULL;
    double last_second = 0.0;
    int rc = 0;
    const int threads = g_threads;
    if (g_cap_scratch < (size_t)threads * HS_SCRATCH_PER_THREAD(CH)) {  /* (under g_busy, like the other buffers) */
        char* ns = (char*)realloc(g_scratch, (size_t)threads * HS_SCRATCH_PER_THREAD(CH));
        if (!ns) { st->has_gauss = had_gauss; st->gauss = old_gauss; stream_free(&s); return -1; }
        g_scratch = ns;
        g_cap_scratch = (size_t)threads * HS_SCRATCH_PER_THREAD(CH);
    }
    cpu_set_t node_cpus;
    const int pin = threads > 1 && pass_cpus(&node_cpus, threads);
    (void)pin;
    while (acc_done < pairs) {
        /* expected attempts for what is missing, plus a margin; at least one chunk */
        int64_t want = (int64_t)((double)(pairs - acc_done) / 0.7853981633974483 * g_pass_factor) + g_pass_extra;
        if (want < 1) want = 1;
        ++g_last_passes;
        const int64_t nch = (want + CH - 1) / CH;
        want = nch * CH;
        hs_pass ps;
        memset(&ps, 0, sizeof(ps));
        ps.blocks_target = ((size_t)s.pos0 + (size_t)(att_done + want) * 4 + MT_N - 1) / MT_N;
        if (ps.blocks_target > s.cap_blocks) {
            size_t cap = s.cap_blocks;
            while (cap < ps.blocks_target) cap *= 2;
            uint32_t* nb = (uint32_t*)realloc(s.blocks, cap * MT_N * sizeof(uint32_t));
            if (!nb) { rc = -1; break; }
            s.blocks = nb;
            s.cap_blocks = cap;
        }
        int64_t* np_ = (int64_t*)realloc(pref, (size_t)nch * sizeof(int64_t));
        if (!np_) { rc = -1; break; }
        pref = np_;
        for (int64_t c = 0; c < nch; ++c) pref[c] = -1;
        ps.s = &s; ps.out = out; ps.scale = scale; ps.ncol = ncol; ps.n = n; ps.done = done;
        ps.pairs = pairs; ps.acc0 = acc_done; ps.att0 = att_done; ps.nch = nch; ps.CH = CH;
        ps.scratch = g_scratch;
        ps.pref = pref; ps.next_chunk = 0; ps.end_attempt = -1;
        {   /* the new blocks [n_blocks, blocks_target) in ranges, one generator thread each (a large draw only) */
            const size_t n0 = s.n_blocks, T = ps.blocks_target > n0 ? ps.blocks_target - n0 : 0;
            size_t R = T;
            int gens = 1;
            if (threads >= 4 && g_gen_threads > 1 && T >= g_gen_min_blocks) {
                gens = g_gen_threads < threads / 2 ? g_gen_threads : threads / 2;
                R = ((T + (size_t)gens - 1) / (size_t)gens + g_gen_round - 1) / g_gen_round * g_gen_round;
                gens = (int)((T + R - 1) / R);
                if (gens > 1 && jump_prepare(R, gens) != 0) gens = 1;
            }
            ps.gens = gens;
            for (int g = 0; g < gens; ++g) {
                ps.range_start[g] = n0 + (size_t)g * R;
                ps.gen[g].ready = ps.range_start[g];
            }
            ps.range_start[gens] = ps.blocks_target > n0 ? ps.blocks_target : n0;
        }
#pragma omp parallel num_threads(threads)
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num();
#else
            const int tid = 0;
#endif
            HS_PIN_BEGIN(&node_cpus, pin);
            if (tid < ps.gens) pass_generate(&ps, tid);
            pass_work(&ps, tid);
            HS_PIN_END();
        }
        for (int g = 0; g < ps.gens; ++g) {   /* the contiguous prefix of what the generators produced */
            s.n_blocks = ps.gen[g].ready;
            if (ps.gen[g].ready < ps.range_start[g + 1]) break;
        }
        if (ps.end_attempt >= 0) {
            att_done += ps.end_attempt + 1;
            acc_done = pairs;
            last_second = ps.last_second;
        } else {
            att_done += want;
            acc_done = pref[nch - 1];
        }
    }
    if (rc != 0) {  /* out of memory: leave the state as it was found */
        st->has_gauss = had_gauss;
        st->gauss = old_gauss;
    } else {
        const int odd = ((n - done) & 1) != 0;
        /* the block the end position lies in may not exist yet when the pass stopped the generator early:
         * cannot happen -- the finishing chunk waited for all of its blocks -- but the position just past
         * the last block needs none (pos = 624 of the block before) */
        stream_state_after(&s, (size_t)att_done * 4, st);
        st->has_gauss = odd;
        st->gauss = odd ? last_second : 0.0;
    }
    free(pref);
    stream_free(&s);
    return rc;
}

/* out[0 .. n) = the next n legacy random_sample() doubles in [0, 1) (what uniform(size=n) returns for
 * low = 0, high = 1: 0.0 + 1.0 * d is exact) */
int pbbi_host_random_sample(hs_state* st, double* out, int64_t n) {
    if (n <= 0) return 0;
    hs_stream s;
    if (stream_init(&s, st)) return -1;
    if (stream_ensure(&s, (size_t)n * 2)) { stream_free(&s); return -1; }
    const uint32_t* w = s.words;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t i = 0; i < n; ++i) out[i] = words_to_double(w[2 * i], w[2 * i + 1]);
    {
        const int hg = st->has_gauss;
        const double g = st->gauss;
        stream_state_after(&s, (size_t)n * 2, st);
        st->has_gauss = hg;  /* uniform draws leave the cached gaussian alone */
        st->gauss = g;
    }
    stream_free(&s);
    return 0;
}

int pbbi_host_threads(void) { return g_threads; }

void pbbi_host_set_threads(int n) {
    if (n > 0) g_threads = n;
}
