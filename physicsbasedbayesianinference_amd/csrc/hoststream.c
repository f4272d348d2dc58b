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
 *   2. the polar transform of the attempts (4 words each), IN PARALLEL over chunks of the word buffer
 *      (OpenMP): accept flags -> per-chunk counts -> exclusive prefix -> every chunk writes its
 *      accepted pairs at their final positions (f*x2 first, then f*x1: legacy_gauss returns the
 *      second variate first and caches the first).
 * The state that goes back to NumPy (key, pos, has_gauss, cached gaussian) is exactly what its own
 * generator would have left, so draws before and after interleave freely with np.random calls.
 *
 * Host-side RNG plumbing, not part of the GPU hot path and not a fallback for it: when this library
 * is absent the Python layer calls np.random itself (identical results, slower).
 * Mirrors numpy/random/src/legacy/legacy-distributions.c (legacy_gauss, legacy_double) and
 * numpy/random/src/mt19937/mt19937.c (mt19937_gen, tempering) -- algorithms restated, no code copied.
 */
#include <math.h>
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
__attribute__((target_clones("avx2", "default")))
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

/* one polar attempt from 4 words: returns 1 and (first, second) = (f*x2, f*x1) when accepted */
static inline int attempt(const uint32_t* w, double* first, double* second) {
    const double x1 = 2.0 * words_to_double(w[0], w[1]) - 1.0;
    const double x2 = 2.0 * words_to_double(w[2], w[3]) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    if (r2 >= 1.0 || r2 == 0.0) return 0;
    {
        const double f = sqrt(-2.0 * log(r2) / r2);
        *first = f * x2;   /* returned now */
        *second = f * x1;  /* cached: returned by the next call */
        return 1;
    }
}

static inline int attempt_accepts(const uint32_t* w) {
    const double x1 = 2.0 * words_to_double(w[0], w[1]) - 1.0;
    const double x2 = 2.0 * words_to_double(w[2], w[3]) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    return !(r2 >= 1.0 || r2 == 0.0);
}

/* out[0 .. n) = the next n legacy standard normals of the state; the state is advanced.
 * Returns 0, or -1 when memory runs out (state untouched). */
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

static int normal_core(hs_state* st, double* out, int64_t n, const double* scale, int64_t ncol) {
    if (n <= 0) return 0;
    const int had_gauss = st->has_gauss;
    const double old_gauss = st->gauss;
    int64_t done = 0;
#define SCALED(i, v) ((scale) ? (v) * scale[(i) % ncol] : (v))
    if (st->has_gauss) {  /* the cached variate goes first */
        out[0] = SCALED(0, st->gauss);
        ++done;
        st->has_gauss = 0;
        st->gauss = 0.0;
        if (done == n) return 0;
    }
    const int64_t pairs = (n - done + 1) / 2;  /* accepted attempts needed (the last may leave a cached variate) */
    hs_stream s;
    if (stream_init(&s, st)) {
        st->has_gauss = had_gauss;
        st->gauss = old_gauss;
        return -1;
    }
    const int64_t CH = 1 << 15;                /* attempts per chunk */
    int64_t att_done = 0, acc_done = 0;        /* attempts examined, pairs written */
    int64_t* counts = NULL;
    int rc = 0;
    while (acc_done < pairs) {
        /* expected attempts for what is missing, plus a margin; at least one chunk */
        int64_t want = (int64_t)((double)(pairs - acc_done) / 0.7853981633974483 * 1.02) + 1024;
        const int64_t nch = (want + CH - 1) / CH;
        want = nch * CH;
        if (stream_ensure(&s, (size_t)(att_done + want) * 4)) { rc = -1; break; }
        int64_t* nc = (int64_t*)realloc(counts, (size_t)(nch + 1) * sizeof(int64_t));
        if (!nc) { rc = -1; break; }
        counts = nc;
        const uint32_t* w = s.words + (size_t)att_done * 4;
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int64_t c = 0; c < nch; ++c) {
            int64_t k = 0;
            const uint32_t* wc = w + (size_t)c * CH * 4;
            for (int64_t a = 0; a < CH; ++a) k += attempt_accepts(wc + a * 4);
            counts[c] = k;
        }
        /* exclusive prefix; the chunk in which the last needed pair falls ends the batch */
        int64_t run = acc_done, last_chunk = nch - 1;
        int finished = 0;
        for (int64_t c = 0; c < nch; ++c) {
            const int64_t k = counts[c];
            counts[c] = run;
            run += k;
            if (!finished && run >= pairs) { last_chunk = c; finished = 1; }
        }
        counts[nch] = run;
        int64_t end_attempt = -1;  /* index (within the batch) of the attempt that supplies the last pair */
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int64_t c = 0; c <= last_chunk; ++c) {
            int64_t k = counts[c];
            const uint32_t* wc = w + (size_t)c * CH * 4;
            for (int64_t a = 0; a < CH && k < pairs; ++a) {
                double f, g;
                if (attempt(wc + a * 4, &f, &g)) {
                    const int64_t o = done + 2 * k;
                    out[o] = SCALED(o, f);
                    if (o + 1 < n) out[o + 1] = SCALED(o + 1, g);
                    else { st->gauss = g; }  /* only the very last pair of an odd request lands here */
                    ++k;
                    if (k == pairs) {
#pragma omp atomic write
                        end_attempt = c * CH + a;
                    }
                }
            }
        }
        if (finished) {
            att_done += end_attempt + 1;
            acc_done = pairs;
        } else {
            att_done += want;
            acc_done = run;
        }
    }
    if (rc != 0) {  /* out of memory: leave the state as it was found */
        st->has_gauss = had_gauss;
        st->gauss = old_gauss;
    }
    if (rc == 0) {
        const int odd = ((n - done) & 1) != 0;
        const double cached = st->gauss;
        stream_state_after(&s, (size_t)att_done * 4, st);
        st->has_gauss = odd;
        st->gauss = odd ? cached : 0.0;
    }
#undef SCALED
    free(counts);
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
