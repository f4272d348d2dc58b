/* asan_driver.c -- the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (test infrastructure).
 * GPU sanitizers are not available on the pool; the oracle is plain C, so its own memory discipline is
 * checked here: every entry point is driven over exactly-sized heap buffers (ragged N, ldn > N, every
 * potential kind, both integrators, per-chain lengths with 0 steps, both Philox draws, a whole run).  The
 * driver is built together with pbbi_oracle.c by `make asan_check`; it prints "asan ok" and exits 0 when the
 * sanitizers stayed silent (they abort the process otherwise).  tests/test_oracle_golden.py runs it. */
#include <stdio.h>

#include "pbbi_oracle.c"

static double* dbuf(size_t n, double fill) {
    double* p = (double*)malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) p[i] = fill + 0.01 * (double)(i % 97);
    return p;
}

int main(void) {
    int bad = 0;
    for (int kind = 0; kind < 4; ++kind) {
        const int D = kind == POT_GAUSS_DENSE ? 7 : (kind == POT_ROSENBROCK ? 5 : 3);
        const int64_t N = 13, ldn = 17;
        double* mean = dbuf((size_t)D, 0.1);
        double* prec = dbuf(kind == POT_GAUSS_DENSE ? (size_t)D * D : (size_t)D, 0.5);
        if (kind == POT_GAUSS_DENSE)
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) prec[i * D + j] = (i == j) ? 1.0 + 0.1 * i : 0.01;
        oracle_pot P = {kind, D, kind == POT_HARMONIC ? NULL : mean, prec, 0.25, 1.0, 100.0, 20.0, NULL, NULL};
        double *q = dbuf((size_t)D * ldn, 0.3), *p = dbuf((size_t)D * ldn, -0.2), *u = dbuf((size_t)N, 0.2);
        double *mass = dbuf((size_t)N, 1.0), *U = dbuf((size_t)N, 0), *g = dbuf((size_t)D * ldn, 0);
        double *v = dbuf((size_t)D * ldn, 0), *ratio = dbuf((size_t)N, 0), *w = dbuf((size_t)N, 0);
        unsigned char* rej = (unsigned char*)malloc((size_t)N);
        int32_t *steps_in = (int32_t*)malloc(sizeof(int32_t) * (size_t)N), *steps_out = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
        for (int64_t n = 0; n < N; ++n) { u[n] = 0.05 * (double)n; steps_in[n] = (int32_t)(n % 5); }
        bad |= oracle_potential(&P, q, N, ldn, U, g);
        for (int method = 0; method < 2; ++method) {
            bad |= oracle_integrate(&P, method, q, p, method ? mass : NULL, N, ldn, 0.01, 4, v);
            bad |= oracle_hmc_iter(&P, method, q, p, u, mass, N, ldn, 0.01, 3, COMPAT_P_FROM_OLDQ, ratio, rej);
            bad |= oracle_hmc_iter_beta(&P, method, q, p, u, NULL, N, ldn, 0.01, 0, 0, 0.5, NULL, NULL);
        }
        bad |= oracle_hmc_iter_dyn(&P, METHOD_LEAPFROG, q, p, u, mass, N, ldn, 0.01, 4, COMPAT_P_FROM_OLDQ, 1.0,
                                   steps_in, 1, steps_out, ratio, rej);
        bad |= oracle_weights(&P, q, p, mass, N, ldn, U, w);
        bad |= oracle_weights_ratio(&P, q, p, g, v, NULL, N, ldn, ratio);
        /* a whole run with each draw; slabs are dense (S, D, N) */
        const int S = 3;
        double *samples = dbuf((size_t)S * D * N, 0), *momenta = dbuf((size_t)S * D * N, 0), *rr = dbuf((size_t)S * N, 0);
        unsigned char* rj2 = (unsigned char*)malloc((size_t)S * N);
        double* qs = dbuf((size_t)D * N, 0.2);
        for (int f64 = 0; f64 < 2; ++f64) {
            bad |= oracle_philox_normal(9, STREAM_POSITION | (f64 ? 0x100 : 0), 0, (uint64_t)1 << 33, D, N, N, 1.0, NULL, qs);
            bad |= oracle_hmc_run_philox(&P, METHOD_LEAPFROG, qs, mass, N, N, 0.01, 3, S,
                                         COMPAT_P_FROM_OLDQ | (f64 ? DRAW_F64 : 0) | BETA_ACCEPT, 9, 2, 5, 1.5, samples,
                                         f64 ? momenta : NULL, rj2, rr);
        }
        bad |= oracle_philox_uniform(9, 1, 2, N, u);
        bad |= oracle_philox_steps(9, 1, 2, N, 7, steps_out);
        bad |= oracle_philox_steps(9, 1, 2, N, 0, steps_out);
        free(mean); free(prec); free(q); free(p); free(u); free(mass); free(U); free(g); free(v); free(ratio);
        free(w); free(rej); free(steps_in); free(steps_out); free(samples); free(momenta); free(rr); free(rj2); free(qs);
    }
    /* empty ensembles */
    {
        double one = 1.0;
        oracle_pot P = {POT_HARMONIC, 1, NULL, &one, 0.0, 1.0, 100.0, 20.0, NULL, NULL};
        bad |= oracle_potential(&P, &one, 0, 0, NULL, NULL);
        bad |= oracle_hmc_iter(&P, 0, &one, &one, &one, NULL, 0, 0, 0.1, 2, 1, NULL, NULL);
    }
    if (bad) { printf("asan driver: an entry point returned an error\n"); return 1; }
    printf("asan ok\n");
    return 0;
}
