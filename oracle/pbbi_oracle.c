/*
 * pbbi_oracle.c -- CPU restatement of the reference's ensemble-HMC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (physicsbasedbayesianinference_amd/) never imports, links or
 * executes anything under oracle/.
 *
 * Parity pinning: PINNED.  Every function below is checked in
 * tests/test_oracle_golden.py against golden vectors produced by running the
 * reference's own unmodified Python (tests/golden/gen_golden.py) on identical
 * seeds: trajectories <= 1e-12 relative, reject masks equal.
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose operation ORDER it follows: per-chain outer loop, sequential dot
 * products, no FMA contraction (build with -ffp-contract=off).
 *
 * Layout: all state arrays are (D, N) C-order with leading stride ldn >= N,
 * i.e. element (d, n) at [d*ldn + n] -- chain index fastest, exactly the
 * reference's np.zeros((numDimensions, numParticles)) (src/ensemble.py:40-41).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_MAXD 8192

enum { POT_HARMONIC = 0, POT_GAUSS_DIAG = 1, POT_GAUSS_DENSE = 2, POT_ROSENBROCK = 3, POT_CUSTOM = 4 };
enum { METHOD_LEAPFROG = 0, METHOD_STORMER_VERLET = 1 };
/* compat flag bit 0: reproduce src/HMC.py:176 (rejected momentum <- oldQ) */
enum { COMPAT_P_FROM_OLDQ = 1, BETA_ACCEPT = 4 /* include/pbbi.h: PBBI_BETA_ACCEPT */, DRAW_F64 = 32 /* PBBI_DRAW_F64 */ };

typedef struct {
    int kind;
    int D;
    const double* mean; /* D entries, or NULL (= 0)                         */
    const double* prec; /* harmonic: spring consts (D); diag: precisions (D);
                           dense: D x D row-major precision matrix           */
    double cst;         /* additive constant of U                            */
    double a, b, s;     /* Rosenbrock parameters                             */
    /* POT_CUSTOM: the user's potential / gradient source (the one the product compiles into its
     * HIP kernels, physicsbasedbayesianinference_amd/custom.py) compiled for the host by
     * oracle.py::pot_custom; `prec` carries its parameter array.  Stands for the reference's
     * arbitrary callables (src/HMC.py:52-60, src/integrator.py:73).                          */
    double (*user_U)(const double* q, int D, const double* prm);
    void (*user_grad)(const double* q, double* g, int D, const double* prm);
} oracle_pot;

int oracle_version(void) { return 1; }

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* --------------------------------------------------------------- potentials
 * U(q) and grad U(q) for ONE chain; q, g are contiguous (D,) vectors.
 * These restate the closed-form NumPy callables the golden generator hands to
 * the reference (tests/golden/gen_golden.py), which in turn stand in for
 *   - harmonicPotentialND                     src/potential.py:18-27
 *   - -multivariate_normal.logpdf(q, mu, cov) src/tests/test_HMC.py:49,125
 *   - Rosenbrock: defined by the build (last row), with pre-combined
 *     constants and explicit fused multiply-adds (see pot_grad); the golden fixture G5 comes
 *     from the plain-NumPy form and agrees to ~1e-16 relative, not bitwise.
 */
static double pot_U(const oracle_pot* P, const double* q) {
    const int D = P->D;
    double acc = 0.0;
    switch (P->kind) {
    case POT_HARMONIC: /* 0.5 * dot(k, q**2)  (src/potential.py:27) */
        for (int d = 0; d < D; ++d) acc += P->prec[d] * (q[d] * q[d]);
        return 0.5 * acc + P->cst;
    case POT_GAUSS_DIAG: /* 0.5 * dot(prec*x, x) + c */
        for (int d = 0; d < D; ++d) {
            const double x = q[d] - (P->mean ? P->mean[d] : 0.0);
            acc += (P->prec[d] * x) * x;
        }
        return 0.5 * acc + P->cst;
    case POT_GAUSS_DENSE: { /* 0.5 * dot(x, P @ x) + c */
        for (int i = 0; i < D; ++i) {
            double gi = 0.0;
            for (int j = 0; j < D; ++j)
                gi += P->prec[(size_t)i * D + j] * (q[j] - (P->mean ? P->mean[j] : 0.0));
            acc += (q[i] - (P->mean ? P->mean[i] : 0.0)) * gi;
        }
        return 0.5 * acc + P->cst;
    }
    case POT_ROSENBROCK: { /* (sum b*t^2 + sum (a-q_i)^2) * (1/s), t = fma(-q_i, q_i, q_{i+1}) */
        double s1 = 0.0, s2 = 0.0;
        const double inv_s = 1.0 / P->s;
        for (int i = 0; i + 1 < D; ++i) {
            const double t = fma(-q[i], q[i], q[i + 1]);
            s1 = fma(P->b * t, t, s1);
        }
        for (int i = 0; i + 1 < D; ++i) {
            const double r = P->a - q[i];
            s2 = fma(r, r, s2);
        }
        return (s1 + s2) * inv_s + P->cst;
    }
    case POT_CUSTOM:
        return P->user_U(q, D, P->prec);
    }
    return NAN;
}

static void pot_grad(const oracle_pot* P, const double* q, double* g) {
    const int D = P->D;
    switch (P->kind) {
    case POT_HARMONIC:
        for (int d = 0; d < D; ++d) g[d] = P->prec[d] * q[d];
        return;
    case POT_GAUSS_DIAG:
        for (int d = 0; d < D; ++d) g[d] = P->prec[d] * (q[d] - (P->mean ? P->mean[d] : 0.0));
        return;
    case POT_GAUSS_DENSE:
        for (int i = 0; i < D; ++i) {
            double gi = 0.0;
            for (int j = 0; j < D; ++j)
                gi += P->prec[(size_t)i * D + j] * (q[j] - (P->mean ? P->mean[j] : 0.0));
            g[i] = gi;
        }
        return;
    case POT_ROSENBROCK: {
        /* The build's own definition (Rosenbrock is not in the reference): constants are
         * pre-combined and fused multiply-adds are part of the definition, because on the GPU
         * the kernel is bound by its fp64 instruction count, not by HBM:
         *   c1 = (-4b)/s, c2 = 2/s, c3 = (2b)/s,  t_i = fma(-q_i, q_i, q_{i+1})
         *   g_i += fma(c1*q_i, t_i, -(c2*(a - q_i)));   g_{i+1} += c3*t_i          (i < D-1) */
        const double inv_s = 1.0 / P->s;
        const double c1 = (-4.0 * P->b) * inv_s, c2 = 2.0 * inv_s, c3 = (2.0 * P->b) * inv_s;
        for (int d = 0; d < D; ++d) g[d] = 0.0;
        for (int i = 0; i + 1 < D; ++i) {
            const double t = fma(-q[i], q[i], q[i + 1]);
            g[i] += fma(c1 * q[i], t, -(c2 * (P->a - q[i])));
            g[i + 1] += c3 * t;
        }
        return;
    }
    case POT_CUSTOM:
        P->user_grad(q, g, D, P->prec);
        return;
    }
}

/* gather / scatter one chain's column */
static void col_get(const double* A, int D, int64_t ldn, int64_t n, double* x) {
    for (int d = 0; d < D; ++d) x[d] = A[(size_t)d * ldn + n];
}
static void col_put(double* A, int D, int64_t ldn, int64_t n, const double* x) {
    for (int d = 0; d < D; ++d) A[(size_t)d * ldn + n] = x[d];
}

/* potential(q) for all chains; grad_out may be NULL.  Mirrors calling the
 * reference's potential / gradient callables column by column. */
int oracle_potential(const oracle_pot* P, const double* q, int64_t N, int64_t ldn,
                     double* U_out, double* grad_out) {
    const int D = P->D;
    if (D > ORACLE_MAXD) return -1;
#pragma omp parallel
    {
        double* x = (double*)malloc(sizeof(double) * 2 * D);
        double* g = x + D;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            col_get(q, D, ldn, n, x);
            if (U_out) U_out[n] = pot_U(P, x);
            if (grad_out) {
                pot_grad(P, x, g);
                col_put(grad_out, D, ldn, n, g);
            }
        }
        free(x);
    }
    return 0;
}

/* ---------------------------------------------------------------- integrators
 * getAccel(i) = -gradient(q[:, i]) / mass[i]            src/integrator.py:61-73
 */
static void accel(const oracle_pot* P, const double* q, double m, double* a, double* tmp) {
    pot_grad(P, q, tmp);
    for (int d = 0; d < P->D; ++d) a[d] = -tmp[d] / m;
}

/* One chain of Leapfrog.integrate                        src/integrator.py:105-120
 *   v = p/m; a = getAccel
 *   repeat L: q += (v*h + (0.5*a)*(h**2)); a' = getAccel; v += (0.5*(a+a'))*h; a = a'
 *   p = v*m
 */
static void leapfrog_chain(const oracle_pot* P, double* q, double* p, double* v, double m,
                           double h, int L, double* a, double* an, double* tmp) {
    const int D = P->D;
    const double h2 = h * h; /* self.stepSize**2 */
    for (int d = 0; d < D; ++d) v[d] = p[d] / m;
    accel(P, q, m, a, tmp);
    for (int j = 0; j < L; ++j) {
        for (int d = 0; d < D; ++d) q[d] += (v[d] * h + (0.5 * a[d]) * h2);
        accel(P, q, m, an, tmp);
        for (int d = 0; d < D; ++d) v[d] += (0.5 * (a[d] + an[d])) * h;
        for (int d = 0; d < D; ++d) a[d] = an[d];
    }
    for (int d = 0; d < D; ++d) p[d] = v[d] * m;
}

/* Leapfrog with the chain's OWN step count (the build's PBBI_PER_CHAIN_STEPS / PBBI_UTURN_STOP,
 * include/pbbi.h; planned in the reference's WeekPlan.md:16-17, not a reference feature): the loop
 * of leapfrog_chain above, at most Ln steps, and under `uturn` left after the first step j at which
 * (q_j - q_0) . v_j < 0 (sequential sum over the dimensions; p = v*m has v's sign).  Returns the
 * steps taken. */
static int leapfrog_chain_dyn(const oracle_pot* P, double* q, double* p, double* v, double m, double h,
                              int Ln, int uturn, double* a, double* an, double* tmp, double* q0) {
    const int D = P->D;
    const double h2 = h * h;
    int steps = 0;
    for (int d = 0; d < D; ++d) v[d] = p[d] / m;
    for (int d = 0; d < D; ++d) q0[d] = q[d];
    accel(P, q, m, a, tmp);
    while (steps < Ln) {
        for (int d = 0; d < D; ++d) q[d] += (v[d] * h + (0.5 * a[d]) * h2);
        accel(P, q, m, an, tmp);
        for (int d = 0; d < D; ++d) v[d] += (0.5 * (a[d] + an[d])) * h;
        for (int d = 0; d < D; ++d) a[d] = an[d];
        ++steps;
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += (q[d] - q0[d]) * v[d];
        if (uturn && dot < 0.0) break;
    }
    for (int d = 0; d < D; ++d) p[d] = v[d] * m;
    return steps;
}

/* One chain of StormerVerlet.integrate                   src/integrator.py:142-163
 *   v = p/m; qPast = q; q = q + v*h + (0.5*a(q))*h**2
 *   repeat L: tmp = q; q = 2*q - qPast + a(q)*h**2; qPast = tmp
 *   v = (q - qPast)/h; p = v*m        (L+1 position steps, backward-difference v)
 */
static void stormer_verlet_chain(const oracle_pot* P, double* q, double* p, double* v, double m,
                                 double h, int L, double* a, double* qpast, double* tmp) {
    const int D = P->D;
    const double h2 = h * h;
    for (int d = 0; d < D; ++d) v[d] = p[d] / m;
    for (int d = 0; d < D; ++d) qpast[d] = q[d];
    accel(P, q, m, a, tmp);
    for (int d = 0; d < D; ++d) q[d] = (q[d] + v[d] * h) + (0.5 * a[d]) * h2;
    for (int j = 0; j < L; ++j) {
        accel(P, q, m, a, tmp);
        for (int d = 0; d < D; ++d) {
            const double cur = q[d];
            q[d] = (2 * cur - qpast[d]) + a[d] * h2;
            qpast[d] = cur;
        }
    }
    for (int d = 0; d < D; ++d) v[d] = (q[d] - qpast[d]) / h;
    for (int d = 0; d < D; ++d) p[d] = v[d] * m;
}

/* integrate() over the whole ensemble, in place.  v_out (D,N) optional
 * (Integrator.v, src/integrator.py:45).  mass NULL = ones. */
int oracle_integrate(const oracle_pot* P, int method, double* q, double* p, const double* mass,
                     int64_t N, int64_t ldn, double h, int L, double* v_out) {
    const int D = P->D;
    if (D > ORACLE_MAXD || L < 0) return -1;
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * 6 * D);
        double *qc = buf, *pc = buf + D, *vc = buf + 2 * D, *a = buf + 3 * D, *b = buf + 4 * D,
               *tmp = buf + 5 * D;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const double m = mass ? mass[n] : 1.0;
            col_get(q, D, ldn, n, qc);
            col_get(p, D, ldn, n, pc);
            if (method == METHOD_LEAPFROG)
                leapfrog_chain(P, qc, pc, vc, m, h, L, a, b, tmp);
            else
                stormer_verlet_chain(P, qc, pc, vc, m, h, L, a, b, tmp);
            col_put(q, D, ldn, n, qc);
            col_put(p, D, ldn, n, pc);
            if (v_out) col_put(v_out, D, ldn, n, vc);
        }
        free(buf);
    }
    return 0;
}

/* ------------------------------------------------------------------ energies
 * H = 0.5*dot(p,p)/mass[i] + potential(q[:,i])            src/HMC.py:100-102
 */
static double hamiltonian(const oracle_pot* P, const double* q, const double* p, double m) {
    double pp = 0.0;
    for (int d = 0; d < P->D; ++d) pp += p[d] * p[d];
    return 0.5 * pp / m + pot_U(P, q);
}

/* HMC.getWeights: exp(-H) per chain                      src/HMC.py:86-104
 * H_out optional. */
int oracle_weights(const oracle_pot* P, const double* q, const double* p, const double* mass,
                   int64_t N, int64_t ldn, double* w_out, double* H_out) {
    const int D = P->D;
    if (D > ORACLE_MAXD) return -1;
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * 2 * D);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            col_get(q, D, ldn, n, buf);
            col_get(p, D, ldn, n, buf + D);
            const double H = hamiltonian(P, buf, buf + D, mass ? mass[n] : 1.0);
            if (H_out) H_out[n] = H;
            if (w_out) w_out[n] = exp(-H);
        }
        free(buf);
    }
    return 0;
}

/* HMC.getWeightsRatio: exp(oldH - newH) per chain        src/HMC.py:106-116 */
int oracle_weights_ratio(const oracle_pot* P, const double* newQ, const double* newP,
                         const double* oldQ, const double* oldP, const double* mass, int64_t N,
                         int64_t ldn, double* ratio_out) {
    const int D = P->D;
    if (D > ORACLE_MAXD) return -1;
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * 2 * D);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const double m = mass ? mass[n] : 1.0;
            col_get(oldQ, D, ldn, n, buf);
            col_get(oldP, D, ldn, n, buf + D);
            const double oldH = hamiltonian(P, buf, buf + D, m);
            col_get(newQ, D, ldn, n, buf);
            col_get(newP, D, ldn, n, buf + D);
            const double newH = hamiltonian(P, buf, buf + D, m);
            ratio_out[n] = exp(oldH - newH);
        }
        free(buf);
    }
    return 0;
}

/* One iteration of the HMC.getSamples loop body           src/HMC.py:154-179
 *   in : q (state, D x N), p (freshly drawn momentum), u (N uniforms)
 *   out: q, p overwritten with what the reference stores into samples_hmc[:,:,i]
 *        and momentum_hmc[:,:,i]; ratio_out (N) and reject_out (N bytes) optional.
 * Semantics restated:
 *   - p = -p only feeds the (p-symmetric) energy; stored momentum is un-negated (:164,:179)
 *   - mask = u > min(1, ratio) is the REJECT mask; NaN ratio => comparison False
 *     => proposal accepted (:168-173)
 *   - rejected: q <- oldQ (:175) and p <- oldQ (:176, reference bug) when
 *     compat & COMPAT_P_FROM_OLDQ, else p <- oldP.
 */
/* beta: factor on (oldH - newH) in the accept test.  1.0 is the reference (src/HMC.py:115 has no
 * temperature); 1/kT is the build's PBBI_BETA_ACCEPT (include/pbbi.h), which matches the momentum
 * draw of src/ensemble.py:88.  x * 1.0 == x bit for bit, so beta = 1 restates the reference exactly. */
int oracle_hmc_iter_beta(const oracle_pot* P, int method, double* q, double* p, const double* u,
                         const double* mass, int64_t N, int64_t ldn, double h, int L, int compat,
                         double beta, double* ratio_out, unsigned char* reject_out);
/* ... and with per-chain trajectory lengths: steps_in (N, each clamped to [0, L]; NULL = L), `uturn`
 * (stop at the first U-turn), steps_out (N, optional).  Leapfrog only when either is in use. */
int oracle_hmc_iter_dyn(const oracle_pot* P, int method, double* q, double* p, const double* u,
                        const double* mass, int64_t N, int64_t ldn, double h, int L, int compat,
                        double beta, const int32_t* steps_in, int uturn, int32_t* steps_out,
                        double* ratio_out, unsigned char* reject_out);

int oracle_hmc_iter(const oracle_pot* P, int method, double* q, double* p, const double* u,
                    const double* mass, int64_t N, int64_t ldn, double h, int L, int compat,
                    double* ratio_out, unsigned char* reject_out) {
    return oracle_hmc_iter_beta(P, method, q, p, u, mass, N, ldn, h, L, compat, 1.0, ratio_out, reject_out);
}

int oracle_hmc_iter_beta(const oracle_pot* P, int method, double* q, double* p, const double* u,
                         const double* mass, int64_t N, int64_t ldn, double h, int L, int compat,
                         double beta, double* ratio_out, unsigned char* reject_out) {
    return oracle_hmc_iter_dyn(P, method, q, p, u, mass, N, ldn, h, L, compat, beta, NULL, 0, NULL, ratio_out,
                               reject_out);
}

int oracle_hmc_iter_dyn(const oracle_pot* P, int method, double* q, double* p, const double* u,
                        const double* mass, int64_t N, int64_t ldn, double h, int L, int compat,
                        double beta, const int32_t* steps_in, int uturn, int32_t* steps_out,
                        double* ratio_out, unsigned char* reject_out) {
    const int D = P->D;
    const int dyn = (steps_in != NULL) || uturn;
    if (dyn && method != METHOD_LEAPFROG) return -2;
    if (D > ORACLE_MAXD || L < 0) return -1;
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * 9 * D);
        double *qc = buf, *pc = buf + D, *vc = buf + 2 * D, *a = buf + 3 * D, *b = buf + 4 * D,
               *tmp = buf + 5 * D, *oq = buf + 6 * D, *op = buf + 7 * D, *q0 = buf + 8 * D;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const double m = mass ? mass[n] : 1.0;
            col_get(q, D, ldn, n, qc);
            col_get(p, D, ldn, n, pc);
            memcpy(oq, qc, sizeof(double) * D); /* oldQ = np.copy(q)  :156 */
            memcpy(op, pc, sizeof(double) * D); /* oldP = np.copy(p)  :157 */
            int taken = L;
            if (dyn) {
                int Ln = steps_in ? steps_in[n] : L;
                Ln = Ln < 0 ? 0 : (Ln > L ? L : Ln);
                taken = leapfrog_chain_dyn(P, qc, pc, vc, m, h, Ln, uturn, a, b, tmp, q0);
            } else if (method == METHOD_LEAPFROG)
                leapfrog_chain(P, qc, pc, vc, m, h, L, a, b, tmp);
            else
                stormer_verlet_chain(P, qc, pc, vc, m, h, L, a, b, tmp);
            for (int d = 0; d < D; ++d) tmp[d] = -pc[d]; /* p = -p  :164 */
            const double oldH = hamiltonian(P, oq, op, m);
            const double newH = hamiltonian(P, qc, tmp, m);
            const double ratio = exp((oldH - newH) * beta);        /* :115 */
            const double acc = (1.0 < ratio || ratio != ratio) ? ((ratio != ratio) ? ratio : 1.0)
                                                               : ratio; /* np.minimum(1, ratio) */
            const int reject = (u[n] > acc); /* False when acc is NaN  :173 */
            if (reject) {
                col_put(q, D, ldn, n, oq);
                col_put(p, D, ldn, n, (compat & COMPAT_P_FROM_OLDQ) ? oq : op);
            } else {
                col_put(q, D, ldn, n, qc);
                col_put(p, D, ldn, n, pc);
            }
            if (ratio_out) ratio_out[n] = ratio;
            if (reject_out) reject_out[n] = (unsigned char)reject;
            if (steps_out) steps_out[n] = taken;
        }
        free(buf);
    }
    return 0;
}

/* One iteration of the self-tuning ("GIST", Bou-Rabee, Carpenter & Marsden 2024) no-U-turn sampler: the
 * reversible per-chain dynamic trajectory length the reference plans ("no u-turn sampling",
 * references/PhysicsBasedHMC_SoHPC2022_WeekPlan.md:16-17) -- the build's own definition, include/pbbi.h
 * pbbi_hmc_run_gist.  With tau(q, p) = number of leapfrog steps until (q_j - q_0) . p_j < 0 first holds,
 * at most Lmax (leapfrog_chain_dyn with uturn):
 *   tau_f = tau(q, p);   L = 1 + floor(u_len * tau_f) (capped at tau_f): uniform on 1..tau_f;
 *   (q', p') = L leapfrog steps from (q, p);   tau_b = tau(q', -p');
 *   accept with probability min(1, exp(beta (H - H')) * tau_f / tau_b * [L <= tau_b]).
 * The length is a Gibbs draw from a state-dependent distribution and the Metropolis ratio carries that
 * distribution's density at both ends, so the chain keeps exp(-beta H) invariant although L adapts to the
 * local geometry.  tau_out: (3, N) = tau_f, L, tau_b.  Rejected chains as in oracle_hmc_iter (compat). */
int oracle_hmc_iter_gist(const oracle_pot* P, double* q, double* p, const double* u_acc, const double* u_len,
                         const double* mass, int64_t N, int64_t ldn, double h, int Lmax, int compat, double beta,
                         double* ratio_out, unsigned char* reject_out, int32_t* tau_out) {
    const int D = P->D;
    if (D > ORACLE_MAXD || Lmax < 1) return -1;
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * 11 * D);
        double *qc = buf, *pc = buf + D, *vc = buf + 2 * D, *a = buf + 3 * D, *b = buf + 4 * D, *tmp = buf + 5 * D,
               *oq = buf + 6 * D, *op = buf + 7 * D, *q0 = buf + 8 * D, *qb = buf + 9 * D, *pb = buf + 10 * D;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const double m = mass ? mass[n] : 1.0;
            col_get(q, D, ldn, n, oq);
            col_get(p, D, ldn, n, op);
            memcpy(qc, oq, sizeof(double) * D);
            memcpy(pc, op, sizeof(double) * D);
            const int tau_f = leapfrog_chain_dyn(P, qc, pc, vc, m, h, Lmax, 1, a, b, tmp, q0);
            int L = 1 + (int)(u_len[n] * (double)tau_f);
            if (L > tau_f) L = tau_f;
            memcpy(qc, oq, sizeof(double) * D);
            memcpy(pc, op, sizeof(double) * D);
            leapfrog_chain_dyn(P, qc, pc, vc, m, h, L, 0, a, b, tmp, q0);
            for (int d = 0; d < D; ++d) { qb[d] = qc[d]; pb[d] = -pc[d]; }
            const double oldH = hamiltonian(P, oq, op, m);
            const double newH = hamiltonian(P, qc, pb, m);
            const int tau_b = leapfrog_chain_dyn(P, qb, pb, vc, m, h, Lmax, 1, a, b, tmp, q0);
            const double hratio = exp((oldH - newH) * beta);
            const double ratio = (L <= tau_b) ? hratio * ((double)tau_f / (double)tau_b) : 0.0;
            const double acc = (1.0 < ratio || ratio != ratio) ? ((ratio != ratio) ? ratio : 1.0) : ratio;
            const int reject = (u_acc[n] > acc);
            if (reject) {
                col_put(q, D, ldn, n, oq);
                col_put(p, D, ldn, n, (compat & COMPAT_P_FROM_OLDQ) ? oq : op);
            } else {
                col_put(q, D, ldn, n, qc);
                col_put(p, D, ldn, n, pc);
            }
            if (ratio_out) ratio_out[n] = ratio;
            if (reject_out) reject_out[n] = (unsigned char)reject;
            if (tau_out) { tau_out[n] = tau_f; tau_out[N + n] = L; tau_out[2 * N + n] = tau_b; }
        }
        free(buf);
    }
    return 0;
}

/* ----------------------------------------------------------------- Philox RNG
 * Counter-based generator of the device ("philox") mode; the product's HIP
 * kernels implement the same contract (include/pbbi.h, "RNG contract").
 * Philox-4x32-10 (Salmon et al., SC'11; Random123 constants).  This is the
 * build's own design, not a reference feature: the reference only has the
 * global NumPy RandomState (src/ensemble.py:72-74,88-91).
 */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox_raw(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    philox4x32_10(ctr, key, out);
}

enum { STREAM_MOMENTUM = 0, STREAM_POSITION = 1, STREAM_UNIFORM = 2 };

static void rng_block(uint64_t seed, uint32_t stream, uint64_t iter, uint64_t chain, uint32_t blk,
                      uint32_t out[4]) {
    const uint32_t ctr[4] = {(uint32_t)chain, blk, (uint32_t)iter,
                             (stream & 0xFFu) | ((uint32_t)(chain >> 32) << 8)};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    philox4x32_10(ctr, key, out);
}

static double u53(uint32_t lo, uint32_t hi) { /* [0,1) with 53 random bits */
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53;
}

/* Single-precision Box-Muller on 2 x 32 bits (host mirror of csrc/pbbi_rng.h::box_muller_f32).
 *   u1 = a*2^-32 + 2^-33 in (0,1],  u2 = (b>>8)*2^-24 in [0,1),  r = sqrt(-2 ln 2 * log2(u1)).
 * The device evaluates log2/sqrt/sin/cos on its transcendental unit (v_log_f32, ...); this
 * mirror uses libm's float functions, so the two agree to ~1e-6 absolute, not bitwise. */
static void box_muller_f32(uint32_t a, uint32_t b, float* zc, float* zs) {
    const float u1 = fmaf((float)a, 0x1.0p-32f, 0x1.0p-33f);
    const float u2 = (float)(b >> 8) * 0x1.0p-24f;
    const float r = sqrtf(-1.3862943611198906f * log2f(u1));
    const double ang = 6.283185307179586476925 * (double)u2;
    *zc = r * (float)cos(ang);
    *zs = r * (float)sin(ang);
}

/* ---- double-precision draw (flag PBBI_DRAW_F64 / stream bit 0x100; include/pbbi.h "RNG contract").
 * Restated here in plain C from the contract's text: the transform is built from +, -, *, /, sqrt and fma
 * only (each correctly rounded on the host and on gfx950) in a FIXED order, so the device's variates and
 * these agree bit for bit -- unlike the single-precision draw, whose transcendental unit has no host twin.
 *   w1 = x1:x0, w2 = x3:x2 of ONE Philox block;
 *   u1 = ((w1 >> 12) + 0.5) * 2^-52  in (0, 1)   (tails to 8.57 sigma);
 *   k2 = w2 >> 11 (53 bits), angle = 2 pi k2 2^-53 = (pi/2)(n + y),  n = (k2 + 2^50) >> 51,
 *        y = (k2 - n 2^51) 2^-51 in [-1/2, 1/2)  (exact integer arithmetic);
 *   r = sqrt(-2 ln u1): ln by fdlibm's e_log.c scheme (u = 2^k m, m in [sqrt(1/2), sqrt 2), f = m - 1,
 *        s = f / (2 + f), minimax polynomial in s^2 with the Lg1..Lg7 of that file, Horner with fma);
 *   sin / cos((pi/2) y): Taylor polynomials in y^2 (9 / 10 terms, coefficients (pi/2)^j / j! rounded from
 *        60-digit decimals by tools/gen_draw_coeffs.py), Horner with fma; rotated by the quadrant n & 3;
 *   z_even = r cos(angle), z_odd = r sin(angle).
 * Block use: the four dims {d, d+4, d+8, d+12} of a group of 16 take blk = ((dim>>4)<<2)|(dim&3) for slots
 * 0, 1 and blk | 0x80000000 for slots 2, 3 (two blocks where the single-precision draw needs one). */
static double draw_log_unit(double u) { /* ln u, u a normal double in (0, 1] */
    static const double ln2_hi = 0x1.62e42fee00000p-1, ln2_lo = 0x1.a39ef35793c76p-33,
                        Lg1 = 0x1.5555555555593p-1, Lg2 = 0x1.999999997fa04p-2, Lg3 = 0x1.2492494229359p-2,
                        Lg4 = 0x1.c71c51d8e78afp-3, Lg5 = 0x1.7466496cb03dep-3, Lg6 = 0x1.39a09d078c69fp-3,
                        Lg7 = 0x1.2f112df3e5244p-3;
    uint64_t bits;
    memcpy(&bits, &u, 8);
    int k = (int)(bits >> 52) - 1023;
    const uint64_t mant = bits & 0xFFFFFFFFFFFFFull;
    uint64_t mb;
    if (mant > 0x6A09E667F3BCCull) { k += 1; mb = mant | (1022ull << 52); }
    else mb = mant | (1023ull << 52);
    double m;
    memcpy(&m, &mb, 8);
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

static void draw_sincos_quarter(double y, double* sn, double* cs) { /* sin, cos of (pi/2) y, |y| <= 1/2 */
    static const double S[9] = {0x1.921fb54442d18p+0, -0x1.4abbce625be53p-1, 0x1.466bc6775aae2p-4,
                                -0x1.32d2cce62bd86p-8, 0x1.50783487ee782p-13, -0x1.e3074fde8871fp-19,
                                0x1.e8f434d018d63p-25, -0x1.6fadb9f155744p-31, 0x1.aaec32af93359p-38};
    static const double Cc[10] = {0x1.0000000000000p+0, -0x1.3bd3cc9be45dep+0, 0x1.03c1f081b5ac4p-2,
                                  -0x1.55d3c7e3cbffap-6, 0x1.e1f506891babbp-11, -0x1.a6d1f2a204a8cp-16,
                                  0x1.f9d38a3763cc3p-22, -0x1.b6e24f44b128fp-28, 0x1.20c62c2f2d7f5p-34,
                                  -0x1.2a0c591af8314p-41};
    const double z = y * y;
    double s = S[8], c = Cc[9];
    for (int k = 7; k >= 0; --k) s = fma(s, z, S[k]);
    for (int k = 8; k >= 0; --k) c = fma(c, z, Cc[k]);
    *sn = y * s;
    *cs = c;
}

static void box_muller_f64(const uint32_t x[4], double* zc, double* zs) {
    const uint64_t w1 = ((uint64_t)x[1] << 32) | x[0], w2 = ((uint64_t)x[3] << 32) | x[2];
    const double u1 = ((double)(w1 >> 12) + 0.5) * 0x1.0p-52;
    const uint64_t k2 = w2 >> 11;
    const uint64_t n = (k2 + (1ull << 50)) >> 51;
    const double y = (double)((int64_t)k2 - (int64_t)(n << 51)) * 0x1.0p-51;
    double sn, cs;
    draw_sincos_quarter(y, &sn, &cs);
    const double r = sqrt(-2.0 * draw_log_unit(u1));
    double c, s;
    switch ((int)(n & 3)) {
        case 0: c = cs; s = sn; break;
        case 1: c = -sn; s = cs; break;
        case 2: c = -cs; s = -sn; break;
        default: c = sn; s = -cs; break;
    }
    *zc = r * c;
    *zs = r * s;
}

static double rng_normal_f64(uint64_t seed, uint32_t stream, uint64_t iter, uint64_t chain, int dim) {
    uint32_t x[4];
    double z[2];
    const int slot = (dim >> 2) & 3;
    const uint32_t blk = (uint32_t)(((dim >> 4) << 2) | (dim & 3)) | (slot >= 2 ? 0x80000000u : 0u);
    rng_block(seed, stream, iter, chain, blk, x);
    box_muller_f64(x, &z[0], &z[1]);
    return z[slot & 1];
}

/* standard normal for element (dim, chain) of draw `iter` in `stream`:
 *   block index blk = ((dim >> 4) << 2) | (dim & 3): dims d, d+4, d+8, d+12 (same d mod 4
 *   inside a group of 16) share a block; slot = (dim >> 2) & 3 picks
 *   0: r1 cos, 1: r1 sin  (from x0, x1);  2: r2 cos, 3: r2 sin  (from x2, x3). */
static double rng_normal(uint64_t seed, uint32_t stream, uint64_t iter, uint64_t chain, int dim) {
    uint32_t x[4];
    float z[4];
    rng_block(seed, stream, iter, chain, (uint32_t)(((dim >> 4) << 2) | (dim & 3)), x);
    box_muller_f32(x[0], x[1], &z[0], &z[1]);
    box_muller_f32(x[2], x[3], &z[2], &z[3]);
    return (double)z[(dim >> 2) & 3];
}

static double rng_uniform(uint64_t seed, uint64_t iter, uint64_t chain) {
    uint32_t x[4];
    rng_block(seed, STREAM_UNIFORM, iter, chain, 0xFFFFFFFFu, x);
    return u53(x[0], x[1]);
}

/* PBBI_PER_CHAIN_STEPS draw (include/pbbi.h): 1 + floor(u * L) capped at L, u from STREAM_STEPS = 3 */
int oracle_philox_steps(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, int L, int32_t* out) {
    for (int64_t n = 0; n < N; ++n) {
        uint32_t x[4];
        rng_block(seed, 3u, iter, chain0 + n, 0xFFFFFFFFu, x);
        int s = 1 + (int)(u53(x[0], x[1]) * (double)L);
        out[n] = L > 0 ? (s > L ? L : s) : 0;
    }
    return 0;
}

/* Replica exchange between temperature rungs (include/pbbi.h pbbi_replica_exchange; the build's own definition):
 * rung r = chains [r*Nr, (r+1)*Nr); for r = parity, parity+2, ... chain n of rung r and of rung r+1 swap positions
 * when u < exp((beta_r - beta_{r+1}) (U_r - U_{r+1})), u = the lower chain's PBBI_STREAM_SWAP (4) uniform. */
int oracle_replica_exchange(const oracle_pot* P, double* q, int64_t Nr, int R, int64_t ldn, const double* betas,
                            int parity, uint64_t seed, uint64_t iter, uint64_t chain0, unsigned char* swapped_out) {
    const int D = P->D;
    if (D > ORACLE_MAXD) return -1;
    double* a = (double*)malloc(sizeof(double) * 2 * D);
    double* b = a + D;
    for (int r = parity; r + 1 < R; r += 2)
        for (int64_t n = 0; n < Nr; ++n) {
            const int64_t lo = (int64_t)r * Nr + n, hi = lo + Nr;
            col_get(q, D, ldn, lo, a);
            col_get(q, D, ldn, hi, b);
            const double arg = (betas[r] - betas[r + 1]) * (pot_U(P, a) - pot_U(P, b));
            uint32_t x[4];
            rng_block(seed, 4u, iter, chain0 + (uint64_t)lo, 0xFFFFFFFFu, x);
            const int swap = u53(x[0], x[1]) < exp(arg);
            if (swapped_out) swapped_out[(int64_t)r * Nr + n] = (unsigned char)swap;
            if (swap) {
                col_put(q, D, ldn, lo, b);
                col_put(q, D, ldn, hi, a);
            }
        }
    free(a);
    return 0;
}

/* the uniform of PBBI_STREAM_STEPS (what PBBI_PER_CHAIN_STEPS and the GIST length draw consume) */
int oracle_philox_steps_uniform(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, double* out) {
    for (int64_t n = 0; n < N; ++n) {
        uint32_t x[4];
        rng_block(seed, 3u, iter, chain0 + n, 0xFFFFFFFFu, x);
        out[n] = u53(x[0], x[1]);
    }
    return 0;
}

/* out[d*ldn + n] = scale_n * z(d, chain0 + n); scale (N) optional else scalar.
 * stream | 0x100 (PBBI_STREAM_DRAW_F64): the double-precision draw above. */
int oracle_philox_normal(uint64_t seed, int stream, uint64_t iter, uint64_t chain0, int D, int64_t N,
                         int64_t ldn, double scale_scalar, const double* scale_per_chain,
                         double* out) {
    const int f64 = (stream & 0x100) != 0;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const double sc = scale_per_chain ? scale_per_chain[n] : scale_scalar;
        for (int d = 0; d < D; ++d)
            out[(size_t)d * ldn + n] = (f64 ? rng_normal_f64(seed, (uint32_t)stream, iter, chain0 + n, d)
                                            : rng_normal(seed, (uint32_t)stream, iter, chain0 + n, d)) * sc;
    }
    return 0;
}

int oracle_philox_uniform(uint64_t seed, uint64_t iter, uint64_t chain0, int64_t N, double* out) {
    for (int64_t n = 0; n < N; ++n) out[n] = rng_uniform(seed, iter, chain0 + n);
    return 0;
}

/* getSamples loop (src/HMC.py:150-179) driven by the Philox contract:
 *   iteration i (global index iter0 + i): p = sqrt(mass*kT) * z, u from STREAM_UNIFORM.
 *   samples_out / momenta_out: (S, D, N) slabs (momenta_out may be NULL),
 *   reject_out (S, N) bytes, ratio_out (S, N) optional.  q is the state (in/out).
 */
int oracle_hmc_run_philox(const oracle_pot* P, int method, double* q, const double* mass,
                          int64_t N, int64_t ldn, double h, int L, int S, int compat,
                          uint64_t seed, uint64_t iter0, uint64_t chain0, double kT,
                          double* samples_out, double* momenta_out, unsigned char* reject_out,
                          double* ratio_out) {
    const int D = P->D;
    double* p = (double*)malloc(sizeof(double) * (size_t)D * ldn);
    double* u = (double*)malloc(sizeof(double) * N);
    double* pstd = (double*)malloc(sizeof(double) * N);
    for (int64_t n = 0; n < N; ++n) pstd[n] = sqrt((mass ? mass[n] : 1.0) * kT);
    int rc = 0;
    for (int i = 0; i < S && rc == 0; ++i) {
        oracle_philox_normal(seed, STREAM_MOMENTUM | ((compat & DRAW_F64) ? 0x100 : 0), iter0 + i, chain0, D, N,
                             ldn, 1.0, pstd, p);
        oracle_philox_uniform(seed, iter0 + i, chain0, N, u);
        rc = oracle_hmc_iter_beta(P, method, q, p, u, mass, N, ldn, h, L, compat,
                                  (compat & BETA_ACCEPT) ? 1.0 / kT : 1.0,
                                  ratio_out ? ratio_out + (size_t)i * N : NULL,
                                  reject_out ? reject_out + (size_t)i * N : NULL);
        for (int d = 0; d < D; ++d) {
            memcpy(samples_out + ((size_t)i * D + d) * N, q + (size_t)d * ldn, sizeof(double) * N);
            if (momenta_out)
                memcpy(momenta_out + ((size_t)i * D + d) * N, p + (size_t)d * ldn,
                       sizeof(double) * N);
        }
    }
    free(p); free(u); free(pstd);
    return rc;
}
