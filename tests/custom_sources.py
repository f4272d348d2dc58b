"""User-potential sources shared by the CPU tests (which compile them with hipcc / g++ here, so the
plugin cache travels to the GPU box) and the GPU parity tests."""
import numpy as np

# U = prm[0] * sum q^4 / 4 with a nearest-neighbour coupling prm[1] * sum (q_j - q_{j+1})^2 / 2:
# polynomial only, so the HIP kernels and the host build agree bit for bit.
QUARTIC = '''
template <class Q>
PBBI_FN T potential(const Q& q, int D, const T* prm) {
    T s = 0, c = 0;
    for (int j = 0; j < D; ++j) s += ((q[j] * q[j]) * (q[j] * q[j]));
    for (int j = 0; j + 1 < D; ++j) { const T d = q[j] - q[j + 1]; c += d * d; }
    return (T(0.25) * prm[0]) * s + (T(0.5) * prm[1]) * c;
}
template <class Q, class G>
PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
    for (int j = 0; j < D; ++j) {
        T gj = prm[0] * ((q[j] * q[j]) * q[j]);
        if (j + 1 < D) gj += prm[1] * (q[j] - q[j + 1]);
        if (j > 0) gj -= prm[1] * (q[j - 1] - q[j]);
        g[j] = gj;
    }
}
'''

# Bayesian logistic regression: the product's own library source (custom.py)
from physicsbasedbayesianinference_amd.custom import COIN_TOSS_SOURCE  # noqa: E402,F401
from physicsbasedbayesianinference_amd.custom import LOGISTIC_REGRESSION_SOURCE as LOGISTIC  # noqa: E402


def logistic_problem(M=40, D=5, seed=0, lam=1.0):
    rs = np.random.RandomState(seed)
    X = rs.standard_normal((M, D))
    w = rs.standard_normal(D)
    y = (rs.uniform(size=M) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)
    prm = np.concatenate([[float(M)], X.ravel(), y, [lam]])
    return X, y, lam, prm


def logistic_numpy(X, y, lam, q):
    """Closed form in NumPy for (D, N) q: (U, grad)."""
    z = X @ q                                    # (M, N)
    U = np.sum(np.logaddexp(0.0, z) - y[:, None] * z, axis=0) + 0.5 * lam * np.sum(q * q, axis=0)
    g = X.T @ (1.0 / (1.0 + np.exp(-z)) - y[:, None]) + lam * q
    return U, g


# ---- potentials WITHOUT a gradient function: the product differentiates them by dual numbers
# (custom.py, csrc/pbbi_autodiff.h).  Same functions as above, written on pbbi_scalar<Q>.
QUARTIC_AD = '''
template <class Q>
PBBI_FN pbbi_scalar<Q> potential(const Q& q, int D, const T* prm) {
    pbbi_scalar<Q> s = 0, c = 0;
    for (int j = 0; j < D; ++j) s += ((q[j] * q[j]) * (q[j] * q[j]));
    for (int j = 0; j + 1 < D; ++j) { const pbbi_scalar<Q> d = q[j] - q[j + 1]; c += d * d; }
    return (T(0.25) * prm[0]) * s + (T(0.5) * prm[1]) * c;
}
'''

LOGISTIC_AD = '''
template <class S> PBBI_FN S softplus(S z) { return (z > 0 ? z : S(0)) + log1p(exp(-fabs(z))); }
template <class Q>
PBBI_FN pbbi_scalar<Q> potential(const Q& q, int D, const T* prm) {
    const int M = (int)prm[0];
    const T* X = prm + 1;
    const T* y = X + (long)M * D;
    const T lam = y[M];
    pbbi_scalar<Q> s = 0, r = 0;
    for (int i = 0; i < M; ++i) {
        pbbi_scalar<Q> z = 0;
        for (int j = 0; j < D; ++j) z += X[i * D + j] * q[j];
        s += softplus(z) - y[i] * z;
    }
    for (int j = 0; j < D; ++j) r += q[j] * q[j];
    return s + (T(0.5) * lam) * r;
}
'''

# the reference's coin toss (samples/NumpyroExamples/CoinToss) in logit space, no gradient given
COIN_TOSS_AD = '''
template <class S> PBBI_FN S softplus(S z) { return (z > 0 ? z : S(0)) + log1p(exp(-fabs(z))); }
template <class Q>
PBBI_FN pbbi_scalar<Q> potential(const Q& q, int D, const T* prm) {
    pbbi_scalar<Q> s = 0;
    for (int j = 0; j < D; ++j) s += prm[2 * j] * softplus(-q[j]) + prm[2 * j + 1] * softplus(q[j]);
    return s;
}
'''
