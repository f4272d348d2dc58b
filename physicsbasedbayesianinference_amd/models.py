"""Bayesian models as potentials, written on the traceable array namespace (trace.py) -- SURVEY 8f row 1, the
model -> potential boundary the reference reaches for with NumPyro (samples/NumpyroExamples/).

`eight_schools_potential` is the reference's hierarchical example
(samples/NumpyroExamples/eight_schools.py:5-10, data samples/NumpyroExamples/eight_schools.data.json):

    mu ~ Normal(0, 5);  tau ~ HalfCauchy(5);  theta_j ~ Normal(mu, tau);  y_j ~ Normal(theta_j, sigma_j)

as -log posterior over the unconstrained vector x = (mu, log tau, eta_1..J) (non-centred: theta = mu + tau eta,
the parametrisation fixed-step HMC can sample the funnel in) or x = (mu, log tau, theta_1..J) (centred, the
reference's own).  The functions are plain Python on `trace`: handing one to HMC / trace_potential compiles it
(potential AND symbolic gradient) into the ensemble-HMC kernels -- there is no CPU evaluation.
"""
import math

import numpy as np

from . import trace as tnp

__all__ = ["EIGHT_SCHOOLS_Y", "EIGHT_SCHOOLS_SIGMA", "eight_schools_potential", "eight_schools_constrain"]

# Rubin (1981); the values of the reference's samples/NumpyroExamples/eight_schools.data.json
EIGHT_SCHOOLS_Y = np.array([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0])
EIGHT_SCHOOLS_SIGMA = np.array([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0])


def eight_schools_potential(y=EIGHT_SCHOOLS_Y, sigma=EIGHT_SCHOOLS_SIGMA, centered=False, mu_scale=5.0,
                            tau_scale=5.0):
    """potential(x) for x = (mu, log tau, eta or theta), D = J + 2; additive constants dropped."""
    y, sigma = np.asarray(y, dtype=np.float64), np.asarray(sigma, dtype=np.float64)
    J = y.size

    def potential(x):
        mu, log_tau = x[0], x[1]
        tau = tnp.exp(log_tau)
        z = x[2:]
        U = 0.5 * (mu / mu_scale) ** 2                                   # mu ~ Normal(0, mu_scale)
        U = U + tnp.log(1.0 + (tau / tau_scale) ** 2) - log_tau          # tau ~ HalfCauchy(tau_scale); |d tau / d log tau|
        if centered:
            theta = z
            U = U + 0.5 * tnp.sum(((theta - mu) / tau) ** 2) + J * log_tau   # theta ~ Normal(mu, tau)
        else:
            theta = mu + tau * z
            U = U + 0.5 * tnp.sum(z * z)                                 # eta ~ Normal(0, 1)
        return U + 0.5 * tnp.sum(((y - theta) / sigma) ** 2)             # y ~ Normal(theta, sigma)
    potential.numDimensions = J + 2
    return potential


def eight_schools_constrain(samples, centered=False):
    """(D, ...) unconstrained draws -> dict(mu, tau, theta) like NumPyro's get_samples()."""
    s = np.asarray(samples)
    mu, tau = s[0], np.exp(s[1])
    theta = s[2:] if centered else mu[None] + tau[None] * s[2:]
    return {"mu": mu, "tau": tau, "theta": theta}
