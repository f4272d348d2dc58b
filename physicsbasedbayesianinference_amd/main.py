"""Tiny driver for BASELINE config 1 (the reference's `src/main.py` is an empty file):

    python -m physicsbasedbayesianinference_amd.main [--dims 1] [--particles 32] [--samples 100]

1-D standard-Gaussian potential, ensemble of 32 chains, simulTime 1.0 / stepSize 0.1 = 10
leapfrog steps per HMC iteration, T = 1/kB, NumPy-stream RNG (seed 1234): the run the golden
vector G3 pins against the reference.  Prints sample statistics and the accept rate."""
import argparse

import numpy as np
from scipy.constants import k as kB

from . import HMC, Ensemble, StandardGaussian


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--dims", type=int, default=1)
    ap.add_argument("--particles", type=int, default=32)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--rng", choices=["numpy", "philox"], default="numpy")
    args = ap.parse_args(argv)
    np.random.seed(args.seed)
    ens = Ensemble(args.dims, args.particles)
    pot = StandardGaussian(args.dims, const=0.5 * args.dims * np.log(2 * np.pi))
    hmc = HMC(ens, 1.0, 0.1, pot.density, potential=pot, rng=args.rng, seed=args.seed)
    samples, momenta = hmc.getSamples(args.samples, 1.0 / kB, 1.0)
    print(f"samples {samples.shape}: mean {samples.mean():+.4f}  var {samples.var():.4f}  "
          f"(target 0, 1)   accept rate {hmc.acceptRate:.3f}")
    return samples, momenta


if __name__ == "__main__":
    main()
