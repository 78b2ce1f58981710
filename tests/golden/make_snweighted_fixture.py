"""Golden vectors of the reference's Fisher-forecast likelihood ``SNWeightedPowerSpectrumLikelihood`` (likelihoods/galaxy_clustering/fisher.py:10-71: signal-to-noise
weighted P(k, mu) on 500 wavenumbers x Gauss-Legendre cosines, diagonal precision from the footprint's volume and shot noise), run here with the reference's own code:

    python tests/golden/make_snweighted_fixture.py        (build container only; writes tests/golden/snweighted.npz)

Harness shim (as tests/golden/make_tns_fixture.py): ``utils.weights_trapz`` calls ``jnp.insert`` with an index one past the end (jax clamps it, numpy raises); the
same weights are computed without the insertion.
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402

from desilike import utils   # noqa: E402


def _weights_trapz(x):
    x = np.asarray(x, dtype='f8')
    return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.


utils.weights_trapz = _weights_trapz

from desilike.theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles, ShapeFitPowerSpectrumTemplate   # noqa: E402
from desilike.likelihoods.galaxy_clustering import SNWeightedPowerSpectrumLikelihood   # noqa: E402
from desilike.observables.galaxy_clustering import BoxFootprint   # noqa: E402


def dump(name='snweighted', size=16, seed=21, mu=8, klim=(0.01, 0.2)):
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.8))
    footprint = BoxFootprint(volume=2e9, nbar=5e-4)
    likelihood = SNWeightedPowerSpectrumLikelihood(theories=[theory], data={'b1': 2., 'sn0': 0.1}, covariance={'b1': 1.9}, footprints=[footprint], klim=klim, mu=mu)
    likelihood()
    names = likelihood.varied_params.names()
    theta = mg.sample_theta(likelihood, size, seed)
    theta[-1, names.index('b1')] = -1.     # outside the prior
    vlike = mg.vmap(likelihood, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({pname: theta[:, i] for i, pname in enumerate(names)})
    flattheory = []
    for row in theta[:3]:
        likelihood(**dict(zip(names, row)))
        flattheory.append(np.array(likelihood.flattheory))
    out = dict(names=np.array(names), theta=theta, loglikelihood=np.array(derived[likelihood._param_loglikelihood]), logprior=np.array(derived[likelihood._param_logprior]),
               flatdata=np.array(likelihood.flatdata), precision=np.array(likelihood.precision), flattheory=np.array(flattheory), mu=mu, klim=np.array(klim),
               volume=footprint.volume, shotnoise=footprint.shotnoise, k=np.array(theory.k), ells=np.array(theory.ells), errors=np.array(sorted(errors), dtype='i8'),
               z=0.8, theory_shotnoise=float(getattr(theory, 'nd', np.nan)) if hasattr(theory, 'nd') else np.nan)
    np.savez(os.path.join(here, name + '.npz'), **out)
    print(name, names, 'n =', likelihood.flatdata.size, 'loglikelihood', out['loglikelihood'][:4], 'errors', errors)


if __name__ == '__main__':
    dump()
