"""Golden vectors of the reference's fiber-collision kernels in configuration space (window.py:1052-1250: TopHatFiberCollisionsCorrelationFunctionMultipoles and
FiberCollisionsCorrelationFunctionMultipoles) folded into the correlation-function window of a damped-BAO xi_ell likelihood, run with the reference's own code:

    python tests/golden/make_fc_xi_fixture.py        (build container only; writes tests/golden/fc_xi.npz)
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402

from desilike.theories.galaxy_clustering import BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles   # noqa: E402
from desilike.observables.galaxy_clustering import (TracerCorrelationFunctionMultipolesObservable, TopHatFiberCollisionsCorrelationFunctionMultipoles,   # noqa: E402
                                                    FiberCollisionsCorrelationFunctionMultipoles)
from desilike.likelihoods import ObservablesGaussianLikelihood   # noqa: E402

SEP = np.array([0.5, 1., 2., 3.5, 5.])
KERNEL = np.array([0.7, 0.6, 0.4, 0.15, 0.05])


def build(kind, binned):
    template = BAOPowerSpectrumTemplate(z=0.5)
    theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='reciso')
    if kind == 'tophat': fiber = TopHatFiberCollisionsCorrelationFunctionMultipoles(fs=0.6, Dfc=4.)
    elif kind == 'tophat_cut': fiber = TopHatFiberCollisionsCorrelationFunctionMultipoles(fs=0.6, Dfc=4., mu_range_cut=True, with_uncorrelated=False)
    else: fiber = FiberCollisionsCorrelationFunctionMultipoles(sep=SEP, kernel=KERNEL)
    kw = dict(sedges=np.linspace(20., 170., 31), wmatrix={'resolution': 2}) if binned else dict(s=np.linspace(22.5, 167.5, 30))
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, ells=(0, 2), theory=theory, fiber_collisions=fiber, **kw)
    for name in ['sigmapar', 'sigmaper']:
        theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
    rng = np.random.RandomState(4)
    A = rng.standard_normal((60, 60)) * 3e-4
    cov = A.dot(A.T) + (3e-3)**2 * np.eye(60)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    return like, obs, fiber, cov


def dump():
    out = {'sep': SEP, 'kernel': KERNEL}
    for kind in ['tophat', 'tophat_cut', 'general']:
        for binned in [False, True]:
            tag = kind + ('_binned' if binned else '')
            like, obs, fiber, cov = build(kind, binned)
            names = like.varied_params.names()
            theta = mg.sample_theta(like, 10, seed=9)
            vlike = mg.vmap(like, backend=None, errors='return', return_derived=True)
            (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
            flat = []
            for row in theta[:2]:
                like(**dict(zip(names, row)))
                flat.append(np.asarray(like.flattheory).copy())
            out.update({tag + '/names': np.array(names), tag + '/theta': theta, tag + '/loglikelihood': np.asarray(derived[like._param_loglikelihood]),
                        tag + '/logprior': np.asarray(derived[like._param_logprior]), tag + '/flattheory': np.array(flat), tag + '/flatdata': np.asarray(like.flatdata),
                        tag + '/kernel_correlated': np.asarray(fiber.kernel_correlated), tag + '/kernel_uncorrelated': np.asarray(fiber.kernel_uncorrelated),
                        tag + '/sin': np.asarray(fiber.s)})
            print(tag, names, np.asarray(derived[like._param_loglikelihood])[:3], np.asarray(fiber.kernel_correlated).shape)
    out['covariance'] = cov
    np.savez(os.path.join(here, 'fc_xi.npz'), **out)


if __name__ == '__main__':
    dump()
