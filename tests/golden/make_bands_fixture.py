"""Golden vectors of the reference's band template of the velocity-divergence power spectrum (power_template.py:868-970) under a Kaiser tracer with a binning window,
run with the reference's own code:

    python tests/golden/make_bands_fixture.py [--boundary]        (build container only; writes tests/golden/bands.npz, boundary_bands.npz)
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402

from desilike.theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles, BandVelocityPowerSpectrumTemplate   # noqa: E402
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable   # noqa: E402
from desilike.likelihoods import ObservablesGaussianLikelihood   # noqa: E402

KP = np.array([0.02, 0.05, 0.09, 0.14, 0.2])


def build():
    template = BandVelocityPowerSpectrumTemplate(z=0.8, kp=KP)
    template.init.params['df'].update(fixed=False)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'dptt1': 1.1}, kedges=np.linspace(0.01, 0.21, 41), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    cov = mg.spd_covariance(80, seed=6, diag=4e4, amp=20.)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    return like, obs, template, cov


def dump():
    like, obs, template, cov = build()
    names = like.varied_params.names()
    theta = mg.sample_theta(like, 32, seed=23)
    rng = np.random.RandomState(24)
    for i, name in enumerate(names):
        if name.startswith('dptt'): theta[:, i] = rng.uniform(0.6, 1.5, len(theta))
    theta[:, names.index('df')] = rng.uniform(0.8, 1.2, len(theta))
    theta[-1, names.index('dptt2')] = 3.5       # outside the prior [0, 3]
    out = mg.run_batch(like, [obs], theta, names)
    c = mg.extract_observable(obs)
    c['template'] = 'bands'
    c['band_kp'], c['band_templates'], c['pk_tt_fid'] = np.asarray(template.kp), np.asarray(template.templates), np.asarray(template.pk_tt_fid)
    mg.save('bands', names=np.array(names), theta=theta, obs0=c, precision=np.asarray(like.precision), covariance=cov,
            priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(mg.prior_spec, like.varied_params)]), **out)
    print(names, out['loglikelihood'][:4])


def boundary():
    import make_boundary_fixture as mb
    like, obs, template, cov = build()
    mb.dump('bands', like, size=16, seed=6)


if __name__ == '__main__':
    dump()
    if '--boundary' in sys.argv: boundary()
