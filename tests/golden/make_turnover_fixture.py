"""Golden vectors of the reference's turn-over template (power_template.py:1293-1340) under a Kaiser tracer with a binning window, run with the reference's own code:

    python tests/golden/make_turnover_fixture.py        (build container only; writes tests/golden/turnover.npz)
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402

from desilike.theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles, TurnOverPowerSpectrumTemplate   # noqa: E402
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable   # noqa: E402
from desilike.likelihoods import ObservablesGaussianLikelihood   # noqa: E402


def build():
    template = TurnOverPowerSpectrumTemplate(z=0.8)
    for name in ['dpto', 'qap', 'df']: template.init.params[name].update(fixed=False)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    kedges = np.linspace(0.001, 0.101, 41)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'm': 0.6, 'n': 0.9}, kedges=kedges, ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    cov = mg.spd_covariance(80, seed=3, diag=400., amp=2.)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    return like, obs, template, cov


def dump():
    like, obs, template, cov = build()
    names = like.varied_params.names()
    theta = mg.sample_theta(like, 32, seed=17)
    rng = np.random.RandomState(18)
    for pname, (lo, hi) in {'m': (0.2, 1.2), 'n': (0.5, 1.4), 'qto': (0.9, 1.1), 'dpto': (0.8, 1.2), 'qap': (0.9, 1.1), 'df': (0.8, 1.2)}.items():
        theta[:, names.index(pname)] = rng.uniform(lo, hi, len(theta))
    theta[-1, names.index('qto')] = 1.6       # outside the prior
    out = mg.run_batch(like, [obs], theta, names)
    c = mg.extract_observable(obs)
    c['template'] = 'turnover'
    c['kTO_fid'], c['pkTO_dd_fid'] = float(template.kTO_fid), float(template.pkTO_dd_fid)
    mg.save('turnover', names=np.array(names), theta=theta, obs0=c, precision=np.asarray(like.precision), covariance=cov,
            priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(mg.prior_spec, like.varied_params)]), **out)
    print(names, c['kTO_fid'], c['pkTO_dd_fid'], out['loglikelihood'][:4])


def boundary():
    """The reference-side binding's key set for this pipeline (integration/desilike_mi355x.py::extract_config): tests/golden/boundary_turnover.npz."""
    import make_boundary_fixture as mb
    like, obs, template, cov = build()
    mb.dump('turnover', like, size=16, seed=5)


if __name__ == '__main__':
    dump()
    if '--boundary' in sys.argv: boundary()
