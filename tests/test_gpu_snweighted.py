"""GPU (-m gpu): the Fisher-forecast likelihood ``SNWeightedPowerSpectrumLikelihood`` (reference likelihoods/galaxy_clustering/fisher.py:10-71) against outputs of the
reference's own class (tests/golden/make_snweighted_fixture.py): fiducial P(k, mu), diagonal precision, log-likelihoods, and the Fisher matrix built on it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    from desilike_amd.theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles, ShapeFitPowerSpectrumTemplate
    from desilike_amd.likelihoods.galaxy_clustering import SNWeightedPowerSpectrumLikelihood
    from desilike_amd.observables.galaxy_clustering import BoxFootprint
    g = dict(np.load(os.path.join(HERE, 'golden', 'snweighted.npz')))
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=float(g['z']), fiducial='synthetic'))
    footprint = BoxFootprint(volume=2e9, nbar=5e-4)
    like = SNWeightedPowerSpectrumLikelihood(theories=[theory], data={'b1': 2., 'sn0': 0.1}, covariance={'b1': 1.9}, footprints=[footprint], klim=tuple(g['klim']), mu=int(g['mu']))
    return g, like


def test_against_the_reference():
    from desilike_amd import vmap
    g, like = build()
    like.initialize()
    assert like.flatdata.shape == g['flatdata'].shape == (500 * int(g['mu']),)
    assert np.allclose(like.flatdata, g['flatdata'], rtol=1e-11, atol=1e-8)
    assert np.allclose(like.precision, g['precision'], rtol=1e-10)
    names = [str(name) for name in g['names']]
    assert like.varied_params.names() == names
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    ref = g['loglikelihood']
    ok = np.isfinite(g['logprior'])
    assert ok.sum() == len(ref) - 1
    assert (np.abs(derived[like._param_loglikelihood][ok] - ref[ok]) <= 1e-10 * np.maximum(1., np.abs(ref[ok]))).all()
    assert np.allclose(derived[like._param_logprior][ok], g['logprior'][ok], rtol=1e-13, atol=1e-13)
    assert np.isneginf(logpost[~ok]).all()
    for i in range(3):
        like(**{name: g['theta'][i, j] for j, name in enumerate(names)})
        assert np.allclose(like.flattheory, g['flattheory'][i], rtol=1e-11, atol=1e-8)


def test_fisher_forecast_runs_on_it():
    """The forecast the class exists for: Fisher matrix at the fiducial point = J^T diag(precision) J with the Jacobian of P(k, mu) (fisher.py of the reference)."""
    from desilike_amd.fisher import Fisher
    g, like = build()
    estimator = Fisher(like)
    estimator(b1=2., sn0=0.1)
    names = like.varied_params.names()
    hessian = np.asarray(estimator.likelihood_fisher._hessian)       # of the likelihood term alone (the Gaussian priors are a separate term)
    # the same matrix from central differences of the flat theory through the call surface
    center = {param.name: param.value for param in like.varied_params}
    center.update(b1=2., sn0=0.1)
    jac = []
    for name in names:
        step = 1e-4 * max(abs(center[name]), 1.)
        up, dn = dict(center), dict(center)
        up[name] += step; dn[name] -= step
        like(**up); fu = np.array(like.flattheory)
        like(**dn); fd = np.array(like.flattheory)
        jac.append((fu - fd) / (2. * step))
    jac = np.array(jac)
    expected = (jac * like.precision).dot(jac.T)
    scale = np.sqrt(np.diag(expected))
    assert np.allclose(-hessian / scale[:, None] / scale[None, :], expected / scale[:, None] / scale[None, :], rtol=0., atol=2e-4)
    assert np.all(np.linalg.eigvalsh(-hessian) > 0.)
