"""Fiber collisions in configuration space (desilike_amd/observables/galaxy_clustering/correlation_function.py; reference window.py:1052-1250) against the reference's own
kernels and likelihood values (tests/golden/make_fc_xi_fixture.py): CPU -- the kernels; GPU -- the damped-BAO xi_ell likelihood with the kernels folded into its window."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TAGS = ['tophat', 'tophat_binned', 'tophat_cut', 'tophat_cut_binned', 'general', 'general_binned']


def load():
    return dict(np.load(os.path.join(HERE, 'golden', 'fc_xi.npz')))


def make_fiber(g, tag):
    from desilike_amd.observables.galaxy_clustering import TopHatFiberCollisionsCorrelationFunctionMultipoles, FiberCollisionsCorrelationFunctionMultipoles
    if tag.startswith('tophat_cut'): return TopHatFiberCollisionsCorrelationFunctionMultipoles(fs=0.6, Dfc=4., mu_range_cut=True, with_uncorrelated=False)
    if tag.startswith('tophat'): return TopHatFiberCollisionsCorrelationFunctionMultipoles(fs=0.6, Dfc=4.)
    return FiberCollisionsCorrelationFunctionMultipoles(sep=g['sep'], kernel=g['kernel'])


@pytest.mark.parametrize('tag', TAGS)
def test_kernels_against_the_reference(tag):
    g = load()

    class Grid(object):      # stands for the theory: the kernels need its separations and multipoles only
        s, ells = g[tag + '/sin'], (0, 2)
        init = {}
        def initialize(self): return self

    fiber = make_fiber(g, tag)
    fiber.init.update(theory=Grid(), ells=(0, 2))
    Grid.init = type('Init', (dict,), {'update': lambda self, **kw: None})()
    fiber.initialize()
    assert np.allclose(fiber.kernel_correlated, g[tag + '/kernel_correlated'], rtol=1e-12, atol=1e-15)
    assert np.allclose(fiber.kernel_uncorrelated, g[tag + '/kernel_uncorrelated'], rtol=1e-12, atol=1e-15)
    if tag == 'general':
        top = fiber.to_tophat()
        assert 0. < top.init['fs'] < 1. and 0. < top.init['Dfc'] < 6.


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
def test_likelihood_with_fiber_collisions_against_the_reference(tag):
    from desilike_amd import vmap
    from desilike_amd.theories.galaxy_clustering import BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load()
    theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=BAOPowerSpectrumTemplate(z=0.5), mode='reciso')
    kw = dict(sedges=np.linspace(20., 170., 31), wmatrix={'resolution': 2}) if tag.endswith('binned') else dict(s=np.linspace(22.5, 167.5, 30))
    obs = TracerCorrelationFunctionMultipolesObservable(data=g[tag + '/flatdata'], ells=(0, 2), theory=theory, fiber_collisions=make_fiber(g, tag), **kw)
    for name in ['sigmapar', 'sigmaper']:
        theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    names = [str(name) for name in g[tag + '/names']]
    assert sorted(like.varied_params.names()) == sorted(names)
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g[tag + '/theta'][:, i] for i, name in enumerate(names)})
    ref = g[tag + '/loglikelihood']
    ok = np.isfinite(g[tag + '/logprior'])
    assert ok.sum() >= 8
    # (the xi_ell fixtures rest on the repo's FFTLog at the Hankel step, like every xi fixture: 1e-9 as in tests/test_gpu_bao.py)
    assert (np.abs(derived[like._param_loglikelihood][ok] - ref[ok]) <= 1e-9 * np.maximum(1., np.abs(ref[ok]))).all(), np.abs(derived[like._param_loglikelihood][ok] - ref[ok]).max()
    like(**{name: g[tag + '/theta'][0, i] for i, name in enumerate(names)})
    assert np.allclose(like.flattheory, g[tag + '/flattheory'][0], rtol=1e-8, atol=1e-9 * np.abs(g[tag + '/flattheory'][0]).max())
