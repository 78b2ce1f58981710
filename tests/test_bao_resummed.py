"""Resummed BAO wiggles (bao.py:165-266, 670-717, 1051-1096) against a fixture from the reference (tests/golden/make_golden.py cfg4_resummed): reciso P_ell with shot
noise, recsym xi_ell with the 'fog-damping_move-all' smooth part, pre-reconstruction 'move-all'.  CPU: oracle (damping scales and wiggle multipoles); GPU: call surface."""
import os

import numpy as np
import pytest

from oracle import np_oracle as orc

TAGS = ['a', 'b', 'c']


def load():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cfg4_bao_resummed.npz'), allow_pickle=False)


@pytest.mark.parametrize('tag', TAGS)
def test_resummed_oracle_vs_reference(tag):
    g = load()
    names = [str(n) for n in g[tag + '_names']]
    mode, model = str(g[tag + '_mode']), str(g[tag + '_model'])
    kin, k11, pk_dd, pknow_dd = (g[tag + '_' + name] for name in ['kin', 'k11', 'pk_dd_fid', 'pknow_dd_fid'])
    scales = orc.bao_resummation_scales(k11, pknow_dd, float(g['rs_drag']), mode=mode)
    ref_scales = g[tag + '_sigmas2']
    assert np.allclose(scales, ref_scales, rtol=1e-12, atol=0.)
    for i, row in enumerate(g[tag + '_theta']):
        p = dict(zip(names, row))
        f = p.get('dbeta', 1.) * float(g['f_fid'])
        power = orc.bao_resummed_power(kin, g['mu'], g['wmu_ell'], k11, pk_dd, pknow_dd, f, scales, shotnoise=float(g[tag + '_shotnoise']), qpar=p['qpar'],
                                       qper=p['qper'], b1=p['b1'], sigmas=p.get('sigmas', 0.), d=p.get('d', 1.), mode=mode, model=model)
        ref = g[tag + '_wiggle_power'][i]
        assert np.allclose(power, ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
def test_resummed_call_surface_vs_reference(tag):
    from desilike_amd import vmap
    from desilike_amd.theories.galaxy_clustering import BAOPowerSpectrumTemplate, ResummedBAOWigglesTracerCorrelationFunctionMultipoles, ResummedBAOWigglesTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable, TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load()
    space, model, mode = str(g[tag + '_space']), str(g[tag + '_model']), str(g[tag + '_mode'])
    template = BAOPowerSpectrumTemplate(z=0.5)
    if space == 'xi':
        theory = ResummedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode=mode, model=model)
        obs = TracerCorrelationFunctionMultipolesObservable(data=g[tag + '_flatdata'], s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
    else:
        theory = ResummedBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode=mode, model=model)
        theory.init.params['d'].update(fixed=False)
        obs = TracerPowerSpectrumMultipolesObservable(data=g[tag + '_flatdata'], kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=3e3)
    for param in theory.init.params.select(basename='al*'):
        param.update(fixed=True)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g[tag + '_covariance'])
    names = [str(n) for n in g[tag + '_names']]
    assert like.varied_params.names() == names
    theory.initialize()
    assert np.allclose([theory.sigma_dd2, theory.sigma_nl2, theory.sigma_x2, theory.sigma_sn2], g[tag + '_sigmas2'], rtol=1e-12, atol=0.) and theory.shotnoise == float(g[tag + '_shotnoise'])
    theta = g[tag + '_theta']
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: theta[:, i] for i, name in enumerate(names)})
    assert errors == {}
    ref = g[tag + '_loglikelihood']
    assert (np.abs(derived['loglikelihood'] - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all()
    assert np.allclose(derived['logprior'], g[tag + '_logprior'], rtol=1e-13, atol=1e-13)
    power = like._get_context().eval_theory_host(theta, iobs=0)
    assert np.allclose(power, g[tag + '_wiggle_power'], rtol=1e-10, atol=1e-11 * np.abs(g[tag + '_wiggle_power']).max())
