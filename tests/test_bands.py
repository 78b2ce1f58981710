"""The band template of the velocity-divergence power spectrum (reference power_template.py:868-970): oracle and host mirror on the CPU, the device path on the GPU, against
outputs of the reference's own template under a Kaiser tracer (tests/golden/make_bands_fixture.py)."""
import numpy as np
import pytest

from golden_utils import load_golden
from oracle import np_oracle as orc


def constants(g):
    c = dict(g['obs0'])
    c['template'] = 'bands'
    c['ellsin'], c['ells'] = tuple(int(ell) for ell in c['ellsin']), tuple(int(ell) for ell in c['ells'])
    return c


def params_of(names, row):
    p = dict(zip(names, row))
    p['b1'] = (p['b1'], p['b1'])
    p['qpar'], p['qper'] = orc.ap_qparqper('qap', 1. / 3., qap=p['qap'])
    p['dptt'] = [p['dptt{:d}'.format(i)] for i in range(5)]
    return p


def test_oracle_against_the_reference():
    g = load_golden('bands')
    c, names = constants(g), [str(n) for n in g['names']]
    assert np.allclose(orc.band_templates(c['k11'], c['band_kp']), c['band_templates'], rtol=1e-14, atol=1e-16)
    for i in range(8):
        out = orc.fullshape_observable(c, params_of(names, g['theta'][i]))
        assert np.allclose(out['pk_dd_template'], g['int_pk_dd_template'][i, 0], rtol=1e-13)
        assert np.allclose(out['power'], g['int_power'][i, 0], rtol=1e-11, atol=1e-8)
    for i in range(len(g['theta'])):
        if not np.isfinite(g['logprior'][i]): continue
        flat = orc.fullshape_observable(c, params_of(names, g['theta'][i]))['flattheory']
        ll = orc.gaussian_loglikelihood(flat, c['flatdata'], g['precision'])[0]
        assert abs(ll - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))


def make_mirror(g):
    from desilike_amd.theories.galaxy_clustering import BandVelocityPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    from desilike_amd.fiducial import TabulatedFiducial
    c = g['obs0']
    template = BandVelocityPowerSpectrumTemplate(z=0.8, kp=c['band_kp'], fiducial=TabulatedFiducial(c['k11'], c['pk_dd_fid'], float(c['f_fid'])), pk_tt_fid=c['pk_tt_fid'])
    template.init.params['df'].update(fixed=False)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data=c['flatdata'], kedges=np.linspace(0.01, 0.21, 41), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    return ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance']), theory, template


def test_host_mirror_against_the_reference_constants():
    g = load_golden('bands')
    like, theory, template = make_mirror(g)
    theory.initialize()
    assert template.apmode == 'qap' and np.allclose(template.templates, g['obs0']['band_templates'], rtol=1e-14, atol=1e-16)
    assert sorted(like.varied_params.names()) == sorted(str(n) for n in g['names'])
    spec = theory._theory_spec()
    assert int(spec['template'][0]) == 3 and spec['band_templates'].shape == (5, len(g['obs0']['k11']))
    assert theory._input_map()['band'] == ['dptt{:d}'.format(i) for i in range(5)]
    from desilike_amd.theories.galaxy_clustering import BandVelocityPowerSpectrumTemplate
    with pytest.raises(ValueError): BandVelocityPowerSpectrumTemplate(k=np.linspace(0.01, 0.2, 50), fiducial='synthetic').initialize()       # no band parameter
    auto = BandVelocityPowerSpectrumTemplate(k=np.linspace(0.01, 0.2, 50), nbands=4, fiducial='synthetic')
    auto.initialize()
    inside = (auto.k >= auto.kp[0]) & (auto.k <= auto.kp[-1])
    assert auto.kp.size == 4 and np.isclose(auto.kp[0], 0.01 + 0.19 / 8.) and np.allclose(auto.templates.sum(axis=0)[inside], 1.)        # the tents partition unity between the pivots


@pytest.mark.gpu
def test_device_against_the_reference():
    from desilike_amd import vmap
    g = load_golden('bands')
    names = [str(n) for n in g['names']]
    like, theory, template = make_mirror(g)
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    ok = np.isfinite(g['logprior'])
    assert (~ok).sum() == 1 and np.isneginf(logpost[~ok]).all()
    assert (np.abs(derived[like._param_loglikelihood] - g['loglikelihood'])[ok] <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood'][ok]))).all(), np.abs(derived[like._param_loglikelihood] - g['loglikelihood'])[ok].max()
    like(**{name: g['theta'][0, i] for i, name in enumerate(names)})
    assert np.allclose(like.flattheory, g['flattheory'][0], rtol=1e-11, atol=1e-8)
