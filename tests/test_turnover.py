"""The turn-over template (reference power_template.py:1293-1340): oracle and host mirror on the CPU, the device path on the GPU, against outputs of the reference's own
template under a Kaiser tracer (tests/golden/make_turnover_fixture.py)."""
import numpy as np
import pytest

from golden_utils import load_golden, observable_constants
from oracle import np_oracle as orc


def params_of(names, row):
    p = dict(zip(names, row))
    p['b1'] = (p['b1'], p['b1'])
    p['qpar'], p['qper'] = orc.ap_qparqper('qap', 1. / 3., qap=p['qap'])
    return p


def test_oracle_against_the_reference():
    g = load_golden('turnover')
    c, names = observable_constants(g), [str(n) for n in g['names']]
    assert c['template'] == 'turnover'
    for i in range(8):
        out = orc.fullshape_observable(c, params_of(names, g['theta'][i]))
        assert np.allclose(out['pk_dd_template'], g['int_pk_dd_template'][i, 0], rtol=1e-13)
        assert np.allclose(out['power'], g['int_power'][i, 0], rtol=1e-11, atol=1e-8)
    for i in range(len(g['theta'])):
        if not np.isfinite(g['logprior'][i]): continue
        flat = orc.fullshape_observable(c, params_of(names, g['theta'][i]))['flattheory']
        ll = orc.gaussian_loglikelihood(flat, c['flatdata'], g['precision'])[0]
        assert abs(ll - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))


def test_host_mirror_parameters_and_turn_over():
    from desilike_amd.theories.galaxy_clustering import TurnOverPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, find_turn_over
    g = load_golden('turnover')
    # the vertex of the parabola through the three highest points: an exact parabola in log-log gives its own maximum back
    k = np.geomspace(1e-3, 1., 200)
    assert np.isclose(find_turn_over(k, 10.**(3. - 0.7 * (np.log10(k) + 1.8)**2)), 10.**-1.8, rtol=1e-12)
    template = TurnOverPowerSpectrumTemplate(z=0.8, fiducial='synthetic', kTO_fid=float(g['obs0']['kTO_fid']), pkTO_dd_fid=float(g['obs0']['pkTO_dd_fid']))
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    theory.initialize()
    assert template.apmode == 'qap' and [p.name for p in template.params if p.varied] == ['m', 'n', 'qto']     # power_template.yaml:424-477: dpto, qap, df fixed
    spec = theory._theory_spec()
    assert int(spec['template'][0]) == 2 and int(spec['apmode'][0]) == 2 and np.isclose(spec['kto_fid'][0], g['obs0']['kTO_fid'])
    imap = theory._input_map()
    assert all(imap[name] == name for name in ['m', 'n', 'qto', 'dpto'])
    # found on the mirror's own fiducial when not given: a maximum inside the table, the power read off the spectrum
    own = TurnOverPowerSpectrumTemplate(z=0.8, fiducial='synthetic')
    own.initialize()
    assert 1e-3 < own.kTO_fid < 0.1 and np.isclose(own.pkTO_dd_fid, own.fiducial.pk_dd(np.array([own.kTO_fid]))[0])


@pytest.mark.gpu
def test_device_against_the_reference():
    from golden_utils import spec_from_golden
    from desilike_amd._lib import Context
    g = load_golden('turnover')
    names = [str(n) for n in g['names']]
    spec = spec_from_golden(g)
    obs = spec['observables'][0]
    obs['template'], obs['apmode'] = np.array([2]), np.array([2])
    obs['kto_fid'], obs['pkto_fid'] = [float(g['obs0']['kTO_fid'])], [float(g['obs0']['pkTO_dd_fid'])]
    for name, default in [('m', 0.6), ('n', 0.9), ('qto', 1.), ('dpto', 1.), ('qap', 1.)]:
        obs['inputs'][name] = (names.index(name), default)
    ctx = Context(spec, device=0)
    loglike, logprior, status, flat = ctx.eval_batch_host(g['theta'], return_flattheory=True)
    ok = np.isfinite(g['logprior'])
    assert (~ok).sum() == 1 and (status[~ok] == 1).all() and (status[ok] == 0).all()
    assert (np.abs(loglike - g['loglikelihood'])[ok] <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood'][ok]))).all(), np.abs(loglike - g['loglikelihood'])[ok].max()
    assert np.allclose(flat[ok], g['flattheory'][ok], rtol=1e-11, atol=1e-8)
    # the host mirror compiles the same pipeline
    from desilike_amd import vmap
    from desilike_amd.theories.galaxy_clustering import TurnOverPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    from desilike_amd.fiducial import TabulatedFiducial
    c = g['obs0']
    template = TurnOverPowerSpectrumTemplate(z=0.8, fiducial=TabulatedFiducial(c['k11'], c['pk_dd_fid'], float(c['f_fid'])), kTO_fid=float(c['kTO_fid']), pkTO_dd_fid=float(c['pkTO_dd_fid']))
    for name in ['dpto', 'qap', 'df']: template.init.params[name].update(fixed=False)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data=c['flatdata'], kedges=np.linspace(0.001, 0.101, 41), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    assert sorted(like.varied_params.names()) == sorted(names)
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert (np.abs(derived[like._param_loglikelihood] - g['loglikelihood'])[ok] <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood'][ok]))).all()
