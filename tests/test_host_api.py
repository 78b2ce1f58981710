"""CPU: the host-side mirror of the reference's calculator API builds, from the reference's own constructor arguments,
the same constants the reference's initialised calculators hold (golden fixtures), and keeps its parameter conventions."""
import numpy as np
import pytest

from golden_utils import load_golden, spec_from_golden
from bench_configs import make_cfg2, make_cfg4, make_cfg5   # noqa: E402,F401  (shared with bench.py / tools)


@pytest.mark.parametrize('dense', [False, True])
def test_spec_matches_reference_constants(dense):
    g, like = make_cfg2(dense=dense)
    assert like.varied_params.names() == [str(n) for n in g['names']]
    spec = like._spec({}, like._flatdata_list(), like.precision)
    ref = spec_from_golden(g)
    assert np.allclose(spec['priors'], ref['priors'], rtol=0, atol=0)
    assert np.allclose(spec['precision'], ref['precision'], rtol=1e-9, atol=1e-14)
    o, r = spec['observables'][0], ref['observables'][0]
    for key in ['kin', 'mu', 'wmu_ell', 'k_t', 'pk_dd_fid', 'f_fid', 'a', 'kp', 'nd', 'wmatrix', 'shotnoise_in', 'shotnoise_out', 'ells_in', 'template', 'theory', 'apmode']:
        assert np.allclose(np.ravel(o[key]), np.ravel(r[key]), rtol=1e-13, atol=1e-300), key
    for name, (col, const) in r['inputs'].items():
        assert o['inputs'][name][0] == col, name


def test_parameter_conventions():
    from desilike_amd import Parameter, ParameterPrior, ParameterCollection
    prior = ParameterPrior(dist='norm', loc=1., scale=2., limits=(-1., 4.))
    assert prior(1.) == 0. and np.isneginf(prior(5.)) and np.isclose(prior(3.), -0.5)      # zero-lag removed, parameter.py:2003-2007
    assert prior(4.) > -np.inf                                                              # closed limits
    uniform = ParameterPrior(limits=(0., 1.))
    assert uniform(0.5) == 0. and np.isneginf(uniform(1.5))
    assert not ParameterPrior().is_proper()
    param = Parameter('LRG.b1', prior=dict(limits=[0., 4.]), ref=dict(limits=[1., 2.]))
    assert param.name == 'LRG.b1' and param.basename == 'b1' and param.namespace == 'LRG' and param.varied and param.value == 1.5
    assert Parameter('sigmapar', value=0.).fixed                                            # no prior, no ref => fixed (parameter.py:789-790)
    solved = Parameter('sn0', prior=dict(dist='norm', loc=0., scale=10.), derived='.marg')
    assert solved.solved and solved.varied
    with pytest.raises(Exception):
        Parameter('x', prior=dict(limits=[0., 1.]), derived='.marg')                      # limited prior cannot be marginalised (parameter.py:769-771)
    params = ParameterCollection({'a': dict(prior=dict(limits=[0., 1.])), 'b': dict(value=2., fixed=True)})
    assert params.names(varied=True) == ['a'] and params.prior(a=0.5) == 0. and np.isneginf(params.prior(a=2.))


def test_multitracer_namespaces():
    # full_shape.py:88-128
    from desilike_amd.theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles
    assert KaiserTracerPowerSpectrumMultipoles().params.names() == ['b1', 'sn0', 'sigmapar', 'sigmaper']
    assert KaiserTracerPowerSpectrumMultipoles(tracers='LRG').params.names() == ['LRG.b1', 'LRG.sn0', 'sigmapar', 'sigmaper']
    cross = KaiserTracerPowerSpectrumMultipoles(tracers=['LRG', 'ELG'])
    assert cross.params.names() == ['LRG.b1', 'ELG.b1', 'LRGxELG.sn0', 'sigmapar', 'sigmaper']
    cross.initialize()
    assert cross._bias_names() == {'b1X': 'LRG.b1', 'b1Y': 'ELG.b1', 'sn0': 'LRGxELG.sn0'}
    eft = EFTLikeKaiserTracerPowerSpectrumMultipoles(ells=(0, 2))
    eft.initialize()
    assert eft.counterterm_params == ['ct0_2', 'ct2_2'] and 'ct4_2' not in eft.params     # terms of absent multipoles are dropped (full_shape.py:598-599)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU / library: no silent CPU evaluation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from desilike_amd import LibraryError
    g, like = make_cfg2()
    with pytest.raises(LibraryError):
        like(b1=2.)


def test_two_tracer_spec_matches_reference():
    g, like = make_cfg5()
    # same parameters as the reference (its pipeline happens to order the tracer blocks differently: only the set is compared)
    assert sorted(like.varied_params.names()) == sorted(str(n) for n in g['names'])
    spec = like._spec({}, like._flatdata_list(), like.precision)
    ref = spec_from_golden(g)
    assert np.allclose(spec['precision'], ref['precision'], rtol=1e-8, atol=1e-14)
    names, rnames = like.varied_params.names(), [str(n) for n in g['names']]
    for o, r in zip(spec['observables'], ref['observables']):
        for key in ['kin', 'mu', 'wmu_ell', 'k_t', 'pk_dd_fid', 'nd', 'wmatrix', 'shotnoise_in', 'shotnoise_out']:
            assert np.allclose(np.ravel(o[key]), np.ravel(r[key]), rtol=1e-13, atol=1e-300), key
        for name, (col, const) in r['inputs'].items():
            if col >= 0: assert names[o['inputs'][name][0]] == rnames[col], name


@pytest.mark.parametrize('space', [pytest.param('xi', marks=pytest.mark.gpu), 'pk'])   # 'xi': the theory builds its Hankel operator with the device FFTLog
def test_bao_spec_matches_reference_constants(space):
    from golden_utils import spec_from_golden_bao
    g, like = make_cfg4(space)
    assert sorted(like.varied_params.names()) == sorted(str(n) for n in g['names'])
    spec = like._spec({}, like._flatdata_list(), like.precision)
    ref = spec_from_golden_bao(g)
    o, r = spec['observables'][0], ref['observables'][0]
    for key in ['kin', 'mu', 'wmu_ell', 'k_t', 'pk_dd_fid', 'pknow_dd_fid', 'f_fid', 'bao_mode', 'smoothing_radius', 'theory', 'template']:
        assert np.allclose(np.ravel(o[key]), np.ravel(r[key]), rtol=1e-13, atol=1e-300), key
    # broadband columns may be ordered differently: compare the window matrix column by parameter name
    names, rnames = like.varied_params.names(), [str(n) for n in g['names']]
    n_in = len(o['kin']) * len(o['ells_in'])
    # 'xi': device-built Hankel operator vs host-built; entries far below the largest one carry the FFT's rounding noise
    assert np.allclose(o['wmatrix'][:, :n_in], r['wmatrix'][:, :n_in], rtol=1e-12 if space == 'pk' else 1e-9, atol=(1e-14 if space == 'pk' else 1e-13) * np.abs(r['wmatrix']).max())
    pcols, rpcols = o['inputs']['pass'][0], r['inputs']['pass'][0]
    for ic, col in enumerate(pcols):
        jc = [rnames[c] for c in rpcols].index(names[col])
        assert np.allclose(o['wmatrix'][:, n_in + ic], r['wmatrix'][:, n_in + jc], rtol=1e-12, atol=0)


def make_cfg3(engine='taylor', data=None, marg=False):
    """REPT velocileptors tracer on an emulated PT node = the stand-in node of fixture cfg3_velocileptors_table as an exact Taylor emulator."""
    from desilike_amd.emulators import EmulatedCalculator, TaylorEmulatorEngine
    from desilike_amd.theories.galaxy_clustering import REPTVelocileptorsTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    from emulator_utils import taylor_state, EMU_PARAMS
    g = load_golden('cfg3_velocileptors_table')
    c = g['obs0']
    engines = {name: TaylorEmulatorEngine(**state) for name, state in taylor_state(g).items()}
    specs = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
             'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}
    pt = EmulatedCalculator(EMU_PARAMS, engines, k=c['kpt'], ells=(0, 2, 4), z=0.8, param_specs=specs)
    theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='LRG')
    if marg:
        for name in ['alpha0p', 'alpha2p', 'alpha4p', 'sn0p', 'sn2p']:
            theory.init.params[name].update(derived='.marg')
    obs = TracerPowerSpectrumMultipolesObservable(data=c['flatdata'] if data is None else data, kedges=np.linspace(0.02, 0.2, 37), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory, shotnoise=8e3)
    return g, ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])


def test_emulated_velocileptors_spec():
    g, like = make_cfg3()
    assert like.varied_params.names() == [str(n) for n in g['names']]
    assert np.allclose([p.prior.spec() for p in like.varied_params], g['priors'])
    spec = like._spec({}, like._flatdata_list(), like.precision)
    obs = spec['observables'][0]
    assert obs['wmatrix'].shape == (108, 6 * 19) and int(obs['mono_mode'][0]) == 2
    theory = like.observables[0].wmatrix.theory
    assert np.allclose(theory.k, g['obs0']['k'], rtol=1e-14) and np.isclose(theory.sigv, g['obs0']['sigv']) and np.isclose(theory.snd, g['obs0']['snd'])


def test_velocileptors_freedom_presets():
    """full_shape.py:1100-1117: 'max' / 'min' presets of the velocileptors tracers, in the standard and the physical prior basis."""
    from desilike_amd.theories.galaxy_clustering import LPTVelocileptorsTracerPowerSpectrumMultipoles as LPT
    std = LPT._default_params(prior_basis='standard', freedom='max')
    assert std['alpha6']['fixed'] and std['alpha6']['value'] == 0. and std['b2']['prior'] == dict(limits=[-15., 15.]) and std['alpha0']['prior'] is None and not std['b3'].get('fixed', False)
    std = LPT._default_params(prior_basis='standard', freedom='min')
    assert all(std[name]['fixed'] for name in ['b3', 'bs', 'alpha6']) and std['b2']['prior'] == dict(dist='norm', loc=0., scale=10.) and std['sn2']['prior'] is None
    phys = LPT._default_params(prior_basis='physical', freedom='min')
    assert all(phys[name]['fixed'] for name in ['b3p', 'bsp', 'alpha6p']) and phys['alpha0p']['prior'] == dict(dist='norm', loc=0., scale=12.5)
    with pytest.raises(ValueError):
        LPT._default_params(freedom='medium')


def test_default_fiducial_warns():
    """the reference's default fiducial 'DESI' needs cosmoprimo; the synthetic stand-in must not be silent, 'synthetic' is the explicit opt-in."""
    import warnings
    from desilike_amd.fiducial import get_fiducial, FiducialWarning, SyntheticFiducial
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate
    with warnings.catch_warnings():
        warnings.simplefilter('error', FiducialWarning)
        assert isinstance(get_fiducial('synthetic'), SyntheticFiducial)
        ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic').initialize()
        for default in (None, 'DESI'):
            with pytest.raises(FiducialWarning):
                get_fiducial(default)
        with pytest.raises(FiducialWarning):
            ShapeFitPowerSpectrumTemplate(z=0.5).initialize()
    with pytest.raises(ValueError):
        get_fiducial('planck')


def test_context_cache_eviction_never_closes_and_replicas_are_distinct(monkeypatch):
    """The cache of compiled contexts is bounded by dropping references, never by closing: a device-resident runner (one replica per chain / stream) may still hold an
    evicted context (more than MAX_CONTEXTS chains on one rank was a use-after-free).  ``SumLikelihood`` hands out one context per replica too."""
    from desilike_amd import _lib
    from desilike_amd.likelihoods import SumLikelihood

    class FakeContext(object):
        closed, created = [], 0

        def __init__(self, spec, device=0):
            FakeContext.created += 1
            self.spec = spec

        def close(self):
            FakeContext.closed.append(self)

    monkeypatch.setattr(_lib, 'Context', FakeContext)
    g, like = make_cfg2()
    like.initialize()
    monkeypatch.setattr(type(like), 'MAX_CONTEXTS', 4)
    held = [like._get_context(replica=i) for i in range(12)]          # what 12 chains of a chain-parallel sampler hold
    assert len(set(map(id, held))) == 12 and FakeContext.closed == []
    assert len(like._contexts) <= 5
    assert like._get_context(replica=11) is held[11]
    like._get_context(replica=3)                                      # an evicted replica is compiled again from the kept spec
    like._get_context({'df': 1.01})
    assert FakeContext.closed == []
    g2, like2 = make_cfg2()
    total = SumLikelihood(likelihoods=[like2])
    a, b = total._get_posterior_context(replica=0)[0], total._get_posterior_context(replica=1)[0]
    assert a is not b and total._get_context(replica=1) is b
