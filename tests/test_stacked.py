"""SURVEY 8 row a12 as the reference ships it -- the jaxeffort layout (emulators/conversion.py:44-98): engines '11' / 'loop' / 'ct' / 'st', one network per (z, ell) stacked in
each, amplitude rescale by the input logA, redshift selection / blend of full_shape.py:1416-1443.  CPU part (``-m "not gpu"``):

* the oracle's restatement (``stacked_mlp_predict``, ``jaxeffort_pktable``) against what the REFERENCE computed on a state dictionary with that layout
  (tests/golden/boundary_cfg3_stacked.npz: pktable, P_ell, windowed theory, log-likelihood of its REPT tracer; tests/golden/make_boundary_fixture.py::stacked_fixtures);
* the device's per-point arithmetic for the keys the reference-side binding extracted (csrc/dl_fullshape.h::dl_emu_point_stacked + csrc/dl_host.hpp, run on the CPU by
  tests/csrc/emulate.cpp) against the reference's log-likelihoods;
* the host mirror (``desilike_amd.emulators.EmulatedCalculator.from_state`` + the REPT tracer) compiles the same keys as the binding read off the reference's objects.
The engines themselves are third-party in the reference (cosmoprimo.emulators.tools, absent: tests/golden/refstub stands in): parity of the engine UNPINNED, stated in DESIGN.md.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import np_oracle as orc
from emulator_utils import STK_PARAMS, STK_SPECS, STK_COMPONENTS, stacked_networks

HERE = os.path.dirname(os.path.abspath(__file__))
ZGRID = np.array([0.3, 0.51, 0.71, 0.92])


def load_fixture(name):
    g = np.load(os.path.join(HERE, 'golden', 'boundary_{}.npz'.format(name)))
    return g, {key[4:]: g[key] for key in g.files if key.startswith('cfg/')}


def oracle_engines(networks, params=STK_PARAMS):
    """The synthetic component networks (tests/emulator_utils.py) in the stacked form of emulators/conversion.py:58-98: per engine dict(xlimits, layers, activation, ylimits, power)."""
    engines = {}
    for (name, nm), power in zip(STK_COMPONENTS, [1, 2, 1, 0]):
        rows = networks[name]
        first = rows[0][0]
        nk = len(first['k_grid'])
        layers = [(np.array([[n['layers'][i][0] for n in row] for row in rows]), np.array([[n['layers'][i][1] for n in row] for row in rows])) for i in range(len(first['layers']))]
        xlimits = np.array(rows[-1][-1]['in_MinMax'], dtype='f8')
        xlimits[list(params).index('h')] /= 100.                                                   # conversion.py:73-74
        ylimits = np.array([[np.asarray(n['out_MinMax']).reshape(nm, nk, 2) for n in row] for row in rows])
        engines[name] = dict(xlimits=xlimits, layers=layers, activation=first['activations'][0], ylimits=ylimits, power=power)
    return engines


def oracle_pktable(engines, X, zgrid, z):
    components = [orc.stacked_mlp_predict(X, STK_PARAMS, e['xlimits'], e['layers'], e['activation'], e['ylimits'], amplitude_power=e['power']) for e in engines.values()]
    return orc.jaxeffort_pktable(components, zgrid=zgrid, z=z)


def test_oracle_against_the_reference():
    g, cfg = load_fixture('cfg3_stacked')
    names = [str(n) for n in g['names']]
    engines = oracle_engines(stacked_networks(ZGRID))
    precision = np.linalg.inv(g['ref/covariance'])
    flatdata = cfg['obs0.flatdata']
    for i in range(6):
        p = dict(zip(names, g['theta'][i]))
        if not np.isfinite(g['logprior'][i]): continue
        pktable = oracle_pktable(engines, p, ZGRID, [0.6])
        assert pktable.shape == g['ref/pktable'][i].shape
        assert np.allclose(pktable, g['ref/pktable'][i], rtol=1e-12, atol=1e-12 * np.abs(g['ref/pktable'][i]).max())
        params = {name: p.get(name, 0.) for name in ['b1', 'b2', 'bs', 'b3', 'alpha0', 'alpha2', 'alpha4', 'alpha6', 'sn0', 'sn2', 'sn4']}
        pars = orc.velocileptors_pars(params, 1., 1., basis='standard', model='rept')
        power = orc.interp1d(g['ref/k'], g['ref/kpt'], orc.tablevel_combine_bias_terms_poles(pktable[..., 0], pars, nd=1e-4).T).T
        assert np.allclose(power, g['ref/power'][i], rtol=1e-11, atol=1e-11 * np.abs(g['ref/power'][i]).max())
        flat = orc.window_apply(power, matrix_full=g['ref/window'], shotnoisein=g['ref/shotnoisein'], shotnoiseout=g['ref/shotnoiseout'])
        assert np.allclose(flat, g['ref/flattheory'][i], rtol=1e-11, atol=1e-11 * np.abs(g['ref/flattheory'][i]).max())
        loglike = orc.gaussian_loglikelihood(flat, flatdata, precision)[0]
        assert abs(loglike - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))


def test_oracle_redshift_selection():
    """``jaxeffort_pktable``: on the emulated grid no blend; on one emulated redshift the neighbour gets weight zero; between two, the reference's weights."""
    rng = np.random.RandomState(0)
    components = [rng.standard_normal((4, 3, nm, 5)) for name, nm in STK_COMPONENTS]
    full = orc.jaxeffort_pktable(components)
    assert full.shape == (3, 5, 19, 4)
    assert np.array_equal(orc.jaxeffort_pktable(components, zgrid=ZGRID, z=ZGRID), full)
    assert np.allclose(orc.jaxeffort_pktable(components, zgrid=ZGRID, z=[0.51])[..., 0], full[..., 1], rtol=0., atol=0.)
    blend = orc.jaxeffort_pktable(components, zgrid=ZGRID, z=[0.6])[..., 0]
    assert np.allclose(blend, full[..., 1] * (1. - 0.09) + full[..., 2] * 0.09, rtol=1e-15)
    with pytest.raises(ValueError): orc.jaxeffort_pktable(components, zgrid=ZGRID, z=[1.2])


class FlatEmulation(object):
    """tests/csrc/emulate.cpp on a flat key -> array set (the reference-side binding's output): the device's host-side folding + per-point phases on the CPU."""

    def __init__(self, cfg):
        from emulation import load_emulation
        self.lib = load_emulation()
        self.cfg = self.lib.emu_config_new()
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        self.keep = []
        for key, value in cfg.items():
            value = np.ascontiguousarray(value)
            if value.dtype.kind in 'iub':
                value = np.ascontiguousarray(value.ravel(), dtype=np.int32)
                self.lib.emu_config_set_i32(self.cfg, key.encode(), value.ctypes.data_as(ip), value.size)
            else:
                value = np.ascontiguousarray(value.ravel(), dtype=np.float64)
                self.lib.emu_config_set_f64(self.cfg, key.encode(), value.ctypes.data_as(dp), value.size)
            self.keep.append(value)
        self.n = sum(cfg['obs{:d}.flatdata'.format(i)].size for i in range(int(cfg['n_obs'][0])))

    def eval_batch(self, theta):
        theta = np.ascontiguousarray(theta, dtype='f8')
        loglike, flat = np.empty(len(theta)), np.empty((len(theta), self.n))
        dp = ctypes.POINTER(ctypes.c_double)
        if self.lib.emu_eval_batch(self.cfg, theta.ctypes.data_as(dp), len(theta), loglike.ctypes.data_as(dp), flat.ctypes.data_as(dp)):
            raise RuntimeError(self.lib.emu_last_error().decode())
        return loglike, flat


@pytest.mark.parametrize('name', ['cfg3_stacked', 'cfg3_stacked_ongrid', 'cfg3_stacked_lpt'])
def test_device_arithmetic_on_the_cpu_against_the_reference(name):
    g, cfg = load_fixture(name)
    inside = np.isfinite(g['logprior'])
    loglike, flat = FlatEmulation(cfg).eval_batch(g['theta'][inside])
    ref = g['loglikelihood'][inside]
    assert (np.abs(loglike - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), np.abs(loglike - ref).max()
    if 'ref/flattheory' in g.files:
        rows = np.flatnonzero(inside)[:4]
        rows = rows[rows < 6]
        for j, i in enumerate(rows): assert np.allclose(flat[j], g['ref/flattheory'][i], rtol=1e-11, atol=1e-11 * np.abs(g['ref/flattheory'][i]).max())


from bench_configs import stacked_state as _stacked_state   # noqa: E402  (shared with bench.py)


def state_like_the_converter(networks, z, params=STK_PARAMS, ells=(0, 2, 4)):
    """The state dictionary of emulators/conversion.py:44-98 as plain data (what ``np.load('emulator.npy', allow_pickle=True)[()]`` returns), from the synthetic networks --
    the same layout tests/golden/make_boundary_fixture.py::jaxeffort_layout_pt hands to the reference's ``Emulator.from_state``."""
    return _stacked_state(networks, z, params, ells=ells)


def mirror_likelihood(z=0.6, marg=False, hidden=(32, 32), activation='tanh', seed=3, nk=12, flatdata=None, covariance=None, zgrid=ZGRID, networks_nk=30):
    """The pipeline of fixture ``cfg3_stacked`` written with the host mirror (import swap)."""
    from desilike_amd.emulators import EmulatedCalculator
    from desilike_amd.theories.galaxy_clustering import REPTVelocileptorsTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    pt = EmulatedCalculator.from_state(state_like_the_converter(stacked_networks(zgrid, hidden=hidden, activation=activation, seed=seed, nk=networks_nk), zgrid), param_specs=STK_SPECS)
    theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, z=z, prior_basis='standard')
    for name in ['b3', 'alpha6', 'sn4']: theory.init.params[name].update(fixed=True)
    for name in ['alpha0', 'alpha2', 'alpha4']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=20.))
    for name in ['sn0', 'sn2']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=2.))
    if marg:
        for name in ['alpha0', 'alpha2', 'alpha4', 'sn0', 'sn2']: theory.init.params[name].update(derived='.marg')
    obs = TracerPowerSpectrumMultipolesObservable(data=flatdata, kedges=np.linspace(0.02, 0.2, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory, shotnoise=8e3)
    return ObservablesGaussianLikelihood(observables=[obs], covariance=covariance), pt, theory


def test_host_mirror_compiles_the_keys_of_the_binding():
    from desilike_amd._lib import fill_config
    g, cfg = load_fixture('cfg3_stacked')
    like, pt, theory = mirror_likelihood(flatdata=cfg['obs0.flatdata'], covariance=g['ref/covariance'])
    like.initialize()
    mirror = {}
    fill_config(like._spec({}, like._flatdata_list(), like.precision), lambda key, a: mirror.__setitem__(key, a), lambda key, a: mirror.__setitem__(key, a))
    assert sorted(like.varied_params.names()) == sorted(str(n) for n in g['names'])
    for key in ['obs0.emu0.type', 'obs0.emu0.widths', 'obs0.emu0.act', 'obs0.emu0.groups', 'obs0.mono_mode']:
        assert np.array_equal(np.ravel(mirror[key]), np.ravel(cfg[key])), key
    for key in ['obs0.emu0.xlimits', 'obs0.emu0.weights', 'obs0.emu0.scale', 'obs0.vconst']:
        assert np.allclose(np.ravel(mirror[key]), np.ravel(cfg[key]), rtol=1e-14, atol=0.), key
    # the folded operator: the binding probed the reference's operations, the mirror multiplies the factors it holds
    wm, wr = np.ravel(mirror['obs0.wmatrix']), np.ravel(cfg['obs0.wmatrix'])
    assert wm.shape == wr.shape and np.allclose(wm, wr, rtol=1e-11, atol=1e-13 * np.abs(wr).max())


def test_from_state_reads_the_converter_layout():
    from desilike_amd.emulators import EmulatedCalculator
    pt = EmulatedCalculator.from_state(state_like_the_converter(stacked_networks(ZGRID), ZGRID), param_specs=STK_SPECS)
    assert pt.stacked and pt.param_names == STK_PARAMS and tuple(pt.ells) == (0, 2, 4) and np.array_equal(pt.z, ZGRID)
    assert pt.engines['11'].amplitude == ('logA', 1e-10, 1) and pt.engines['loop'].amplitude == ('logA', 1e-10, 2) and pt.engines['st'].amplitude is None
    assert pt.engines['loop'].yshape == (4, 3, 9, 30) and pt.engines['ct'].activation == 'tanh' and pt.engines['11'].hidden == [32, 32]
    with pytest.raises(ValueError, match='outside of the range'):
        mirror_likelihood(z=1.2, flatdata=np.zeros(36), covariance=np.eye(36))[0].initialize()


@pytest.mark.parametrize('case', ['monomial twice', 'scale size', 'width', 'weights size', 'group range', 'scalar stack'])
def test_malformed_descriptions_are_refused_with_a_message(case):
    """The host-side parser of ``emu0.type = 2`` (csrc/dl_host.hpp::dl_build_emulated_obs, the code ``dl_create`` runs) names what is wrong instead of evaluating garbage."""
    g, cfg = load_fixture('cfg3_stacked')
    cfg = {key: np.array(value) for key, value in cfg.items()}
    groups = cfg['obs0.emu0.groups'].reshape(-1, 4)
    expect = None
    if case == 'monomial twice':
        groups[1, 2] = groups[0, 2]; expect = 'belongs to one group'
    elif case == 'scale size':
        cfg['obs0.emu0.scale'] = cfg['obs0.emu0.scale'][:-1]; expect = 'scale f64'
    elif case == 'width':
        cfg['obs0.emu0.widths'][1] = 200; expect = 'layer widths'
    elif case == 'weights size':
        cfg['obs0.emu0.weights'] = cfg['obs0.emu0.weights'][:-3]; expect = 'weights size'
    elif case == 'group range':
        groups[0, 1] = 10000; expect = 'weights size'     # (one past the last network sets the number of networks the weights must hold)
    elif case == 'scalar stack':
        for key in list(cfg):
            if key.startswith('obs0.emu0.'): cfg[key.replace('emu0', 'emu1')] = cfg[key]
        expect = 'only the table engine'
    cfg['obs0.emu0.groups'] = groups.ravel()
    theta = g['theta'][np.isfinite(g['logprior'])][:2]
    with pytest.raises(RuntimeError) as info: FlatEmulation(cfg).eval_batch(theta)
    assert expect in str(info.value), str(info.value)


def test_benchmarked_shape_against_the_reference():
    """the stacked configuration bench.py and tests/test_gpu_stacked.py run (``bench_configs.make_cfg3_stacked``: 7 redshifts, 5 x 64 tanh, 60 wavenumbers,
    window 120 x 1200, 5 solved parameters) against ``boundary_cfg3_stacked_bench.npz`` -- the key set the binding read off the REFERENCE's pipeline on the same synthetic
    weights (its ``Emulator.from_state`` + REPT tracer) and the exact quadratic of the reference's own log-posterior at 12 points: (i) the host mirror compiles those keys
    (window, grids, precision, data vector and folded operator are the reference's, not only the mirror's own), (ii) the oracle chain reproduces the reference's marginalised
    log-posterior to 1e-10."""
    from desilike_amd._lib import fill_config
    from bench_configs import make_cfg3_stacked, cfg3_stacked_oracle_solution
    g, cfg = load_fixture('cfg3_stacked_bench')
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=True, data=cfg['obs0.flatdata'])    # (the data vector: the reference's -- the mirror evaluates its own on the device: tests/test_gpu_stacked.py)
    like.initialize()
    mirror = {}
    fill_config(like._spec({}, like._flatdata_list(), like.precision), lambda key, a: mirror.__setitem__(key, a), lambda key, a: mirror.__setitem__(key, a))
    names = like.varied_params.names()
    assert names == [str(n) for n in g['names']] and solved == [str(n) for n in g['solved']]
    # (the binding also hands over sigma8 / fsigma8 constants, which the standard prior basis never reads)
    assert int(np.ravel(cfg['obs0.mono_mode'])[0]) >= 3 and set(cfg) - set(mirror) <= {'obs0.emu1.const', 'obs0.emu2.const'} and not set(mirror) - set(cfg), set(mirror) ^ set(cfg)
    for key in mirror:
        a, b = np.asarray(mirror[key], dtype='f8').reshape(np.asarray(cfg[key]).shape), np.asarray(cfg[key], dtype='f8')
        if key.startswith('obs0.in.'):     # input maps (column of theta, constant): the constant counts where there is no column
            assert np.array_equal(a[:, 0], b[:, 0]) and np.array_equal(a[a[:, 0] < 0, 1], b[b[:, 0] < 0, 1]), key
            continue
        scale = np.abs(b[np.isfinite(b)]).max() if b.size else 1.
        assert np.allclose(a, b, rtol=1e-11, atol=1e-12 * scale), (key, np.abs(a - b).max(), scale)
    marg = np.asarray(cfg['marg.kind']).astype(bool)
    for i in (0, 5, 11):
        c, grad, H = g['marg_c'][i], g['marg_g'][i], g['marg_H'][i]
        ref = c - 0.5 * grad.dot(np.linalg.solve(H, grad)) - 0.5 * np.linalg.slogdet(-H[np.ix_(marg, marg)])[1]
        sol = cfg3_stacked_oracle_solution(like, pt, theory, solved, g['theta'][i])
        prior = sum(like.all_params[name].prior(value) for name, value in zip(names, g['theta'][i]))
        got = sol['loglikelihood'] + sol['logprior_solved'] + prior      # (likelihoods/base.py:385-404: the best-fit prior term of the solved parameters sits in the log-prior)
        assert abs(got - ref) <= 1e-10 * max(1., abs(ref)), (i, got, ref)
        assert np.allclose(sol['x'], g['marg_x0'] - np.linalg.solve(H, grad), rtol=1e-7, atol=1e-9)
