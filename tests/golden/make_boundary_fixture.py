"""Prove the drop-in boundary from the REFERENCE side (build container only; the reference never travels to the GPU box):

    python tests/golden/make_boundary_fixture.py

Builds real desilike likelihoods with the reference's own classes (through tests/golden/refstub for the absent cosmoprimo / lsstypes), initialises them, runs
``integration/desilike_mi355x.py::extract_config`` -- the reference-side binding of INTEGRATION.md section 2 -- on them, and stores in ``boundary_<name>.npz``:
the flat ``dl_config`` key -> array set (``cfg/<key>``), a theta batch, and the reference's own ``vmap(likelihood, return_derived=True)`` outputs.
tests/test_gpu_boundary.py creates the device context from these keys alone (ctypes, no desilike_amd host mirror) and must reproduce the reference's numbers.
"""
import os
import sys
import warnings

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'integration'))
warnings.filterwarnings('ignore')

from desilike.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
from desilike.likelihoods import ObservablesGaussianLikelihood
from desilike.base import vmap

from desilike_mi355x import extract_config
from make_golden import dense_window, sample_theta


def covariance(n, seed):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal((n, n)) * 30.
    return A.dot(A.T) + 1e4 * np.eye(n)


def exact_quadratic(logpost, x0):
    """(c, g, H) with ``logpost(x0 + d) = c + g.d + d.H.d / 2`` EXACTLY: the reference's log-posterior is a quadratic polynomial of the parameters its theory is linear in, so
    the values at x0, x0 +- s_i e_i and x0 + s_i e_i + s_j e_j determine it with no truncation error, whatever the steps.  The steps are chosen for the ROUNDING: a first pass
    at unit-like steps estimates the curvatures, the second takes s_i = 2^n ~ sqrt(2 |c| / |H_ii|) -- the second difference is then as large as the values themselves, so that
    g and H carry the relative rounding of the reference's own values (~1e-15), not ~1e-9 as small-step finite differences of numbers of size |c| do.  Steps are powers of two:
    x0 +- s, s^2 and s_i s_j are exact; differences are taken in extended precision."""
    ld = np.longdouble
    ns = len(x0)
    f0 = ld(logpost(x0))
    if not np.isfinite(float(f0)):      # a point outside the prior of a sampled parameter: nothing to solve
        return float(f0), np.full(ns, np.nan), np.full((ns, ns), np.nan)

    def diagonal(steps):
        fp, fm = np.zeros(ns, dtype=ld), np.zeros(ns, dtype=ld)
        for i in range(ns):
            e = np.zeros(ns); e[i] = steps[i]
            fp[i], fm[i] = logpost(x0 + e), logpost(x0 - e)
        return fp, fm

    steps = np.ones(ns)
    for _ in range(2):
        fp, fm = diagonal(steps)
        assert np.isfinite(fp.astype('f8')).all() and np.isfinite(fm.astype('f8')).all(), 'step outside the prior of a solved parameter'
        curvature = np.abs(((fp - 2 * f0 + fm) / ld(steps)**2).astype('f8'))
        steps = 2.**np.round(np.log2(np.sqrt(2. * max(abs(float(f0)), 1.) / curvature)))
    fp, fm = diagonal(steps)
    s = steps.astype(ld)
    g = (fp - fm) / (2 * s)
    H = np.zeros((ns, ns), dtype=ld)
    H[np.diag_indices(ns)] = (fp - 2 * f0 + fm) / s**2
    for i in range(ns):
        for j in range(i + 1, ns):
            e = np.zeros(ns); e[i], e[j] = steps[i], steps[j]
            H[i, j] = H[j, i] = (ld(logpost(x0 + e)) - fp[i] - fp[j] + f0) / (s[i] * s[j])
    # the quadratic reproduces the reference at a point it was not built from
    rng = np.random.RandomState(0)
    d = (rng.uniform(-1., 1., ns) * steps)
    check = float(f0 + g.dot(d) + 0.5 * d.dot(H).dot(d))
    value = logpost(x0 + d)
    assert abs(check - value) <= 1e-12 * max(1., abs(value), abs(float(f0))), (check, value)
    return float(f0), g.astype('f8'), H.astype('f8')


SELECTED = set(sys.argv[1:])     # fixture names to (re)generate; none: all of them


def dump(name, likelihood, size=48, seed=42, unsolved=None, extras=None):
    """``likelihood`` / ``unsolved``: likelihoods, or callables that build them (so that only the selected fixtures cost anything).
    ``unsolved``: for likelihoods with analytically solved parameters (the reference's ``_solve`` needs jax: not runnable here) the same pipeline WITHOUT the
    '.marg' / '.best' flags -- the reference evaluates that one on a stencil of the solved parameters, which gives the exact quadratic form (value, gradient, Hessian)
    of its log-posterior in them; the closed forms of the marginalised / profiled posterior follow (the pin of tests/golden/make_golden.py::marg_multi)."""
    if SELECTED and name not in SELECTED: return
    if not hasattr(likelihood, 'all_params'): likelihood = likelihood()
    if unsolved is not None and not hasattr(unsolved, 'all_params'): unsolved = unsolved()
    try:
        likelihood()
    except (ModuleNotFoundError, AttributeError) as exc:   # analytic solve (likelihoods/base.py:130, 157: jax): the pipeline itself has been initialised and calculated by then
        assert 'jax' in str(exc) and unsolved is not None
    cfg = extract_config(likelihood)
    names = [str(n) for n in cfg['__varied__']]
    theta = sample_theta(likelihood, size, seed)
    theta[3, 0] = likelihood.varied_params[names[0]].prior.limits[1] + 0.01     # one row outside the prior
    out = {'cfg/' + key: value for key, value in cfg.items() if not key.startswith('__')}
    out.update(names=np.array(names), theta=theta)
    solved = [str(n) for n in cfg['__solved__']]
    if solved:
        out['solved'] = np.array(solved)
        unsolved()
        x0 = np.array([likelihood.all_params[n].value for n in solved])
        rows = []
        for row in theta[:12]:
            base = dict(zip(names, row))
            rows.append(exact_quadratic(lambda x: unsolved(**{**base, **dict(zip(solved, x))}), x0))
        out.update(marg_x0=x0, marg_c=np.array([r[0] for r in rows]), marg_g=np.array([r[1] for r in rows]), marg_H=np.array([r[2] for r in rows]))
        errors = {}
    else:
        (logpost, derived), errors = vmap(likelihood, backend=None, errors='return', return_derived=True)({n: theta[:, i] for i, n in enumerate(names)})
        out.update(loglikelihood=np.asarray(derived[likelihood._param_loglikelihood]), logprior=np.asarray(derived[likelihood._param_logprior]), logposterior=np.asarray(logpost))
    if extras is not None:      # what an oracle needs to follow the reference step by step (keys 'ref/...'): e.g. the window matrix, the theory vectors of the first rows
        out.update({'ref/' + key: value for key, value in extras(unsolved if solved else likelihood, names, theta).items()})
    fn = os.path.join(here, 'boundary_{}.npz'.format(name))
    np.savez_compressed(fn, **out)
    print('saved', fn, '{:.1f} kB'.format(os.path.getsize(fn) / 1e3), 'keys', len(cfg) - 1, 'errors', len(errors))


def emulated_pt(cls, engines, params, specs, k, ells=(0, 2, 4), z=0.8):
    """A REAL ``EmulatedCalculator`` of the reference (emulators/__init__.py:394-418) built by the reference's own ``Emulator.to_calculator`` (150-208) from a fitted
    state: the engines are the third-party part (cosmoprimo.emulators.tools, absent: tests/golden/refstub stands in with the attribute surface the reference's code
    fixes -- ``center / powers / derivatives`` (emulators/__init__.py:471-507), ``model_operations`` / ``xoperations`` / ``yoperations`` with ``_locals``
    (emulators/conversion.py:20-96)); they are DATA here: weights and limits from tests/emulator_utils.py, nothing is trained."""
    from desilike.emulators import Emulator, Operation, MLPEmulatorEngine, TaylorEmulatorEngine
    from desilike.io import BaseConfig
    from desilike.parameter import ParameterCollection, Parameter
    from desilike.utils import serialize_class

    def mlp(e):
        engine = MLPEmulatorEngine.__new__(MLPEmulatorEngine)
        engine.model_operations = []
        for ilayer, (kernel, bias) in enumerate(e['layers']):                                   # the reference's own expression strings: emulators/conversion.py:25-34, 75-79
            engine.model_operations.append(Operation('(v[..., None, :] @ kernel)[..., 0, :] + bias', locals={'kernel': kernel, 'bias': bias}))
            if ilayer < len(e['layers']) - 1: engine.model_operations.append(Operation('v / (1 + jnp.exp(-v))', locals={}))
        yshape = tuple(e['yshape']) if tuple(e['yshape']) != (1,) else ()
        engine.xoperations = [Operation('(v - limits[..., 0]) / (limits[..., 1] - limits[..., 0])', locals={'limits': np.asarray(e['xlimits'])})]
        engine.yoperations = [Operation('((v - limits[..., 0]) / (limits[..., 1] - limits[..., 0]))', inverse='v * (limits[..., 1] - limits[..., 0]) + limits[..., 0]',
                                        locals={'limits': np.asarray(e['ylimits']).reshape(yshape + (2,))})]
        engine.params, engine.xshape, engine.yshape = list(params), (len(params),), yshape
        return engine

    def taylor(e):
        engine = TaylorEmulatorEngine.__new__(TaylorEmulatorEngine)
        engine.center, engine.powers, engine.derivatives = np.asarray(e['center'], dtype='f8'), np.asarray(e['powers'], dtype='i4'), np.asarray(e['derivatives'], dtype='f8')
        engine.xoperations, engine.yoperations = [], []
        engine.params, engine.xshape, engine.yshape = list(params), (len(params),), engine.derivatives.shape[1:]
        return engine

    emulator = Emulator.__new__(Emulator)
    emulator.engines = {name: (taylor(e) if 'derivatives' in e else mlp(e)) for name, e in engines.items()}
    emulator.xoperations, emulator.yoperations, emulator.defaults = [], [], {}
    emulator.fixed = {'k': np.asarray(k, dtype='f8'), 'ells': tuple(ells), 'z': np.array(z)}
    emulator.varied_params = list(params)
    emulator.in_calculator_state = ['pktable']
    emulator.is_calculator_sequence = False
    emulator.calculator__class__ = serialize_class(cls)
    emulator.yaml_data = BaseConfig({'class': cls.__name__, 'info': {}, 'params': {}})
    emulator.all_params = ParameterCollection([Parameter(name, **spec) for name, spec in specs.items()])
    return emulator.to_calculator()


def jaxeffort_layout_pt(cls, networks, z, params, specs, ells=(0, 2, 4), drop_z=False):
    """A REAL ``EmulatedCalculator`` of the reference from a state dictionary with the layout ``convert_jaxeffort_to_desilike`` writes (emulators/conversion.py:44-98), through
    the reference's own ``Emulator.from_state`` / ``__setstate__`` (emulators/__init__.py:218-238) and ``to_calculator`` (150-208).  ``networks``: what jaxeffort's
    component emulators hold (tests/emulator_utils.py::stacked_networks: synthetic weights).  Layout, field by field:

    * four engines '11', 'loop', 'ct', 'st' of kind 'mlp', ``yshape = (n_z, n_ell, n_m, n_k)``; ``model_operations``: the dense-layer / activation expressions of conversion.py:25-34
      with every ``kernel`` / ``bias`` stacked ``[n_z, n_ell, ...]`` (``merge_operations``, 58-66); ``xoperations``: ONE min-max scaler, the 'h' row divided by 100 (73-75);
      ``yoperations``: the min-max scaler with limits ``[n_z, n_ell, n_m, n_k, 2]`` (76-79) and, in front of it, the amplitude operation (88-92);
    * emulator-level ``yoperations``: the split / concatenate pair of conversion.py:50-51 (``pktable [n_ell, n_k, 19, n_z]``); ``fixed``: ells, k, z; ``in_calculator_state = ['pktable']``.

    ``drop_z``: the single-redshift variant an LPT node needs (its ``combine_bias_terms_poles`` takes ``pktable [n_ell, n_k, 19]``, full_shape.py:1207-1208): the emulator-level
    inverse additionally selects ``[..., 0]`` -- NOT something conversion.py writes, an adaptation of its layout, stated as such in DESIGN.md."""
    from desilike.emulators import Emulator, Operation
    from desilike.parameter import ParameterCollection, Parameter
    from desilike.utils import serialize_class
    activation_expressions = {'silu': 'v / (1 + jnp.exp(-v))', 'relu': 'jnp.maximum(v, 0.)', 'tanh': 'jnp.tanh(v)'}                  # conversion.py:27-34
    concat = "v['pktable'] = jnp.moveaxis(jnp.concatenate([v.pop('11'), v.pop('loop'), v.pop('ct'), v.pop('st')], axis=-2), [0, -1], [-1, 1])"
    if drop_z: concat += "[..., 0]"
    state = {'engines': {}, 'xoperations': [], 'defaults': {}, 'fixed': {},
             'yoperations': [Operation("v['11'], v['loop'], v['ct'], v['st'] = jnp.split(v.pop('pktable'), [3, 12, 16], axis=2); v", concat + "; v").__getstate__()],
             'varied_params': list(params), 'in_calculator_state': ['pktable'], 'calculator__class__': serialize_class(cls), 'is_calculator_sequence': False,
             'yaml_data': {'class': cls.__name__, 'info': {}, 'params': {}},
             'all_params': ParameterCollection([Parameter(name, **spec) for name, spec in specs.items()]).__getstate__()}

    def stack(function):
        return np.array([[function(networks_iz_ell) for networks_iz_ell in row] for row in component_networks])

    for component, component_networks in networks.items():
        first = component_networks[0][0]
        k, nlayers = first['k_grid'], len(first['layers'])
        model_operations = []
        for ilayer in range(nlayers):
            model_operations.append(Operation('(v[..., None, :] @ kernel)[..., 0, :] + bias', locals={'kernel': stack(lambda n: n['layers'][ilayer][0]), 'bias': stack(lambda n: n['layers'][ilayer][1])}))
            if ilayer < nlayers - 1: model_operations.append(Operation(activation_expressions[first['activations'][ilayer]], locals={}))
        limits = np.array(component_networks[-1][-1]['in_MinMax'], dtype='f8')                       # the scaler of the last network read (conversion.py:72-75)
        if 'h' in params: limits[list(params).index('h')] /= 100.
        xoperations = [Operation('(v - limits[..., 0]) / (limits[..., 1] - limits[..., 0])', locals={'limits': limits})]
        yoperations = [Operation('((v - limits[..., 0]) / (limits[..., 1] - limits[..., 0]))', inverse='v * (limits[..., 1] - limits[..., 0]) + limits[..., 0]',
                                 locals={'limits': stack(lambda n: np.asarray(n['out_MinMax']).reshape(-1, len(k), 2))})]
        if 'logA' in params:
            if component in ['11', 'ct']: yoperations.insert(0, Operation("v / (jnp.exp(X['logA']) * 1e-10)", inverse="v * jnp.exp(X['logA']) * 1e-10"))
            if component in ['loop']: yoperations.insert(0, Operation("v / (jnp.exp(X['logA']) * 1e-10)**2", inverse="v * (jnp.exp(X['logA']) * 1e-10)**2"))
        yshape = (len(z), len(ells), model_operations[-1]._locals['bias'].size // (len(z) * len(ells) * len(k)), len(k))
        state['engines'][component] = {'name': 'mlp', 'params': list(params), 'xshape': (len(params),), 'yshape': yshape, 'xoperations': [operation.__getstate__() for operation in xoperations],
                                       'yoperations': [operation.__getstate__() for operation in yoperations], 'model_operations': [operation.__getstate__() for operation in model_operations],
                                       'model_yoperations': []}
    state['fixed'].update(ells=list(ells), k=k, z=np.array(z))
    return Emulator.from_state(state).to_calculator()


def stacked_fixtures():
    """SURVEY 8 row a12 as the reference ships it: the jaxeffort layout (four engines x (z, ell) stacks, amplitude rescale by logA), the redshift selection / blend the REPT node
    inserts (full_shape.py:1416-1443), under the reference's REPT tracer (``z`` between two emulated redshifts, on one of them) and -- single redshift -- its LPT tracer."""
    sys.path.insert(0, os.path.join(root, 'tests'))
    from desilike.theories.galaxy_clustering.full_shape import (LPTVelocileptorsPowerSpectrumMultipoles, LPTVelocileptorsTracerPowerSpectrumMultipoles,
                                                                REPTVelocileptorsPowerSpectrumMultipoles, REPTVelocileptorsTracerPowerSpectrumMultipoles)
    from emulator_utils import STK_PARAMS, STK_SPECS, stacked_networks
    from make_golden import spd_covariance
    zgrid = np.array([0.3, 0.51, 0.71, 0.92])

    def rept(z, marg, hidden=(32, 32), activation='tanh', seed=3, nk=12):
        pt = jaxeffort_layout_pt(REPTVelocileptorsPowerSpectrumMultipoles, stacked_networks(zgrid, hidden=hidden, activation=activation, seed=seed), zgrid, STK_PARAMS, STK_SPECS)
        theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, z=z, prior_basis=None)                     # (as conversion.py:130 uses it)
        for name in ['b3', 'alpha6', 'sn4']: theory.init.params[name].update(fixed=True)
        for name in ['alpha0', 'alpha2', 'alpha4']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=20.))
        for name in ['sn0', 'sn2']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=2.))
        if marg:
            for name in ['alpha0', 'alpha2', 'alpha4', 'sn0', 'sn2']: theory.init.params[name].update(derived='.marg')
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.7, 'b2': 0.4, 'alpha0': 3.}, kedges=np.linspace(0.02, 0.2, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory,
                                                      shotnoise=8e3)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=spd_covariance(3 * nk, seed=11, diag=4e4, amp=40.))

    def extras(likelihood, names, theta):
        obs = likelihood.observables[0]
        wm, theory = obs.wmatrix, obs.wmatrix.theory
        flat, power, pktable = [], [], []
        for row in theta[:6]:
            if not np.isfinite(likelihood(**dict(zip(names, row)))): flat.append(np.full(obs.flatdata.size, np.nan)); power.append(np.full(theory.power.shape, np.nan)); pktable.append(np.full(theory.pt.pktable.shape, np.nan)); continue
            flat.append(np.array(obs.flattheory)); power.append(np.array(theory.power)); pktable.append(np.array(theory.pt.pktable))
        return dict(window=np.asarray(wm.matrix_full), k=np.asarray(theory.k), kpt=np.asarray(theory.pt.k), shotnoiseout=np.asarray(wm.shotnoiseout), shotnoisein=np.asarray(wm.shotnoisein),
                    flattheory=np.array(flat), power=np.array(power), pktable=np.array(pktable), covariance=np.linalg.inv(likelihood.precision))

    dump('cfg3_stacked', lambda: rept(0.6, False), size=24, seed=31, extras=extras)
    dump('cfg3_stacked_marg', lambda: rept(0.6, True), size=24, seed=31, unsolved=lambda: rept(0.6, False))
    dump('cfg3_stacked_ongrid', lambda: rept(0.51, False, hidden=(16, 24, 16), activation='silu', seed=5, nk=10), size=24, seed=33)

    # THE BENCHMARKED SHAPE: bench_configs.py::make_cfg3_stacked as bench.py and tests/test_gpu_stacked.py run it -- seven emulated redshifts, 5 x 64 tanh, 60
    # wavenumbers per table, tracer at z = 0.8 between two of them, standard prior basis, 40 bins x 3 multipoles, binning window at resolution 10 (120 x 1200), five solved parameters;
    # the same synthetic weights (stacked_networks(seed = 11)) and the same covariance, through the reference's own Emulator.from_state + REPT tracer
    def bench(marg):
        from bench_configs import STACKED_ZGRID
        pt = jaxeffort_layout_pt(REPTVelocileptorsPowerSpectrumMultipoles, stacked_networks(STACKED_ZGRID, hidden=(64, 64, 64, 64, 64), activation='tanh', seed=11, nk=60), STACKED_ZGRID, STK_PARAMS, STK_SPECS)
        theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, z=0.8, prior_basis='standard')
        for name in ['b3', 'alpha6', 'sn4']: theory.init.params[name].update(fixed=True)
        for name in ['alpha0', 'alpha2', 'alpha4']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=20.))
        for name in ['sn0', 'sn2']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=2.))
        if marg:
            for name in ['alpha0', 'alpha2', 'alpha4', 'sn0', 'sn2']: theory.init.params[name].update(derived='.marg')
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.7, 'b2': 0.4, 'alpha0': 3.}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=8e3)
        A = np.random.RandomState(5).standard_normal((120, 120)) * 40.
        return ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + 4e4 * np.eye(120))

    dump('cfg3_stacked_bench', lambda: bench(True), size=24, seed=37, unsolved=lambda: bench(False))

    def lpt(marg):
        pt = jaxeffort_layout_pt(LPTVelocileptorsPowerSpectrumMultipoles, stacked_networks(zgrid[:1], hidden=(16, 16), activation='silu', seed=7), zgrid[:1], STK_PARAMS, STK_SPECS, drop_z=True)
        theory = LPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, prior_basis='standard')
        for name in ['b3', 'alpha6', 'sn4']: theory.init.params[name].update(fixed=True)
        for name in ['alpha0', 'alpha2', 'sn0']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=20.))
        if marg:
            for name in ['alpha0', 'alpha2', 'sn0']: theory.init.params[name].update(derived='.marg')
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 0.7, 'b2': 0.4, 'alpha0': 3.}, kedges=np.linspace(0.02, 0.2, 11), ells=(0, 2), wmatrix={'resolution': 2}, theory=theory,
                                                      shotnoise=8e3)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=spd_covariance(20, seed=13, diag=4e4, amp=40.))

    dump('cfg3_stacked_lpt', lambda: lpt(False), size=24, seed=35)
    dump('cfg3_stacked_lpt_marg', lambda: lpt(True), size=24, seed=35, unsolved=lambda: lpt(False))


def emulated_fixtures():
    """BASELINE configs[2] through the reference-side binding: the reference's velocileptors tracer classes on top of an ``EmulatedCalculator`` node."""
    sys.path.insert(0, os.path.join(root, 'tests'))
    from desilike.theories.galaxy_clustering.full_shape import (LPTVelocileptorsPowerSpectrumMultipoles, LPTVelocileptorsTracerPowerSpectrumMultipoles,
                                                                REPTVelocileptorsPowerSpectrumMultipoles, REPTVelocileptorsTracerPowerSpectrumMultipoles)
    from emulator_utils import CFG3_PARAMS, CFG3_SPECS, cfg3_full_kpt, cfg3_full_engines, taylor_state, EMU_PARAMS
    from golden_utils import load_golden
    from make_golden import spd_covariance
    kedges = np.linspace(0., 0.2, 41)

    # the size SURVEY 8d states: MLP 6 -> 4 x 64 silu -> 3 * 128 * 19 outputs, n_kin = 400, binning window 120 x 1200; LPT, physical prior basis
    def cfg3(marg):
        pt = emulated_pt(LPTVelocileptorsPowerSpectrumMultipoles, cfg3_full_engines(), CFG3_PARAMS, CFG3_SPECS, cfg3_full_kpt())
        theory = LPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='LRG')
        if marg:
            for name in ['alpha0p', 'alpha2p', 'alpha4p', 'sn0p', 'sn2p']: theory.init.params[name].update(derived='.marg')
        theory.init.params['sn4p'].update(fixed=True, value=0.3)
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1p': 1.6, 'b2p': 0.3, 'alpha0p': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=8e3)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=spd_covariance(120, seed=9, diag=4e4, amp=40.))

    dump('cfg3', lambda: cfg3(False), size=24, seed=19)
    dump('cfg3_marg', lambda: cfg3(True), size=24, seed=19, unsolved=lambda: cfg3(False))

    # Taylor engines (exact second-order tables of fixture cfg3_velocileptors_table), REPT with its co-evolution shift, physical basis; and the standard basis
    def cfg3_taylor(prior_basis):
        g = load_golden('cfg3_velocileptors_table')
        specs = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
                 'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}
        pt = emulated_pt(REPTVelocileptorsPowerSpectrumMultipoles, taylor_state(g), EMU_PARAMS, specs, g['obs0']['kpt'])
        theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='ELG', prior_basis=prior_basis)
        data = {'b1p': 1.2, 'b2p': 0.4} if prior_basis == 'physical' else {'b1': 1.7, 'b2': 0.4, 'alpha0': 3.}
        obs = TracerPowerSpectrumMultipolesObservable(data=data, kedges=np.linspace(0.02, 0.2, 37), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory, shotnoise=8e3)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])

    def cfg3_taylor_marg(marg):
        like = cfg3_taylor('physical')
        if marg:
            theory = like.observables[0].wmatrix.theory
            for name in ['alpha0p', 'alpha2p', 'sn0p']: theory.init.params[name].update(derived='.marg')
        return like

    dump('cfg3_taylor_marg', lambda: cfg3_taylor_marg(True), size=24, seed=29, unsolved=lambda: cfg3_taylor_marg(False))
    dump('cfg3_taylor', lambda: cfg3_taylor('physical'), size=24, seed=29)
    dump('cfg3_taylor_standard', lambda: cfg3_taylor('standard'), size=24, seed=39)


def main():
    kedges = np.linspace(0., 0.2, 41)
    # (1) BASELINE configs[1]: ShapeFit + Kaiser, dense survey-like window
    kin, wmat = dense_window(kedges, (0, 2, 4))
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix=wmat, kin=kin, ellsin=(0, 2, 4), theory=theory, shotnoise=1e4)
    dump('cfg2_dense', ObservablesGaussianLikelihood(observables=[obs], covariance=covariance(120, 1)))
    # (2) BASELINE configs[4] geometry: two tracers (namespaced b1 / sn0, shared ShapeFit parameters), joint covariance, binning windows
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    observables = []
    for tracer, kmax, b1, shotnoise in [('LRG', 0.2, 2., 1e4), ('ELG', 0.15, 1.3, 4e3)]:
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        nk = int(round(kmax / 0.005))
        observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1}, kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 4},
                                                                   theory=theory, shotnoise=shotnoise))
    dump('two_tracers', ObservablesGaussianLikelihood(observables=observables, covariance=covariance(210, 2)))
    # (3) EFT-like Kaiser (counter / stochastic terms), qisoqap AP mode
    template = ShapeFitPowerSpectrumTemplate(z=0.5, apmode='qisoqap')
    theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.8, 'ct0_2': 1.5}, kedges=kedges, ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    dump('eft_qisoqap', ObservablesGaussianLikelihood(observables=[obs], covariance=covariance(120, 3)))
    # (4) BASELINE configs[3]: damped-BAO xi_ell (ell = 0, 2; 30 s-bins), 'power' broadband; and the P_ell version with a binning window
    from desilike.theories.galaxy_clustering import (BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles, DampedBAOWigglesTracerPowerSpectrumMultipoles,
                                                     KaiserTracerCorrelationFunctionMultipoles)
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable

    def bao_likelihood(space, marg=False):
        template = BAOPowerSpectrumTemplate(z=0.5)
        if space == 'xi':
            theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='reciso')
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        else:
            theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template)
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
            n, scale = 112, 30.
        for name in ['sigmapar', 'sigmaper']:
            theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
        if marg:
            for param in theory.init.params.select(basename='al*'):
                param.update(derived='.marg')
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        return ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (10. * scale)**2 * np.eye(n))

    # (4b) VERDICT r5 item 6: the wiggle models beyond 'standard' and the binned correlation-function window through the binding
    def bao_models(kind):
        from desilike.theories.galaxy_clustering import (ResummedBAOWigglesTracerPowerSpectrumMultipoles, ResummedBAOWigglesTracerCorrelationFunctionMultipoles,
                                                         FlexibleBAOWigglesTracerPowerSpectrumMultipoles, FlexibleBAOWigglesTracerCorrelationFunctionMultipoles)
        template = BAOPowerSpectrumTemplate(z=0.5)
        pk = dict(kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3})
        if kind == 'resummed':       # bao.py:165-266: reciso reconstruction, shot-noise damping, growth rescaling d varied
            theory = ResummedBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode='reciso', model='standard')
            theory.init.params['d'].update(fixed=False)
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, theory=theory, shotnoise=3e3, **pk)
            n, scale = 112, 30.
        elif kind == 'resummed_xi':  # bao.py:1051-1096: recsym, Beutler-like smooth part, every scale moved; binned separations (window.py:536-733)
            theory = ResummedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='recsym', model='fog-damping_move-all')
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, sedges=np.linspace(20., 170., 31), wmatrix={'resolution': 2}, ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        elif kind == 'flexible':     # bao.py:269-391: 'pcs' nodes multiplying the wiggles
            theory = FlexibleBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode='reciso', model='standard', wiggles='pcs')
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, theory=theory, **pk)
            n, scale = 112, 30.
        elif kind == 'models':       # bao.py:137-150: Beutler 2016 with every scale moved
            theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode='recsym', model='fog-damping_move-all')
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, theory=theory, **pk)
            for name in ['sigmapar', 'sigmaper']: theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
            n, scale = 112, 30.
        for param in theory.init.params.select(basename='al*'):
            if kind != 'resummed_xi': param.update(fixed=True)
        for param in theory.init.params.select(basename='ml*'):
            param.update(ref=dict(limits=[-0.3, 0.3]))
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        return ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (10. * scale)**2 * np.eye(n))

    def bao_kernels(space):      # kernel broadbands (bao.py:468-523, 833-905): 'pcs' nodes for P_ell; for xi_ell the same kernels Hankel-transformed + powers of s ('pcs2')
        template = BAOPowerSpectrumTemplate(z=0.5)
        if space == 'xi':
            theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='recsym', broadband='pcs2')
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        else:
            theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template, broadband='pcs')
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
            n, scale = 112, 30.
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        return ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (10. * scale)**2 * np.eye(n))

    dump('cfg4_pk_pcs', lambda: bao_kernels('pk'), size=24, seed=71)
    dump('cfg4_xi_pcs2', lambda: bao_kernels('xi'), size=24, seed=73)
    dump('cfg4_resummed', lambda: bao_models('resummed'), size=24, seed=61)
    dump('cfg4_resummed_xi_binned', lambda: bao_models('resummed_xi'), size=24, seed=63)
    dump('cfg4_flexible', lambda: bao_models('flexible'), size=24, seed=65)
    dump('cfg4_models', lambda: bao_models('models'), size=24, seed=67)
    def xi_binned():   # Kaiser xi_ell through the bin-integration window (window.py:536-733) with scale cuts that differ per multipole
        theory = KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
        obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, slim={0: (25., 165., 5.), 2: (40., 150., 5.)}, wmatrix={'resolution': 2}, theory=theory)
        n = 28 + 22
        rng = np.random.RandomState(24)
        A = rng.standard_normal((n, n)) * 3e-4
        return ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (3e-3)**2 * np.eye(n))

    dump('xi_binned', xi_binned, size=24, seed=69)
    dump('cfg4_xi', lambda: bao_likelihood('xi'), size=24, seed=9)
    dump('cfg4_pk', lambda: bao_likelihood('pk'), size=24, seed=9)
    # (5) the DESI-style BAO fit: every broadband term solved analytically
    dump('cfg4_xi_marg', lambda: bao_likelihood('xi', marg=True), size=24, seed=9, unsolved=lambda: bao_likelihood('xi'))
    # (6) Kaiser xi_ell (ShapeFit template)
    theory = KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory)
    rng = np.random.RandomState(14)
    A = rng.standard_normal((90, 90)) * 3e-4
    dump('kaiser_xi', ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (3e-3)**2 * np.eye(90)), size=24, seed=19)
    # (7) two tracers with both shot-noise terms marginalised (Gaussian priors)
    template = ShapeFitPowerSpectrumTemplate(z=0.5)

    def two_tracers(marg):
        observables = []
        for tracer, kmax, b1, shotnoise in [('LRG', 0.2, 2., 1e4), ('ELG', 0.15, 1.3, 4e3)]:
            theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
            theory.init.params[tracer + '.sn0'].update(prior=dict(dist='norm', loc=0.1, scale=2.), **({'derived': '.marg'} if marg else {}))
            nk = int(round(kmax / 0.005))
            observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1, tracer + '.sn0': 0.3}, kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4),
                                                                       wmatrix={'resolution': 4}, theory=theory, shotnoise=shotnoise))
        return ObservablesGaussianLikelihood(observables=observables, covariance=covariance(210, 22))

    dump('two_tracers_marg', lambda: two_tracers(True), size=24, seed=24, unsolved=lambda: two_tracers(False))

    def two_tracers_mixed():   # LRG.sn0 marginalised, ELG.sn0 at its best fit ('.best': likelihoods/base.py:336-404 -- no determinant for it)
        like = two_tracers(True)
        like.observables[1].wmatrix.theory.init.params['ELG.sn0'].update(derived='.best')
        return like

    dump('two_tracers_mixed', two_tracers_mixed, size=24, seed=24, unsolved=lambda: two_tracers(False))
    # (8) BASELINE configs[2]: emulated perturbation-theory node
    emulated_fixtures()
    # (9) the emulator layout the reference ships (jaxeffort conversion)
    stacked_fixtures()


if __name__ == '__main__':
    main()
