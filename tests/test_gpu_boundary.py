"""The drop-in boundary, proven from the reference side.

``tests/golden/boundary_*.npz`` hold the flat ``dl_config`` key -> array sets that ``integration/desilike_mi355x.py::extract_config`` -- the binding a desilike
maintainer adds -- read off REAL initialised reference likelihoods in the build container (tests/golden/make_boundary_fixture.py), with
the reference's own ``vmap(likelihood, return_derived=True)`` outputs on a theta batch.  GPU: the context is created from those keys alone through ctypes (no
``desilike_amd`` host mirror, no oracle) and must reproduce the reference's log-likelihoods to the north star's 1e-10.  CPU: the same keys are what the host mirror
compiles for the same pipeline."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, 'integration'))

FIXTURES = ['cfg2_dense', 'two_tracers', 'eft_qisoqap', 'cfg4_xi', 'cfg4_pk', 'cfg4_resummed', 'cfg4_resummed_xi_binned', 'cfg4_flexible', 'cfg4_models', 'xi_binned', 'cfg4_pk_pcs', 'cfg4_xi_pcs2', 'kaiser_xi', 'tns', 'tns_eft', 'png', 'turnover', 'bands', 'png_velocity', 'cfg3', 'cfg3_taylor', 'cfg3_taylor_standard', 'cfg3_stacked', 'cfg3_stacked_ongrid', 'cfg3_stacked_lpt']     # cfg3_stacked*: the emulator layout the reference ships (emulators/conversion.py:44-98) under its REPT tracer; cfg4_*: BASELINE configs[3] (damped BAO); *_xi: the reference's own get_corr as a folded operator; tns*: the reference's one-loop TNS theory (tests/golden/make_tns_fixture.py)
MARG_FIXTURES = ['cfg4_xi_marg', 'two_tracers_marg', 'cfg3_marg', 'two_tracers_mixed', 'cfg3_taylor_marg', 'cfg3_stacked_marg', 'cfg3_stacked_lpt_marg', 'cfg3_stacked_bench']                                     # analytically solved parameters ('.marg'); cfg3*: BASELINE configs[2], the reference's
# velocileptors tracer on a real EmulatedCalculator node (MLP 6 -> 4 x 64 -> 7296 / Taylor engines) built by the reference's own Emulator.to_calculator


def load_fixture(name):
    g = np.load(os.path.join(HERE, 'golden', 'boundary_{}.npz'.format(name)))
    cfg = {key[4:]: g[key] for key in g.files if key.startswith('cfg/')}
    return g, cfg


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_context_from_reference_side_keys(name):
    import torch  # noqa: F401  (loads the HIP runtime PyTorch bundles first: one runtime per process)
    from desilike_mi355x import Library
    g, cfg = load_fixture(name)
    library = Library(os.path.join(ROOT, 'desilike_amd', 'lib', 'libdesilike_amd.so'))
    ctx = library.create(cfg, device=0)
    loglike, logprior, status = library.eval_batch(ctx, g['theta'])
    ref_ll, ref_lp = g['loglikelihood'], g['logprior']
    inside = np.isfinite(ref_lp)
    assert (~inside).sum() >= 1 and np.array_equal(status[~inside], np.ones((~inside).sum(), dtype='i4')) and np.isneginf(logprior[~inside]).all()
    assert (status[inside] == 0).all()
    tol = 1e-10 * np.maximum(1., np.abs(ref_ll))
    assert (np.abs(loglike - ref_ll)[inside] <= tol[inside]).all(), np.abs(loglike - ref_ll)[inside].max()
    assert np.allclose(logprior[inside], ref_lp[inside], rtol=1e-13, atol=1e-13)
    library.lib.dl_destroy(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize('name', MARG_FIXTURES)
def test_marginalised_context_from_reference_side_keys(name):
    """Contexts with analytically solved parameters, created from the reference-side keys through ctypes.  The reference's own ``_solve`` needs jax (absent here): the
    fixture holds the exact quadratic form (c, g, H) of the reference's NON-marginalised log-posterior in the solved parameters x (the posterior IS a quadratic polynomial
    of them: values at x0, x0 +- s_i e_i, x0 + s_i e_i + s_j e_j with steps as large as the curvature allows determine it to the rounding of the reference's own values:
    tests/golden/make_boundary_fixture.py::exact_quadratic), from which  x* = x0 - H^-1 g,  logposterior = c - g H^-1 g / 2 - logdet(-H) / 2  (likelihoods/base.py:385-404:
    no 2 pi) follow exactly.  Tolerance: the north star's 1e-10."""
    import torch  # noqa: F401
    from desilike_mi355x import Library
    g, cfg = load_fixture(name)
    library = Library(os.path.join(ROOT, 'desilike_amd', 'lib', 'libdesilike_amd.so'))
    ctx = library.create(cfg, device=0)
    ns = len(g['solved'])
    n = len(g['marg_c'])
    theta = g['theta'][:n]
    loglike, logprior, status, solved = library.eval_batch(ctx, theta, n_solved=ns)
    inside = status == 0
    assert inside.sum() >= n - 1
    for i in np.flatnonzero(inside):
        c, grad, H = g['marg_c'][i], g['marg_g'][i], g['marg_H'][i]
        dx = -np.linalg.solve(H, grad)
        marg = np.asarray(cfg['marg.kind']).astype(bool)           # '.marg' parameters contribute their determinant, '.best' ones do not (likelihoods/base.py:394-404)
        ref = c + 0.5 * grad.dot(dx) - (0.5 * np.linalg.slogdet(-H[np.ix_(marg, marg)])[1] if marg.any() else 0.)
        got = loglike[i] + logprior[i]
        assert abs(got - ref) <= 1e-10 * max(1., abs(ref)), (i, got, ref)
        assert np.allclose(solved[i], g['marg_x0'] + dx, rtol=1e-7, atol=1e-9 * np.abs(g['marg_x0'] + dx).max())
    library.lib.dl_destroy(ctx)


@pytest.mark.parametrize('name', FIXTURES + MARG_FIXTURES)
def test_reference_side_keys_are_the_keys_of_the_header(name):
    """Every extracted key is one the header documents, the store accepts them, and without a GPU dl_create refuses (no CPU fallback)."""
    import re
    import torch
    from desilike_mi355x import Library
    g, cfg = load_fixture(name)
    header = open(os.path.join(ROOT, 'include', 'desilike_amd.h')).read()
    for key in cfg:
        tail = re.sub(r'^obs\d+\.', '', key)
        needle = tail if not tail.startswith('in.') else 'in.'
        assert needle.split('.')[-1] in header or needle in header, key
    if not torch.cuda.is_available():
        library = Library(os.path.join(ROOT, 'desilike_amd', 'lib', 'libdesilike_amd.so'))
        with pytest.raises(RuntimeError, match='no HIP device'):
            library.create(cfg, device=0)


def test_host_mirror_compiles_the_same_keys():
    """The same pipeline written with the host mirror (import swap) compiles to the key set the reference-side extraction produced."""
    from desilike_amd._lib import fill_config
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g, cfg = load_fixture('two_tracers')
    template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    observables = []
    for iobs, (tracer, kmax, shotnoise) in enumerate([('LRG', 0.2, 1e4), ('ELG', 0.15, 4e3)]):
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        nk = int(round(kmax / 0.005))
        observables.append(TracerPowerSpectrumMultipolesObservable(data=cfg['obs{:d}.flatdata'.format(iobs)], kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4),
                                                                   wmatrix={'resolution': 4}, theory=theory, shotnoise=shotnoise))
    rng = np.random.RandomState(2)
    A = rng.standard_normal((210, 210)) * 30.
    like = ObservablesGaussianLikelihood(observables=observables, covariance=A.dot(A.T) + 1e4 * np.eye(210))
    like.initialize()
    mirror = {}
    fill_config(like._spec({}, like._flatdata_list(), like.precision), lambda key, a: mirror.__setitem__(key, a), lambda key, a: mirror.__setitem__(key, a))
    names, rnames = like.varied_params.names(), [str(n) for n in g['names']]
    assert sorted(names) == sorted(rnames)
    for key, value in cfg.items():
        assert key in mirror, key
        if '.in.' in key:   # (theta column, default): columns may be ordered differently, the parameter they name must be the same
            col, rcol = int(mirror[key][0]), int(value[0])
            assert (col < 0) == (rcol < 0) and (col < 0 or names[col] == rnames[rcol]), key
        elif key == 'priors':
            for iname, name in enumerate(rnames):
                assert np.array_equal(mirror[key].reshape(-1, 5)[names.index(name)], value[iname]), name
        else:
            assert np.allclose(np.ravel(mirror[key]), np.ravel(value), rtol=1e-9 if key == 'precision' else 1e-13, atol=1e-14 if key == 'precision' else 1e-300), key


def test_host_mirror_compiles_the_same_keys_for_the_emulated_node():
    """BASELINE configs[2]: the keys ``extract_config`` read off the reference's LPT velocileptors tracer on a real ``EmulatedCalculator`` node are the keys the host mirror
    compiles for the same pipeline written with its own classes and the same engine state."""
    from desilike_amd._lib import fill_config
    from desilike_amd.emulators import EmulatedCalculator, MLPEmulatorEngine
    from desilike_amd.theories.galaxy_clustering import LPTVelocileptorsTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    from emulator_utils import CFG3_PARAMS, CFG3_SPECS, cfg3_full_kpt, cfg3_full_engines
    for name, solved in [('cfg3', []), ('cfg3_marg', ['alpha0p', 'alpha2p', 'alpha4p', 'sn0p', 'sn2p'])]:
        g, cfg = load_fixture(name)
        engines = {key: MLPEmulatorEngine(xlimits=e['xlimits'], layers=e['layers'], activation='silu', ylimits=e['ylimits'], yshape=e['yshape']) for key, e in cfg3_full_engines().items()}
        pt = EmulatedCalculator(CFG3_PARAMS, engines, k=cfg3_full_kpt(), ells=(0, 2, 4), z=0.8, param_specs=CFG3_SPECS)
        theory = LPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='LRG')
        for pname in solved: theory.init.params[pname].update(derived='.marg')
        theory.init.params['sn4p'].update(fixed=True, value=0.3)
        obs = TracerPowerSpectrumMultipolesObservable(data=cfg['obs0.flatdata'], kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=8e3)
        rng = np.random.RandomState(9)
        A = rng.standard_normal((120, 120)) * 40.
        like = ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + 4e4 * np.eye(120))
        like.initialize()
        mirror = {}
        fill_config(like._spec({}, like._flatdata_list(), like.precision), lambda key, a: mirror.__setitem__(key, a), lambda key, a: mirror.__setitem__(key, a))
        names, rnames = like.varied_params.names(), [str(n) for n in g['names']]
        assert sorted(names) == sorted(rnames) and like.solved_params.names() == solved == [str(n) for n in g['solved']] if solved else sorted(names) == sorted(rnames)
        for key, value in cfg.items():
            assert key in mirror, key
            if '.in.' in key:
                for (col, const), (rcol, rconst) in zip(mirror[key].reshape(-1, 2), value.reshape(-1, 2)):
                    assert (col < 0) == (rcol < 0) and (names[int(col)] == rnames[int(rcol)] if col >= 0 else const == rconst), key
            elif key == 'priors':
                for iname, pname in enumerate(rnames):
                    assert np.array_equal(mirror[key].reshape(-1, 5)[names.index(pname)], value[iname]), pname
            else:
                scale = np.abs(np.ravel(value)).max() if np.ravel(value).dtype.kind == 'f' else 0.
                assert np.allclose(np.ravel(mirror[key]), np.ravel(value), rtol=1e-9 if key == 'precision' else 1e-12, atol=1e-13 * scale), key


@pytest.mark.gpu
def test_metropolis_hastings_through_the_binding():
    """The reference-side ctypes binding of the sampler ABI (``dl_mh_*`` with host record arrays): a context created from the keys of a real desilike likelihood, chains
    equal to the oracle's restatement of the reference's MHSampler + BlockProposer with the same counter-based draws."""
    import torch  # noqa: F401
    from desilike_mi355x import Library, MetropolisHastings
    from oracle import np_oracle as orc
    g, cfg = load_fixture(FIXTURES[0])
    library = Library(os.path.join(ROOT, 'desilike_amd', 'lib', 'libdesilike_amd.so'))
    ctx = library.create(cfg, device=0)
    inside = np.isfinite(g['logprior'])
    theta = g['theta'][inside]
    ndim = theta.shape[1]
    sigma = 0.05 * theta.std(axis=0)
    covariance = np.diag(sigma**2)
    blocks, over, order, seed = [ndim - 2, 2], [1, 2], np.arange(ndim)[::-1].copy(), 77
    start = theta[:2] * 1.
    sampler = MetropolisHastings(library, ctx, covariance, nchains=2, vectorize=3, blocks=blocks, oversample_factors=over, order=order, seed=seed)
    chains = sampler.sample(start, iterations=40, thin_by=1)
    more = sampler.sample(None, iterations=20, thin_by=1)

    def log_prob_fn(x_sorted):
        x = np.empty_like(np.atleast_2d(x_sorted)); x[:, order] = np.atleast_2d(x_sorted)
        loglike, logprior, status = library.eval_batch(ctx, x)
        return np.where(status == 0, loglike + logprior, -np.inf)

    transforms = orc.mh_transforms(covariance[np.ix_(order, order)], blocks)
    for c in range(2):
        draws = orc.MHPhiloxDraws(seed, c, blocks, over)
        chain, weight, logp, final = orc.mh_sample(log_prob_fn, start[c][order], draws, transforms, ntries=60, vectorize=3)
        got = np.concatenate([chains[c][0], more[c][0]])[:, order]
        assert len(weight) > 5 and np.array_equal(np.concatenate([chains[c][1], more[c][1]]), weight)
        assert np.allclose(got, chain, rtol=1e-11, atol=1e-13)
        assert np.allclose(np.concatenate([chains[c][2], more[c][2]]), logp, rtol=1e-10, atol=1e-9)
    sampler.close()
    library.lib.dl_destroy(ctx)
