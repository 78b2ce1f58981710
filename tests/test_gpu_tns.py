"""GPU (-m gpu): the TNS one-loop theory on the device (csrc/dl_tns.hip), through the C ABI, against golden vectors of the reference itself
(tests/golden/make_tns_fixture.py: the reference's tns_kernels / tns_pt / TNSPowerSpectrumMultipoles / TNSTracerPowerSpectrumMultipoles, full_shape.py:688-971)
and against the NumPy oracle on seeded points.  Tolerances: 1e-10 on logL (relative above 1); the loop tables to 1e-10 of their largest entry."""
import numpy as np
import pytest

from test_oracle_tns import load, tns_oracle_point, FIXTURES
from bench_configs import spec_from_tns_golden   # noqa: E402,F401  (shared with bench.py / tools)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def contexts():
    from desilike_amd._lib import Context
    cache = {}

    def get(name):
        if name not in cache:
            g = load(name)
            cache[name] = (g, Context(spec_from_tns_golden(g), device=0))
        return cache[name]

    yield get
    for g, ctx in cache.values(): ctx.close()


def clean_rows(g, n):
    theta = g['theta'][:n].copy()
    for i in range(n):
        if not np.all(np.isfinite(theta[i])): theta[i] = g['theta'][0]
    return theta


@pytest.mark.parametrize('name', FIXTURES)
def test_loop_tables_vs_reference(contexts, name):
    g, ctx = contexts(name)
    nint = g['int_tables'].shape[0]
    tables = ctx.eval_tns_tables(clean_rows(g, nint), len(g['k11_table'])).cpu().numpy()
    for i in range(nint):
        for r in range(29):
            ref = g['int_tables'][i][r]
            assert np.max(np.abs(tables[i, r] - ref)) <= 1e-10 * np.max(np.abs(ref)), (i, r, np.max(np.abs(tables[i, r] - ref)) / np.max(np.abs(ref)))


@pytest.mark.parametrize('name', FIXTURES)
def test_power_vs_reference(contexts, name):
    g, ctx = contexts(name)
    nint = g['int_power'].shape[0]
    power = ctx.eval_theory_host(clean_rows(g, nint), iobs=0)
    ref = g['int_power']
    assert np.allclose(power, ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max()), np.abs(power - ref).max() / np.abs(ref).max()


@pytest.mark.parametrize('name', FIXTURES)
def test_loglikelihood_vs_reference(contexts, name):
    g, ctx = contexts(name)
    loglike, logprior, status, flat = ctx.eval_batch_host(g['theta'], return_flattheory=True)
    ok = np.isfinite(g['theta']).all(axis=1)
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood'][ok]))
    assert (np.abs(loglike[ok] - g['loglikelihood'][ok]) <= tol).all(), (np.abs(loglike[ok] - g['loglikelihood'][ok]) / np.maximum(1., np.abs(g['loglikelihood'][ok]))).max()
    finite = np.isfinite(g['logprior'])
    assert np.allclose(logprior[finite], g['logprior'][finite], rtol=1e-13, atol=1e-13)
    assert np.array_equal(status == 1, np.isneginf(g['logprior']) & ok)


def test_vs_oracle_ragged_batch(contexts):
    """67 seeded points (three 32-point tiles, the last one ragged) against the oracle."""
    from oracle import np_oracle as oc
    g, ctx = contexts('tns')
    names = [str(n) for n in g['names']]
    rng = np.random.RandomState(7)
    lo = dict(qpar=0.95, qper=0.95, dm=-0.05, df=0.8, sigmav=0., b1=1., b2=-1., bs=-1., b3=-1., sn0=-0.5)
    hi = dict(qpar=1.05, qper=1.05, dm=0.05, df=1.2, sigmav=6., b1=3., b2=1., bs=1., b3=1., sn0=0.5)
    theta = np.column_stack([rng.uniform(lo[n], hi[n], 67) for n in names])
    loglike, logprior, status = ctx.eval_batch_host(theta)
    q = g['c.k11']
    kernels = oc.tns_kernels(g['k11_table'], q, oc.weights_trapz(q))
    idx = list(range(0, 67, 6)) + [64, 65, 66]
    ref = np.array([tns_oracle_point(g, theta[i], kernels=kernels) for i in idx])
    assert (np.abs(loglike[idx] - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), (np.abs(loglike[idx] - ref) / np.maximum(1., np.abs(ref))).max()
    assert (status == 0).all()


def test_host_mirror_matches_oracle():
    """The mirror of the reference's classes (desilike_amd TNSTracerPowerSpectrumMultipoles + observable + likelihood) compiles to the same pipeline: its
    log-likelihood at seeded points equals the oracle's on the mirror's own constants."""
    from oracle import np_oracle as oc
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, TNSTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = TNSTracerPowerSpectrumMultipoles(template=template, fog='gaussian', freedom='max')
    kedges = np.linspace(0., 0.2, 21)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'b2': 0.3, 'sigmav': 2.}, kedges=kedges, ells=(0, 2), wmatrix={'resolution': 2}, theory=theory, shotnoise=1e4)
    rng = np.random.RandomState(5)
    A = rng.standard_normal((40, 40)) * 50.
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + 1e5 * np.eye(40))
    like.initialize()
    names = like.varied_params.names()
    assert set(['b1', 'b2', 'bs', 'b3', 'sn0', 'sigmav']) <= set(names)
    theta = np.column_stack([like.varied_params[n].ref.sample(size=5, random_state=rng) if like.varied_params[n].ref.is_proper() else np.full(5, like.varied_params[n].value) for n in names])
    if 'sigmav' in names: theta[:, names.index('sigmav')] = rng.uniform(0.5, 4., 5)
    loglike = np.array([like(**dict(zip(names, row))) - like.logprior for row in theta])
    wm = obs.wmatrix
    q = template.k
    k11 = oc.tns_k11(theory.k)
    kernels = oc.tns_kernels(k11, q, oc.weights_trapz(q))
    for i, row in enumerate(theta):
        p = dict(zip(names, row))
        pk_q = template.pk_dd_fid * oc.shapefit_factor(q, template.kp, template.a, dm=p.get('dm', 0.), dn=p.get('dn', 0.))
        pt = oc.tns_pktable(theory.k, theory.mu, theory.wmu, q, pk_q, template.f_fid * p.get('df', 1.), qpar=p.get('qpar', 1.), qper=p.get('qper', 1.), sigmav=p['sigmav'], fog='gaussian',
                            kernels=kernels, k11=k11)
        power = oc.tns_tracer_power(pt, theory.nd, b1=p['b1'], b2=p['b2'], bs=p['bs'], b3=p['b3'], sn0=p['sn0'])
        flat = oc.window_apply(power, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout)
        ref = oc.gaussian_loglikelihood(flat, obs.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)


def _mirror_tns(cls_name, marg=None, xi=False, fog='lorentzian', seed=5):
    import desilike_amd.theories.galaxy_clustering as tg
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable, TracerCorrelationFunctionMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    kw = {} if 'EFTLike' in cls_name else {'fog': fog}
    theory = getattr(tg, cls_name)(template=template, **kw)
    for pname, conf in (marg or {}).items(): theory.init.params[pname].update(**conf)
    rng = np.random.RandomState(seed)
    data = {'b1': 2., 'b2': 0.4}
    if xi:
        obs = TracerCorrelationFunctionMultipolesObservable(data=data, s=np.linspace(32.5, 147.5, 16), ells=(0, 2), theory=theory)
        n = 32
    else:
        obs = TracerPowerSpectrumMultipolesObservable(data=data, kedges=np.linspace(0., 0.2, 21), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory, shotnoise=1e4)
        n = 60
    A = rng.standard_normal((n, n))
    scale = 1e-4 if xi else 50.
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=(A.dot(A.T) + 10. * n * np.eye(n)) * scale**2)
    like.initialize()
    return like, obs, theory, template


def _oracle_flat(theory, template, obs, p, ct=None, sn=None):
    from oracle import np_oracle as oc
    q = template.k
    k11 = oc.tns_k11(theory.k)
    if not hasattr(theory, '_oracle_kernels'): theory._oracle_kernels = oc.tns_kernels(k11, q, oc.weights_trapz(q))
    pk_q = template.pk_dd_fid * oc.shapefit_factor(q, template.kp, template.a, dm=p.get('dm', 0.), dn=p.get('dn', 0.))
    pt = oc.tns_pktable(theory.k, theory.mu, theory.wmu, q, pk_q, template.f_fid * p.get('df', 1.), qpar=p.get('qpar', 1.), qper=p.get('qper', 1.), sigmav=p.get('sigmav', 0.),
                        fog=theory.fog, kernels=theory._oracle_kernels, k11=k11)
    power = oc.tns_tracer_power(pt, theory.nd, b1=p['b1'], b2=p.get('b2', 0.), bs=p.get('bs', 0.), b3=p.get('b3', 0.), sn0=p.get('sn0', 0.))
    if ct is not None:
        power = oc.eftlike_addon(power, list(theory.ells), pt['pk11'], theory.counterterm_matrix, [2. * v for v in ct], theory.stochastic_matrix, sn, theory.nd)
    return power


def test_analytic_marginalisation_of_tns_terms():
    """sn0 on EVERY multipole (full_shape.py:961) and an EFT-like counter term (derivative proportional to the projected linear spectrum) solved analytically: the
    device's solve against the oracle's ``_solve`` restatement with derivative columns from unit steps of the (exactly linear) parameters."""
    from oracle import np_oracle as oc
    like, obs, theory, template = _mirror_tns('EFTLikeTNSTracerPowerSpectrumMultipoles',
                                              marg={'sn0': dict(derived='.marg', prior=dict(dist='norm', loc=0., scale=2.)), 'ct0_2': dict(derived='.marg', prior=dict(dist='norm', loc=0., scale=40.)),
                                                    'ct2_2': dict(derived='.best'), 'ct4_2': dict(fixed=True, value=0.), 'sn2_2': dict(fixed=True, value=0.), 'sn4_2': dict(fixed=True, value=0.)})
    names = like.varied_params.names()
    solved = like.solved_params.names()
    assert sorted(solved) == ['ct0_2', 'ct2_2', 'sn0']
    rng = np.random.RandomState(3)
    theta = np.column_stack([param.ref.sample(size=9, random_state=rng) for param in like.varied_params])
    loglike, logprior, status, xs = like._get_context().eval_batch_host(theta, return_solved=True)
    assert (status == 0).all()
    wm = obs.wmatrix
    ctn, snn = theory.counterterm_params, theory.stochastic_params
    x0 = np.array([param.value for param in like.solved_params])
    loc = np.array([param.prior.loc if param.prior.dist == 'norm' else 0. for param in like.solved_params])
    scale = np.array([param.prior.scale if param.prior.dist == 'norm' else np.inf for param in like.solved_params])
    mask = [str(param.derived).startswith('.marg') for param in like.solved_params]

    def flat(row, x):
        p = dict(zip(names, row)); p.update(dict(zip(solved, x)))
        power = _oracle_flat(theory, template, obs, p, ct=[p.get(n, 0.) for n in ctn], sn=[p.get(n, 0.) for n in snn])
        return oc.window_apply(power, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout)

    for i, row in enumerate(theta[:5]):
        f0 = flat(row, x0)
        T = np.array([flat(row, x0 + np.eye(len(solved))[s]) - f0 for s in range(len(solved))])
        sol = oc.solve_marginalized(f0 - obs.flatdata, T, like.precision, x0=x0, prior_loc=loc, prior_scale=scale, marg_mask=mask)
        assert abs(loglike[i] - sol['loglikelihood']) <= 1e-10 * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
        assert np.allclose(xs[i], sol['x'], rtol=1e-7, atol=1e-9)


def test_correlation_function_multipoles():
    """TNS xi_ell: the power spectrum multipoles on the 300-point log grid (k11: 480 table wavenumbers), Hankel transform folded into the window."""
    from oracle import np_oracle as oc
    like, obs, theory, template = _mirror_tns('TNSTracerCorrelationFunctionMultipoles', xi=True, fog='gaussian')
    names = like.varied_params.names()
    assert 'sn0' not in names and 'sigmav' in names
    rng = np.random.RandomState(1)
    theta = np.column_stack([param.ref.sample(size=3, random_state=rng) if param.ref.is_proper() else np.full(3, param.value) for param in like.varied_params])
    theta[:, names.index('sigmav')] = rng.uniform(0.5, 3., 3)
    loglike = like._get_context().eval_batch_host(theta)[0]
    for i, row in enumerate(theta):
        p = dict(zip(names, row))
        power = _oracle_flat(theory, template, obs, p)
        flat = np.ravel(oc.get_corr(power, theory.kin, theory.s, theory.ells))
        ref = oc.gaussian_loglikelihood(flat, obs.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)


def test_ensemble_sampler_on_a_tns_likelihood():
    """The device-resident stretch-move sampler drives the TNS likelihood like any other: chain log-posteriors equal direct evaluations of the chain's points."""
    from desilike_amd.samplers import EmceeSampler
    like, obs, theory, template = _mirror_tns('TNSTracerPowerSpectrumMultipoles')
    sampler = EmceeSampler(like, nwalkers=32, seed=7)
    sampler.run(niterations=6)
    chain = sampler.chain
    names = like.varied_params.names()
    last = np.column_stack([np.asarray(chain[name])[-1] for name in names])
    direct = like._get_posterior_context()[0].eval_logposterior_host(last)[0] + like._get_posterior_context()[1]
    assert np.allclose(np.asarray(chain['logposterior'])[-1], direct, rtol=1e-12, atol=1e-9)
    assert 0. < np.mean(sampler.acceptance_fraction) <= 1.


def test_tns_next_to_a_kaiser_observable():
    """Two observables in one likelihood, one on the TNS theory (tracer namespace) and one on the Kaiser theory, shared template: every observable goes through its own
    kernels into its columns of the theory vector; the log-likelihood equals the oracle's on the concatenated theory."""
    from oracle import np_oracle as oc
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, TNSTracerPowerSpectrumMultipoles, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    tns = TNSTracerPowerSpectrumMultipoles(template=template, tracers='LRG')
    kaiser = KaiserTracerPowerSpectrumMultipoles(template=template, tracers='ELG')
    obs1 = TracerPowerSpectrumMultipolesObservable(data={'LRG.b1': 2., 'LRG.b2': 0.3}, kedges=np.linspace(0., 0.2, 21), ells=(0, 2), wmatrix={'resolution': 2}, theory=tns, shotnoise=1e4)
    obs2 = TracerPowerSpectrumMultipolesObservable(data={'ELG.b1': 1.3}, kedges=np.linspace(0., 0.15, 16), ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=kaiser, shotnoise=4e3)
    rng = np.random.RandomState(8)
    n = 40 + 45
    A = rng.standard_normal((n, n)) * 40.
    like = ObservablesGaussianLikelihood(observables=[obs1, obs2], covariance=A.dot(A.T) + 1e5 * np.eye(n))
    like.initialize()
    names = like.varied_params.names()
    assert {'LRG.b1', 'LRG.b2', 'LRG.sn0', 'ELG.b1', 'ELG.sn0', 'sigmav'} <= set(names)
    theta = np.column_stack([param.ref.sample(size=4, random_state=rng) if param.ref.is_proper() else np.full(4, param.value) for param in like.varied_params])
    theta[:, names.index('sigmav')] = rng.uniform(0.5, 4., 4)
    loglike = like._get_context().eval_batch_host(theta)[0]
    for i, row in enumerate(theta):
        p = dict(zip(names, row))
        shared = {key: p[key] for key in ['qpar', 'qper', 'dm', 'df'] if key in p}
        p1 = dict(shared, b1=p['LRG.b1'], b2=p['LRG.b2'], sn0=p['LRG.sn0'], sigmav=p['sigmav'])
        power1 = _oracle_flat(tns, template, obs1, p1)
        flat1 = oc.window_apply(power1, matrix_full=obs1.wmatrix.matrix_full, shotnoisein=obs1.wmatrix.shotnoisein, shotnoiseout=obs1.wmatrix.shotnoiseout)
        q = template.k
        pk_q = template.pk_dd_fid * oc.shapefit_factor(q, template.kp, template.a, dm=p.get('dm', 0.), dn=0.)
        dd, dt, tt = oc.kaiser_pktable(kaiser.k, kaiser.mu, kaiser.wmu, q, pk_q, template.f_fid * p.get('df', 1.), qpar=p.get('qpar', 1.), qper=p.get('qper', 1.))
        power2 = oc.kaiser_tracer_power(kaiser.ells, dd, dt, tt, kaiser.nd, p['ELG.b1'], p['ELG.b1'], p['ELG.sn0'])
        flat2 = oc.window_apply(power2, matrix_full=obs2.wmatrix.matrix_full, shotnoisein=obs2.wmatrix.shotnoisein, shotnoiseout=obs2.wmatrix.shotnoiseout)
        ref = oc.gaussian_loglikelihood(np.concatenate([flat1, flat2]), like.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)


def test_large_batch_properties(contexts):
    """4100 rows (one wavenumber per wave; ragged last tile) built from 20 distinct points: equal rows give bit-equal results wherever they sit in the batch, and the
    numbers agree with the 20-point evaluation (split-K loop kernel, one point per assembly workgroup) to 1e-12 -- summation orders differ, nothing else."""
    g, ctx = contexts('tns')
    ok = np.isfinite(g['theta']).all(axis=1) & np.isfinite(g['logprior'])
    base = g['theta'][ok][:20]
    small = ctx.eval_batch_host(base)[0]
    assert (np.abs(small - g['loglikelihood'][ok][:20]) <= 1e-10 * np.maximum(1., np.abs(small))).all()
    rng = np.random.RandomState(0)
    idx = rng.randint(0, len(base), size=4100)
    big = ctx.eval_batch_host(base[idx])[0]
    for i in range(len(base)):
        vals = big[idx == i]
        assert len(vals) > 100 and (vals == vals[0]).all(), i
    assert (np.abs(big - small[idx]) <= 1e-12 * np.maximum(1., np.abs(small[idx]))).all(), (np.abs(big - small[idx]) / np.maximum(1., np.abs(small[idx]))).max()


def test_eftlike_correlation_function_multipoles():
    """EFT-like TNS xi_ell (counter terms times the projected linear spectrum, Hankel operator folded into the window; no stochastic terms in configuration space)."""
    from oracle import np_oracle as oc
    like, obs, theory, template = _mirror_tns('EFTLikeTNSTracerCorrelationFunctionMultipoles', xi=True)
    names = like.varied_params.names()
    assert 'ct0_2' in names and 'ct2_2' in names and not any(name.startswith('sn') for name in names) and 'sigmav' not in names
    rng = np.random.RandomState(4)
    theta = np.column_stack([param.ref.sample(size=2, random_state=rng) if param.ref.is_proper() else np.full(2, param.value) for param in like.varied_params])
    loglike = like._get_context().eval_batch_host(theta)[0]
    for i, row in enumerate(theta):
        p = dict(zip(names, row))
        power = _oracle_flat(theory, template, obs, p, ct=[p.get(n, 0.) for n in theory.counterterm_params], sn=[0. for n in theory.stochastic_params])
        flat = np.ravel(oc.get_corr(power, theory.kin, theory.s, theory.ells))
        ref = oc.gaussian_loglikelihood(flat, obs.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)


def test_edge_cases(contexts):
    """NaN inputs, a single point (the scalar ``likelihood()`` call surface), an empty batch, and the 'data generated from the theory' identity."""
    g, ctx = contexts('tns')
    ok = np.isfinite(g['theta']).all(axis=1) & np.isfinite(g['logprior'])
    theta = g['theta'][ok][:5].copy()
    theta[1, 2] = np.nan
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert status[1] == 3 and status[0] == 0 and np.isfinite(loglike[[0, 2, 3, 4]]).all()
    l1 = ctx.eval_batch_host(g['theta'][ok][:1])[0]
    assert abs(l1[0] - g['loglikelihood'][ok][0]) <= 1e-10 * max(1., abs(g['loglikelihood'][ok][0]))
    assert ctx.eval_batch_host(np.zeros((0, ctx.n_params)))[0].size == 0
    # the fixture's data are the reference's theory at (b1, b2, sigmav) = (2, 0.5, 3), everything else at its default: logL = 0 there (likelihoods/tests/test_galaxy_clustering.py:6-16)
    names = [str(n) for n in g['names']]
    fid = dict(qpar=1., qper=1., dm=0., df=1., sigmav=3., b1=2., b2=0.5, bs=0., b3=0., sn0=0.)
    lf = ctx.eval_batch_host(np.array([[fid[name] for name in names]]))[0]
    assert abs(lf[0]) < 1e-9, lf
