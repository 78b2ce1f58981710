"""GPU (-m gpu): analytic marginalisation / best fit of linear nuisance parameters on the HIP path (dl_finalize_marg_kernel)
against (i) the reference's own non-marginalised posterior on a grid (fixture from the reference) and (ii) the NumPy oracle
restatement of likelihoods/base.py:129-200, 314-413 on seeded inputs, including point-dependent derivative columns (counter terms)."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, observable_constants
from test_oracle_marg import reference_parabola

pytestmark = pytest.mark.gpu


def make_marg_likelihood(g, solved='.marg'):
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    theory.init.params['sn0'].update(prior=dict(dist='norm', loc=0.2, scale=1.5), derived=solved)
    obs = TracerPowerSpectrumMultipolesObservable(data=g['obs0']['flatdata'], kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=1e4)
    return ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])


@pytest.mark.parametrize('solved', ['.marg', '.best', '.auto'])
def test_sn0_marginalisation_vs_reference_grid(solved):
    from desilike_amd import vmap
    g = load_golden('marg_sn0_grid')
    like = make_marg_likelihood(g, solved=solved)
    names = [str(n) for n in g['names']]
    assert like.varied_params.names() == [name for name in names if name != 'sn0'] and like.solved_params.names() == ['sn0']
    theta = {name: g['theta'][:, i] for i, name in enumerate(names) if name != 'sn0'}
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)(theta)
    assert errors == {}
    for ip in range(len(g['theta'])):
        a, smax, cmax, resid = reference_parabola(g, ip)
        expected = cmax - (0.5 * np.log(a) if solved != '.best' else 0.)
        assert abs(logpost[ip] - expected) < 1e-6 * max(1., abs(expected)), (ip, logpost[ip], expected)
        assert np.isclose(derived['sn0'][ip], smax, rtol=1e-6, atol=1e-8)


def test_marg_vs_oracle_with_counterterms():
    """EFT-like Kaiser: ct (point-dependent derivative) and sn terms solved; two of them marginalised, one at best fit, flat and Gaussian priors."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, EFTLikeKaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg2_shapefit_window_dense')
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=template)
    theory.init.params['ct0_2'].update(derived='.marg', prior=dict(dist='norm', loc=0., scale=30.))
    theory.init.params['ct2_2'].update(derived='.best', prior=dict(dist='norm', loc=1., scale=50.))
    theory.init.params['sn0_2'].update(derived='.marg', prior=dict(dist='uniform'))
    for name in ['ct4_2', 'sn2_2', 'sn4_2']:
        theory.init.params[name].update(fixed=True, value=0.)
    obs = TracerPowerSpectrumMultipolesObservable(data=g['obs0']['flatdata'], kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix=g['obs0']['matrix_full'], kin=g['obs0']['kin'],
                                                  ellsin=(0, 2, 4), theory=theory, shotnoise=1e4)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    names = like.varied_params.names()
    assert like.solved_params.names() == ['ct0_2', 'ct2_2', 'sn0_2']
    rng = np.random.RandomState(5)
    theta = np.column_stack([param.ref.sample(size=33, random_state=rng) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status, solved = ctx.eval_batch_host(theta, return_solved=True)
    assert (status == 0).all()
    # derived outputs: likelihood Hessian w.r.t. the solved parameters, same numbers through the derived entry point and the call surface
    d_loglike, d_logprior, d_status, d_solved, hessian = ctx.eval_batch_derived_host(theta)
    assert np.array_equal(d_loglike, loglike) and np.array_equal(d_logprior, logprior) and np.array_equal(d_solved, solved) and np.array_equal(d_status, status)
    from desilike_amd import vmap
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: theta[:, i] for i, name in enumerate(names)})
    assert errors == {} and np.allclose(logpost, loglike + logprior, rtol=1e-14, atol=0.)
    assert np.array_equal(derived['loglikelihood.ct0_2.sn0_2'], hessian[:, 0, 2]) and np.array_equal(derived['ct2_2'], solved[:, 1])
    assert np.allclose(derived['logprior.ct0_2.ct0_2'], -1. / 30.**2) and np.allclose(derived['logprior.sn0_2.sn0_2'], 0.)
    # oracle: theory at x0 and derivative columns by unit steps of the (exactly linear) solved parameters
    c = observable_constants(g)
    theory.initialize()
    c.update(ct_matrix=theory.counterterm_matrix, sn_matrix=theory.stochastic_matrix)
    ctn, snn = theory.counterterm_params, theory.stochastic_params
    x0 = np.array([param.value for param in like.solved_params])

    def flat(row, x):
        p = dict(zip(names, row)); p['b1'] = (p['b1'], p['b1'])
        vals = dict(zip(['ct0_2', 'ct2_2', 'sn0_2'], x))
        p['ct'] = [2. * vals.get(n, 0.) for n in ctn]
        p['sn'] = [vals.get(n, p.get(n, 0.)) if n != 'sn0' else p['sn0'] for n in snn]
        return orc.fullshape_observable(c, p)['flattheory']

    for i, row in enumerate(theta):
        f0 = flat(row, x0)
        T = np.array([flat(row, x0 + np.eye(3)[s]) - f0 for s in range(3)])
        sol = orc.solve_marginalized(f0 - c['flatdata'], T, like.precision, x0=x0, prior_loc=[0., 1., 0.], prior_scale=[30., 50., np.inf], marg_mask=[True, False, True])
        assert abs(loglike[i] - sol['loglikelihood']) <= 1e-10 * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
        assert np.allclose(solved[i], sol['x'], rtol=1e-8, atol=1e-10)
        assert np.allclose(hessian[i], sol['likelihood_hessian'], rtol=1e-10, atol=1e-12 * np.abs(sol['likelihood_hessian']).max())
        lp_ref = orc.logprior(row, [dict(dist=['uniform', 'norm'][int(pr[0])], limits=(pr[1], pr[2]), loc=pr[3], scale=pr[4]) for pr in [p.prior.spec() for p in like.varied_params]]) + sol['logprior_solved']
        assert np.isclose(logprior[i], lp_ref, rtol=1e-10, atol=1e-10)


def test_marg_two_tracers_vs_oracle():
    """Two observables with a full cross-covariance (BASELINE config 5 shape): each tracer's shot-noise term solved analytically, one marginalised and one at
    its best fit; the oracle solves the stacked system (likelihoods/base.py:314-413 over the concatenated data vector)."""
    g = load_golden('cfg5_two_tracers')
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    observables = []
    for iobs, (tracer, kmax, mode) in enumerate([('LRG', 0.2, '.marg'), ('ELG', 0.15, '.best')]):
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        theory.init.params[tracer + '.sn0'].update(derived=mode, prior=dict(dist='norm', loc=0., scale=2.) if tracer == 'LRG' else dict(dist='uniform'))
        nk = int(round(kmax / 0.005))
        observables.append(TracerPowerSpectrumMultipolesObservable(data=g['obs{:d}'.format(iobs)]['flatdata'], kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4),
                                                                   wmatrix={'resolution': 4}, theory=theory, shotnoise=1e4 if tracer == 'LRG' else 4e3))
    like = ObservablesGaussianLikelihood(observables=observables, covariance=g['covariance'])
    names = like.varied_params.names()
    solved = like.solved_params.names()
    assert solved == ['LRG.sn0', 'ELG.sn0'] and 'LRG.sn0' not in names
    rng = np.random.RandomState(8)
    theta = np.column_stack([np.clip(param.ref.sample(size=20, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status, xs = ctx.eval_batch_host(theta, return_solved=True)
    assert (status == 0).all()
    cs = [observable_constants(g, iobs) for iobs in range(2)]
    flatdata = np.concatenate([c['flatdata'] for c in cs])

    def flat(row, x):
        p = dict(zip(names, row))
        out = []
        for tracer, c, sn0 in zip(['LRG', 'ELG'], cs, x):
            q = {name: p[name] for name in ['qpar', 'qper', 'dm', 'df']}
            q.update(b1=(p[tracer + '.b1'], p[tracer + '.b1']), sn0=sn0)
            out.append(orc.fullshape_observable(c, q)['flattheory'])
        return np.concatenate(out)

    x0 = np.array([param.value for param in like.solved_params])
    for i, row in enumerate(theta):
        f0 = flat(row, x0)
        T = np.array([flat(row, x0 + np.eye(2)[s]) - f0 for s in range(2)])
        sol = orc.solve_marginalized(f0 - flatdata, T, like.precision, x0=x0, prior_loc=[0., 0.], prior_scale=[2., np.inf], marg_mask=[True, False])
        assert abs(loglike[i] - sol['loglikelihood']) <= 1e-10 * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
        assert np.allclose(xs[i], sol['x'], rtol=1e-8, atol=1e-10)


def test_prec_one_off_precision_marginalisation():
    """'.prec' (likelihoods/base.py:257-312): the linear parameter is marginalised ONCE into the precision matrix and the data vector.  Against the oracle's
    restatement with the derivative taken through the oracle, the Woodbury identity P_new^-1 = C + scale^2 T T^T, and the per-point '.marg' result:
    logposterior('.prec') = logposterior('.marg') + 1/2 log(T P T^T + 1 / scale^2) when the derivative does not depend on the other parameters (sn0)."""
    from desilike_amd import vmap
    g = load_golden('marg_sn0_grid')
    like, like_marg = make_marg_likelihood(g, solved='.prec'), make_marg_likelihood(g, solved='.marg')
    names = [str(n) for n in g['names']]
    vnames = [name for name in names if name != 'sn0']
    assert like.varied_params.names() == vnames and like.prec_params.names() == ['sn0'] and len(like.solved_params) == 0
    assert like.all_params['sn0'].value == 0.
    theta = {name: g['theta'][:, i] for i, name in enumerate(names) if name != 'sn0'}
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)(theta)
    (logpost_marg, derived_marg), errors_marg = vmap(like_marg, errors='return', return_derived=True)(theta)
    assert errors == {} and errors_marg == {}
    # derivative through the oracle at the default values of the other parameters
    c = observable_constants(g)
    p = {param.name: param.value for param in like.varied_params}
    p['b1'] = (p['b1'], p['b1'])
    T = (orc.fullshape_observable(c, dict(p, sn0=1.))['flattheory'] - orc.fullshape_observable(c, dict(p, sn0=0.))['flattheory'])[None, :]
    assert np.allclose(like.prec_derivatives, T, rtol=1e-9, atol=1e-12 * np.abs(T).max())
    P0 = like._precision_input
    P_ref = orc.marginalize_precision(P0, T, [1.5])
    assert np.allclose(like.precision, P_ref, rtol=1e-8, atol=1e-10 * np.abs(P_ref).max())
    cov_new = np.linalg.inv(P0) + 1.5**2 * T.T.dot(T)
    assert np.allclose(like.precision.dot(cov_new), np.eye(len(cov_new)), rtol=0., atol=1e-8)
    assert np.allclose(like.flatdata, c['flatdata'] - 0.2 * T[0], rtol=1e-12, atol=1e-12 * np.abs(c['flatdata']).max())
    a = float(T.dot(P0).dot(T.T)[0, 0]) + 1.5**(-2)
    assert np.allclose(logpost, logpost_marg + 0.5 * np.log(a), rtol=1e-9, atol=1e-9)
    # and directly against the oracle's Gaussian likelihood with the marginalised precision and shifted data
    for i in range(0, len(logpost), 7):
        q = dict(zip(names, g['theta'][i])); q['b1'] = (q['b1'], q['b1']); q['sn0'] = 0.
        ref = orc.gaussian_loglikelihood(orc.fullshape_observable(c, q)['flattheory'], c['flatdata'] - 0.2 * T[0], P_ref)[0]
        assert abs(derived['loglikelihood'][i] - ref) <= 1e-10 * max(1., abs(ref))


@pytest.mark.parametrize('case', ['sn0_marg', 'sn0_best', 'bao_broadband', 'templates'])
def test_posterior_context_equals_per_point_marginalisation(case):
    """Solved parameters with point-independent derivative rows: marginalising them once into the precision factor (``_get_posterior_context``: what the samplers'
    fast path evaluates, one chi2 GEMM, no per-point solve) gives the same loglikelihood + logprior as the per-point solve (likelihoods/base.py:314-413) -- Gaussian and
    flat priors (singular marginalised precision), '.marg' and '.best', x0 != loc."""
    if case.startswith('sn0'):
        g = load_golden('marg_sn0_grid')
        like = make_marg_likelihood(g, solved='.marg' if case == 'sn0_marg' else '.best')
    elif case == 'bao_broadband':
        from test_host_api import make_cfg4
        g, like = make_cfg4('xi')
        like.initialize()
        for param in like.observables[0].wmatrix.theory.init.params.select(basename='al*'):
            param.update(derived='.marg')
        like._invalidate()
    else:
        from test_window_extras import make_likelihood
        g = load_golden('cfg2_fc_syst')
        like = make_likelihood(g, solved='.marg')[0]
    assert like._solved_are_constant()
    rng = np.random.RandomState(11)
    theta = np.column_stack([np.clip(param.ref.sample(size=64, random_state=rng), *param.prior.limits) for param in like.varied_params])
    theta[5, 0] = np.nan
    loglike, logprior, status = like._get_context().eval_batch_host(theta)[:3]
    ctx, offset = like._get_posterior_context()
    assert ctx.n_solved == 0
    logpost, status_p = ctx.eval_logposterior_host(theta)
    logpost = logpost + offset
    ok = status == 0
    assert ok.sum() == 63 and np.array_equal(status_p == 0, ok) and np.isneginf(logpost[~ok]).all()
    ref = loglike[ok] + logprior[ok]
    assert (np.abs(logpost[ok] - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), np.abs(logpost[ok] - ref).max()
    # the samplers take this route
    from desilike_amd.samplers import BasePosteriorSampler
    assert np.allclose(BasePosteriorSampler(like).logposterior(theta[:5]), ref[:5], rtol=1e-9, atol=1e-9)
    import torch
    dev = like.evaluate_logposterior(torch.as_tensor(theta, dtype=torch.float64, device='cuda:0').contiguous())
    torch.cuda.synchronize()
    assert np.allclose(dev.cpu().numpy()[ok], ref, rtol=1e-9, atol=1e-9)


def test_posterior_context_falls_back_for_point_dependent_derivatives():
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, EFTLikeKaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg2_shapefit_window')
    theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    theory.init.params['ct0_2'].update(derived='.marg')
    obs = TracerPowerSpectrumMultipolesObservable(data=g['obs0']['flatdata'], kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=1e4)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    assert not like._solved_are_constant()
    ctx, offset = like._get_posterior_context()
    assert ctx is like._get_context() and offset == 0. and ctx.n_solved == 1
