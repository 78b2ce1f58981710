"""Multi-parameter pin of analytic marginalisation (row a10) on REFERENCE outputs: fixture marg_multi.npz holds, for a few points, the exact quadratic form
(c, g, H) of the reference's NON-marginalised log-posterior in the linear parameters (tests/golden/make_golden.py::marg_multi: the reference evaluated on a stencil;
its own `_solve` needs jax).  Closed forms (reference conventions, likelihoods/base.py:394-404):  x* = x0 - H^-1 g,  logposterior(.best) = c - g H^-1 g / 2,
logposterior(.marg over M) = that - logdet(-H[M, M]) / 2.   CPU: the oracle's ``solve_marginalized`` on the oracle's own theory vectors; GPU: the HIP path.
Cases: (a) EFT-like Kaiser, two counter terms (point-dependent derivative rows) + one stochastic term; (b) two tracers, both shot-noise terms."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, constants_from_mirror


def make_case(case, derived):
    """Host mirror of the fixture's pipeline with the solved parameters flagged ``derived`` (list, one entry per solved parameter)."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, EFTLikeKaiserTracerPowerSpectrumMultipoles, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('marg_multi')[case]
    template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    if case == 'a':
        theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=template)
        for name, (loc, scale), flag in zip(['ct0_2', 'ct2_2', 'sn0_2'], g['prior'], derived):
            theory.init.params[name].update(prior=dict(dist='norm', loc=loc, scale=scale), derived=flag)
        for name in ['ct4_2', 'sn2_2', 'sn4_2']:
            theory.init.params[name].update(fixed=True, value=0.)
        observables = [TracerPowerSpectrumMultipolesObservable(data=g['flatdata'], kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)]
    else:
        observables = []
        for iobs, (tracer, kmax) in enumerate([('LRG', 0.2), ('ELG', 0.15)]):
            theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
            theory.init.params[tracer + '.sn0'].update(prior=dict(dist='norm', loc=g['prior'][iobs][0], scale=g['prior'][iobs][1]), derived=derived[iobs])
            nk = int(round(kmax / 0.005))
            observables.append(TracerPowerSpectrumMultipolesObservable(data=g['flatdata{:d}'.format(iobs)], kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 4},
                                                                       theory=theory, shotnoise=1e4 if tracer == 'LRG' else 4e3))
    return g, ObservablesGaussianLikelihood(observables=observables, covariance=g['covariance'])


def closed_form(g, ip, marg_mask):
    c, grad, H = g['c'][ip], g['g'][ip], g['H'][ip]
    dx = -np.linalg.solve(H, grad)
    value = c - 0.5 * grad.dot(np.linalg.solve(H, grad))
    marg = np.flatnonzero(marg_mask)
    if marg.size: value -= 0.5 * np.linalg.slogdet(-H[np.ix_(marg, marg)])[1]
    return g['x0'] + dx, value


@pytest.mark.parametrize('case,flags', [('a', ['.marg', '.best', '.marg']), ('a', ['.marg', '.marg', '.marg']), ('b', ['.marg', '.marg']), ('b', ['.best', '.marg'])])
def test_oracle_solve_vs_reference_quadratic_form(case, flags):
    g, like = make_case(case, [False] * len(flags))      # mirror used for constants only (no GPU): the oracle evaluates theory vectors
    like.initialize()
    names = like.varied_params.names()
    solved = [str(n) for n in g['solved']]
    others = [str(n) for n in g['names']]
    constants = [constants_from_mirror(obs) for obs in like.observables]
    precision = like.precision
    flatdata = np.concatenate([c['flatdata'] for c in constants])

    def flattheory(p):
        out = []
        for iobs, c in enumerate(constants):
            q = dict(p)
            if case == 'a':
                q['b1'] = (p['b1'], p['b1'])
                q['ct'] = [2. * p.get(str(n), 0.) for n in c['ct_params']]        # auto-spectrum: both tracer inputs of a term are the same parameter
                q['sn'] = [p.get(str(n), 0.) for n in c['sn_params']]
            else:
                tracer = ['LRG', 'ELG'][iobs]
                q['b1'] = (p[tracer + '.b1'], p[tracer + '.b1']); q['sn0'] = p[tracer + '.sn0']
            out.append(orc.fullshape_observable(c, q)['flattheory'])
        return np.concatenate(out)

    loc, scale = g['prior'][:, 0], g['prior'][:, 1]
    mask = np.array([flag == '.marg' for flag in flags])
    priors = {param.name: param.prior for param in like.varied_params}
    for ip, row in enumerate(g['theta']):
        p = dict(zip(others, row))
        p.update(dict(zip(solved, g['x0'])))
        f0 = flattheory(p)
        T = np.array([flattheory({**p, name: p[name] + 1.}) - f0 for name in solved])
        sol = orc.solve_marginalized(f0 - flatdata, T, precision, x0=g['x0'], prior_loc=loc, prior_scale=scale, marg_mask=mask)
        logprior_others = sum(float(priors[name](p[name])) for name in others)
        xstar, expected = closed_form(g, ip, mask)
        assert np.allclose(sol['x'], xstar, rtol=1e-7, atol=1e-8), (ip, sol['x'], xstar)
        total = sol['loglikelihood'] + sol['logprior_solved'] + logprior_others
        assert abs(total - expected) <= 1e-8 * max(1., abs(expected)), (ip, total, expected)


@pytest.mark.gpu
@pytest.mark.parametrize('case,flags', [('a', ['.marg', '.best', '.marg']), ('a', ['.marg', '.marg', '.marg']), ('b', ['.marg', '.marg']), ('b', ['.best', '.marg'])])
def test_hip_marginalisation_vs_reference_quadratic_form(case, flags):
    from desilike_amd import vmap
    g, like = make_case(case, flags)
    others = [str(n) for n in g['names']]
    solved = [str(n) for n in g['solved']]
    assert sorted(like.varied_params.names()) == sorted(others) and sorted(like.solved_params.names()) == sorted(solved)   # (the reference orders the tracers' blocks differently)
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(others)})
    assert errors == {}
    mask = np.array([flag == '.marg' for flag in flags])
    for ip in range(len(g['theta'])):
        xstar, expected = closed_form(g, ip, mask)
        assert abs(logpost[ip] - expected) <= 1e-8 * max(1., abs(expected)), (ip, logpost[ip], expected)
        assert np.allclose([derived[name][ip] for name in solved], xstar, rtol=1e-7, atol=1e-8)
    # the Hessian entries the reference attaches to loglikelihood + logprior are the quadratic form's H (likelihoods/base.py:372, 389)
    for i1, p1 in enumerate(solved):
        for i2 in range(i1, len(solved)):
            p2 = solved[i2]
            total = derived['loglikelihood.{}.{}'.format(p1, p2)] + (derived['logprior.{}.{}'.format(p1, p2)] if p1 == p2 else 0.)
            assert np.allclose(total, g['H'][:, i1, i2], rtol=1e-7, atol=1e-9 * np.abs(g['H']).max())
