"""GPU (-m gpu): full-shape correlation function multipoles (Kaiser / EFT-like Kaiser xi_ell, full_shape.py:553-574, 664-687) through the reference's call
surface: the device evaluates P_ell on the 300-point log grid, the Hankel operator (built by one batch of the device FFTLog) is folded into the window.
Against fixtures captured from the reference and the oracle on a seeded batch; counter terms marginalised analytically."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden
from test_oracle_kaiser_xi import kaiser_xi_point

pytestmark = pytest.mark.gpu


def make_kaiser_xi(name, data=None):
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerCorrelationFunctionMultipoles, EFTLikeKaiserTracerCorrelationFunctionMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden(name)
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    kwargs = dict(interp_order=3) if name.endswith('cubic') else {}
    theory = (EFTLikeKaiserTracerCorrelationFunctionMultipoles if name.endswith('eft') else KaiserTracerCorrelationFunctionMultipoles)(template=template, **kwargs)
    obs = TracerCorrelationFunctionMultipolesObservable(data=g['obs0']['flatdata'] if data is None else data, s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory)
    return g, ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])


@pytest.mark.parametrize('name', ['kaiser_xi', 'kaiser_xi_eft', 'kaiser_xi_cubic'])
def test_kaiser_xi_call_surface_vs_reference(name):
    from desilike_amd import vmap
    g, like = make_kaiser_xi(name)
    c = g['obs0']
    rnames = [str(n) for n in g['names']]
    assert like.varied_params.names() == rnames
    spec = like._spec({}, like._flatdata_list(), like.precision)['observables'][0]
    for key, ref in [('kin', c['kin']), ('mu', c['mu']), ('wmu_ell', c['wmu_ell']), ('k_t', c['k11']), ('pk_dd_fid', c['pk_dd_fid']), ('f_fid', c['f_fid']), ('nd', c['nd'])]:
        assert np.allclose(np.ravel(spec[key]), np.ravel(ref), rtol=1e-13, atol=1e-300), key
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({n: g['theta'][:, i] for i, n in enumerate(rnames)})
    assert errors == {}
    assert (np.abs(derived['loglikelihood'] - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all()
    assert np.allclose(derived['logprior'], g['logprior'], rtol=1e-13, atol=1e-13)
    ctx = like._get_context()
    power = ctx.eval_theory_host(g['theta'], iobs=0)
    assert np.allclose(power, g['power'], rtol=1e-11, atol=1e-12 * np.abs(g['power']).max())
    flat = ctx.eval_batch_host(g['theta'], return_flattheory=True)[3]
    assert np.allclose(flat, g['flattheory'], rtol=1e-9, atol=1e-12 * np.abs(g['flattheory']).max())
    # data generated from theory => likelihood(fiducial) == 0 (likelihoods/tests/test_galaxy_clustering.py:6-16)
    g2, like2 = make_kaiser_xi(name, data={'b1': 2.})
    assert abs(like2(b1=2.)) < 1e-12
    assert np.allclose(like2.observables[0].flatdata, c['flatdata'], rtol=1e-9, atol=1e-12 * np.abs(c['flatdata']).max())


def test_kaiser_xi_eft_seeded_batch_and_marginalised_counterterms():
    g, like = make_kaiser_xi('kaiser_xi_eft')
    names = like.varied_params.names()
    rng = np.random.RandomState(31)
    theta = np.column_stack([np.clip(param.ref.sample(size=4096, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert (status == 0).all() and np.isfinite(loglike).all()
    gfix = dict(g); gfix['names'] = np.array(names)
    for i in range(0, 4096, 256):
        corr = kaiser_xi_point(gfix, theta[i])[1]
        ref = orc.gaussian_loglikelihood(np.ravel(corr), g['obs0']['flatdata'], like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)
    # counter terms solved analytically: derivative rows depend on the point (ct_matrix . pk11), xi = Hankel . P
    g3, like3 = make_kaiser_xi('kaiser_xi_eft')
    like3.initialize()
    theory = like3.observables[0].wmatrix.theory
    for param in theory.init.params.select(basename='ct*'):
        param.update(derived='.marg')
    like3._invalidate()
    solved = like3.solved_params.names()
    assert solved == ['ct0_2', 'ct2_2', 'ct4_2']
    vnames = like3.varied_params.names()
    sub = theta[:12][:, [names.index(n) for n in vnames]]
    ll3, lp3, st3, xs = like3._get_context().eval_batch_host(sub, return_solved=True)
    assert (st3 == 0).all()
    for i in range(12):
        row = dict(zip(vnames, sub[i]))

        def flat(x):
            full = np.array([row[n] if n in row else x[solved.index(n)] for n in names])
            return np.ravel(kaiser_xi_point(gfix, full)[1])

        f0 = flat(np.zeros(3))
        T = np.array([flat(np.eye(3)[s]) - f0 for s in range(3)])
        sol = orc.solve_marginalized(f0 - g['obs0']['flatdata'], T, like3.precision, x0=np.zeros(3), prior_loc=np.zeros(3), prior_scale=np.full(3, 100.), marg_mask=np.ones(3, dtype='?'))
        assert abs(ll3[i] - sol['loglikelihood']) <= 1e-9 * max(1., abs(sol['loglikelihood'])), (ll3[i], sol['loglikelihood'])
        assert np.allclose(xs[i], sol['x'], rtol=1e-7, atol=1e-9)


def test_xi_systematic_template_is_an_additive_linear_term():
    """window.py:1363-1384, 731-733: flatcorr += syst_0 * template, on top of the fixture-pinned xi_ell."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerCorrelationFunctionMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('kaiser_xi')
    theory = KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    obs = TracerCorrelationFunctionMultipolesObservable(data=g['obs0']['flatdata'], s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory,
                                                        systematic_templates=[lambda ell, s: (ell == 2) * 1e-3 * (s / 100.)**-2])
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    names = like.varied_params.names()
    assert 'syst_0' in names
    rnames = [str(n) for n in g['names']]
    theta = np.column_stack([g['theta'][:, rnames.index(n)] if n in rnames else np.linspace(-1., 1., len(g['theta'])) for n in names])
    loglike, logprior, status, flat = like._get_context().eval_batch_host(theta, return_flattheory=True)
    s = np.linspace(22.5, 167.5, 30)
    template = np.concatenate([np.zeros(30), 1e-3 * (s / 100.)**-2, np.zeros(30)])
    expected = g['flattheory'] + theta[:, names.index('syst_0')][:, None] * template
    assert np.allclose(flat, expected, rtol=1e-9, atol=1e-12 * np.abs(expected).max())
    for i in range(len(theta)):
        ref = orc.gaussian_loglikelihood(expected[i], g['obs0']['flatdata'], like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-9 * max(1., abs(ref))
