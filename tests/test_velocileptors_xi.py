"""REPT velocileptors correlation function multipoles (full_shape.py:1603-1629: table combination on the 300-point log grid, then get_corr) against a fixture from the
reference run on a stand-in PT node (tests/golden/make_golden.py cfg3_table_xi).  CPU: oracle chain; GPU (-m gpu): emulated tables (exact Taylor emulator of the node),
separable feature path with the Hankel operator folded in, counter terms marginalised."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, prior_list
from emulator_utils import taylor_state, EMU_PARAMS
from test_oracle_emulator import table_point


def test_velocileptors_xi_chain_vs_reference():
    g = load_golden('cfg3_velocileptors_table_xi')
    c = g['obs0']
    state = taylor_state(g)
    priors = prior_list(g)
    for i, row in enumerate(g['theta']):
        power = table_point(g, row, state)
        assert np.allclose(power, g['power'][i], rtol=1e-12, atol=1e-12 * np.abs(g['power'][i]).max())
        corr = orc.get_corr(power, c['k'], c['s'], (0, 2, 4))
        assert np.allclose(corr, g['theory'][i], rtol=1e-10, atol=1e-12 * np.abs(g['theory'][i]).max())
        logl = orc.gaussian_loglikelihood(np.ravel(corr), c['flatdata'], g['precision'])[0]
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))
        assert np.isclose(orc.logprior(row, priors), g['logprior'][i], rtol=1e-13, atol=1e-13)


def make_likelihood(marg=False):
    from desilike_amd.emulators import EmulatedCalculator, TaylorEmulatorEngine
    from desilike_amd.theories.galaxy_clustering import REPTVelocileptorsTracerCorrelationFunctionMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg3_velocileptors_table_xi')
    c = g['obs0']
    engines = {name: TaylorEmulatorEngine(**state) for name, state in taylor_state(g).items()}
    specs = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
             'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}
    pt = EmulatedCalculator(EMU_PARAMS, engines, k=c['kpt'], ells=(0, 2, 4), z=0.8, param_specs=specs)
    theory = REPTVelocileptorsTracerCorrelationFunctionMultipoles(pt=pt, tracer='LRG')
    if marg:
        for name in ['alpha0p', 'alpha2p', 'alpha4p']:
            theory.init.params[name].update(derived='.marg')
    obs = TracerCorrelationFunctionMultipolesObservable(data=c['flatdata'], s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory)
    return g, ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])


@pytest.mark.gpu
def test_velocileptors_xi_call_surface_vs_reference():
    from desilike_amd import vmap
    g, like = make_likelihood()
    names = [str(n) for n in g['names']]
    assert like.varied_params.names() == names
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert errors == {}
    assert (np.abs(derived['loglikelihood'] - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all()
    assert np.allclose(derived['logprior'], g['logprior'], rtol=1e-13, atol=1e-13)
    flat = like._get_context().eval_batch_host(g['theta'], return_flattheory=True)[3]
    assert np.allclose(flat, g['flattheory'], rtol=1e-9, atol=1e-12 * np.abs(g['flattheory']).max())


@pytest.mark.gpu
def test_velocileptors_xi_marginalised_counterterms():
    g, like = make_likelihood(marg=True)
    c = g['obs0']
    names = [str(n) for n in g['names']]
    solved, vnames = like.solved_params.names(), like.varied_params.names()
    assert solved == ['alpha0p', 'alpha2p', 'alpha4p']
    sub = g['theta'][:, [names.index(n) for n in vnames]]
    loglike, logprior, status, xs = like._get_context().eval_batch_host(sub, return_solved=True)
    assert (status == 0).all()
    state = taylor_state(g)
    for i in range(0, len(sub), 3):
        row = dict(zip(vnames, sub[i]))

        def flat(x):
            full = np.array([row[n] if n in row else x[solved.index(n)] for n in names])
            return np.ravel(orc.get_corr(table_point(g, full, state), c['k'], c['s'], (0, 2, 4)))

        f0 = flat(np.zeros(3))
        T = np.array([flat(np.eye(3)[s]) - f0 for s in range(3)])
        sol = orc.solve_marginalized(f0 - c['flatdata'], T, like.precision, x0=np.zeros(3), prior_loc=np.zeros(3), prior_scale=np.full(3, 12.5), marg_mask=np.ones(3, dtype='?'))
        assert abs(loglike[i] - sol['loglikelihood']) <= 1e-8 * max(1., abs(sol['loglikelihood'])), (loglike[i], sol['loglikelihood'])
        assert np.allclose(xs[i], sol['x'], rtol=1e-6, atol=1e-8)
