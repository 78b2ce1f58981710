"""GPU (-m gpu): BASELINE config 4 -- Damped-BAO xi_ell through the folded Hankel operator (and the P_ell version with a window),
against fixtures captured from the reference (running on the oracle's FFTLog: the reference's own transform is third-party, unpinned)
and against the NumPy oracle on a seeded batch; broadband parameters analytically marginalised."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, spec_from_golden_bao
from test_host_api import make_cfg4
from test_oracle_bao import bao_point

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('space', ['xi', 'pk'])
def test_bao_c_abi_vs_reference(space):
    from desilike_amd._lib import Context
    g = load_golden('cfg4_bao_' + space)
    ctx = Context(spec_from_golden_bao(g), device=0)
    power = ctx.eval_theory_host(g['theta'], iobs=0)
    assert np.allclose(power, g['wiggle_power'], rtol=1e-11, atol=1e-12 * np.abs(g['wiggle_power']).max())
    loglike, logprior, status, flat = ctx.eval_batch_host(g['theta'], return_flattheory=True)
    assert (status == 0).all()
    assert np.allclose(flat, g['flattheory'], rtol=1e-10, atol=1e-12 * np.abs(g['flattheory']).max())
    assert (np.abs(loglike - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all()
    assert np.allclose(logprior, g['logprior'], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize('space', ['xi', 'pk'])
def test_bao_call_surface_vs_reference(space):
    from desilike_amd import vmap
    g, like = make_cfg4(space)
    rnames = [str(n) for n in g['names']]
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(rnames)})
    assert errors == {}
    assert (np.abs(logpost - g['logposterior']) <= 1e-10 * np.maximum(1., np.abs(g['logposterior']))).all()
    # data generated from theory => likelihood(fiducial) == 0
    g2, like2 = make_cfg4(space, data={'b1': 2., 'sigmas': 2.})
    assert abs(like2(b1=2., sigmas=2.)) < 1e-12
    assert np.allclose(like2.observables[0].flatdata, g['obs0']['flatdata'], rtol=1e-9, atol=1e-12 * np.abs(g['obs0']['flatdata']).max())


def test_bao_xi_vs_oracle_seeded_and_marginalised_broadband():
    """8192-point batch property + oracle on a seeded sub-batch; then the 10 broadband parameters marginalised analytically."""
    g, like = make_cfg4('xi')
    c = g['obs0']
    names = like.varied_params.names()
    rng = np.random.RandomState(77)
    theta = np.column_stack([np.clip(param.ref.sample(size=8192, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert (status == 0).all() and np.isfinite(loglike).all()
    gfix = dict(g); gfix['names'] = np.array(names)
    for i in range(0, 8192, 512):
        power, broadband = bao_point(gfix, theta[i])
        flat = np.ravel(orc.get_corr(power, c['kin'], c['s'], (0, 2)) + broadband)
        ref = orc.gaussian_loglikelihood(flat, c['flatdata'], like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)
    # linearity in the broadband parameters (size-independent property): logL is exactly quadratic along al0_0
    ial = names.index('al0_0')
    line = np.repeat(theta[:1], 5, axis=0)
    line[:, ial] = np.linspace(-2e-3, 2e-3, 5)
    ll = ctx.eval_batch_host(line)[0]
    assert abs(np.diff(ll, 3)).max() < 1e-7 * abs(ll).max()
    # marginalise all broadband terms
    g3, like3 = make_cfg4('xi')
    like3.initialize()
    theory = like3.observables[0].wmatrix.theory
    for param in theory.init.params.select(basename='al*'):
        param.update(derived='.marg')
    like3._invalidate()
    assert len(like3.solved_params) == 10 and len(like3.varied_params) == len(names) - 10
    vnames = like3.varied_params.names()
    nsub = 64
    sub = theta[:nsub][:, [names.index(n) for n in vnames]]
    ctx3 = like3._get_context()
    ll3, lp3, st3, solved = ctx3.eval_batch_host(sub, return_solved=True)
    assert (st3 == 0).all()
    fold = theory._fold()
    nbb = fold.shape[1] - like3.observables[0].wmatrix.theory._hankel_block.shape[1]
    T = fold[:, -nbb:].T                                                   # d(flattheory) / d(al): constant
    worst = 0.
    for i in range(nsub):
        row = dict(zip(vnames, sub[i]))
        full = np.array([row.get(n, 0.) for n in names])
        power, broadband = bao_point(gfix, full)
        flat = np.ravel(orc.get_corr(power, c['kin'], c['s'], (0, 2)))
        sol = orc.solve_marginalized(flat - c['flatdata'], T, like3.precision, x0=np.zeros(nbb), prior_loc=np.zeros(nbb), prior_scale=np.full(nbb, np.inf), marg_mask=np.ones(nbb, dtype='?'))
        err = abs(ll3[i] - sol['loglikelihood']) / max(1., abs(sol['loglikelihood']))
        worst = max(worst, err)
        assert err <= 1e-10, (ll3[i], sol['loglikelihood'], err)     # north star: 1e-10 on logL, the marginalised value included
    print('cfg4 marginalised broadband, {:d} points: max relative error on logL {:.2e}'.format(nsub, worst))
