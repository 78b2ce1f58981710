"""Row a9, the general branch: priors other than uniform / norm (desilike/parameter.py:1958-1966, 2012-2016: scipy.stats frozen rv, value at loc removed).
CPU: the host class and the oracle against scipy; GPU: the device table (csrc/dl_prior.h) against the oracle through the C ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from test_host_api import make_cfg2

DISTS = ['expon', 'laplace', 'cauchy', 'logistic', 'halfnorm', 'halfcauchy', 'gumbel_r', 'gumbel_l']


def closed_form(dist, y):
    """The unnormalised log-densities of csrc/dl_prior.h, restated."""
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        return {'expon': np.where(y >= 0., -y, -np.inf), 'laplace': -np.abs(y), 'cauchy': -np.log1p(y**2),
                'logistic': -np.abs(y) - 2. * np.log1p(np.exp(-np.abs(y))) + 2. * np.log(2.), 'halfnorm': np.where(y >= 0., -0.5 * y**2, -np.inf),
                'halfcauchy': np.where(y >= 0., -np.log1p(y**2), -np.inf), 'gumbel_r': 1. - (y + np.exp(-y)), 'gumbel_l': y - np.exp(y) + 1.}[dist]


@pytest.mark.parametrize('dist', DISTS)
def test_host_prior_matches_scipy(dist):
    from scipy import stats
    from desilike_amd import ParameterPrior
    from desilike_amd.parameter import ParameterError, PRIOR_KINDS
    from oracle import np_oracle as orc
    loc, scale = 0.7, 1.3
    prior = ParameterPrior(dist=dist, loc=loc, scale=scale)
    x = np.concatenate([np.linspace(-4., 6., 41), [loc]])
    rv = getattr(stats, dist)(loc=loc, scale=scale)
    with np.errstate(divide='ignore'):
        ref = rv.logpdf(x) - rv.logpdf(loc)                      # parameter.py:2012-2016
    assert np.allclose(prior(x), ref, rtol=0, atol=1e-14, equal_nan=True) and prior(loc) == 0.
    assert np.allclose(orc.prior_logpdf(x, dist=dist, loc=loc, scale=scale), ref, rtol=0, atol=1e-14)
    mask = np.isfinite(ref)
    assert np.allclose(closed_form(dist, (x - loc) / scale)[mask], ref[mask], rtol=0, atol=1e-13) and np.array_equal(np.isneginf(closed_form(dist, (x - loc) / scale)), ~mask)
    assert prior.is_proper() and not prior.is_limited() and prior.spec() == [float(PRIOR_KINDS.index(dist)), -np.inf, np.inf, loc, scale]
    draws = prior.sample(size=2000, random_state=3)
    assert np.isfinite(prior(draws)).all() and abs(np.median(draws) - rv.median()) < 0.2 * scale
    assert ParameterPrior(**prior.__getstate__()).__getstate__() == prior.__getstate__()
    with pytest.raises(ParameterError):
        ParameterPrior(dist=dist, loc=0., limits=(0., 1.))       # the reference's trunc<dist> route is not reproduced
    with pytest.raises(ParameterError):
        ParameterPrior(dist=dist)                                 # no loc: the reference subtracts logpdf(nan)
    with pytest.raises(ParameterError):
        ParameterPrior(dist='gamma', a=2., loc=0.)


def test_solved_needs_norm_or_flat():
    from desilike_amd import Parameter
    with pytest.raises(Exception):
        Parameter('sn0', prior=dict(dist='laplace', loc=0., scale=1.), derived='.marg')   # parameter.py:762-764


@pytest.mark.gpu
def test_device_priors_match_oracle():
    from oracle import np_oracle as orc
    g, like = make_cfg2(dense=True)
    names = like.varied_params.names()
    # every family at once: the six sampled parameters of config 2 + two more through fixed -> varied
    like.all_params['sigmapar'].update(fixed=False, value=2., prior=dict(dist='halfnorm', loc=0., scale=4.), ref=dict(limits=[1., 3.]))
    like.all_params['sigmaper'].update(fixed=False, value=2., prior=dict(dist='halfcauchy', loc=0., scale=3.), ref=dict(limits=[1., 3.]))
    assign = {'qpar': ('laplace', 1., 0.05), 'qper': ('cauchy', 1., 0.04), 'dm': ('logistic', 0., 0.02), 'df': ('gumbel_r', 1., 0.1), 'b1': ('expon', 1., 1.5), 'sn0': ('gumbel_l', 0., 0.3)}
    for name, (dist, loc, scale) in assign.items():
        like.all_params[name].update(prior=dict(dist=dist, loc=loc, scale=scale), ref=dict(dist='norm', loc=like.all_params[name].value, scale=1e-3))
    varied = like.varied_params
    assert set(varied.names()) == set(names) | {'sigmapar', 'sigmaper'}
    rng = np.random.RandomState(7)
    B = 257
    theta = np.empty((B, len(varied)))
    for i, param in enumerate(varied):
        loc, scale = param.prior.loc, param.prior.scale
        theta[:, i] = loc + scale * rng.uniform(-0.5 if param.name in ('b1', 'sigmapar', 'sigmaper') else -3., 3., B)   # some rows below the support of the one-sided families
    theta[:, varied.names().index('qpar')] = np.clip(theta[:, varied.names().index('qpar')], 0.9, 1.1)
    theta[:, varied.names().index('qper')] = np.clip(theta[:, varied.names().index('qper')], 0.9, 1.1)
    ctx = like._get_context()
    loglike, logprior, status = ctx.eval_batch_host(theta)
    expected = orc.logprior(theta, [dict(dist=param.prior.dist, limits=param.prior.limits, loc=param.prior.loc, scale=param.prior.scale) for param in varied])
    out = np.isneginf(expected)
    assert out.any() and (~out).sum() > 50
    assert np.array_equal(np.isneginf(logprior), out) and np.array_equal(status == 1, out)
    assert np.allclose(logprior[~out], expected[~out], rtol=1e-13, atol=1e-12)
    # the samplers' entry point and the likelihood's own surface agree
    post, st = like._get_posterior_context()[0].eval_logposterior_host(theta)
    ok = status == 0
    assert np.array_equal(np.isneginf(post), ~ok) and np.allclose(post[ok], (loglike + logprior)[ok], rtol=1e-13, atol=1e-10)
    i0 = int(np.flatnonzero(ok)[0])
    value = like({name: theta[i0, i] for i, name in enumerate(varied.names())})
    assert np.isclose(value, loglike[i0] + logprior[i0], rtol=1e-13, atol=1e-10) and np.isclose(like.logprior, expected[i0], rtol=1e-13, atol=1e-12)
    # host class and device agree on the same numbers
    assert np.allclose(sum(param.prior(theta[~out, i]) for i, param in enumerate(varied)), logprior[~out], rtol=1e-13, atol=1e-12)
    # device-resident ensemble with these priors: finite chain, rows never leave the supports
    from desilike_amd.samplers import EmceeSampler
    sampler = EmceeSampler(like, nwalkers=32, seed=5)
    assert sampler.device_resident
    chain = sampler.run(niterations=20)
    assert np.isfinite(chain['logposterior']).all() and (chain['b1'] >= 1.).all() and (chain['sigmapar'] >= 0.).all()
