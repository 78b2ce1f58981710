"""The restatement of the reference's Metropolis-Hastings sampler (oracle/np_oracle.py: mh_sample, MHNumpyDraws; reference desilike/samplers/mcmc.py) against chains the
reference's own MHSampler + BlockProposer produced from the same seed (tests/golden/make_mh_fixture.py): bit for bit.  Then the counter-based draws of the device sampler
(MHPhiloxDraws = csrc/dl_mh.h): the distributions the reference draws from."""
import os

import numpy as np
import pytest

from oracle import np_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return dict(np.load(os.path.join(HERE, 'golden', name + '.npz')))


def log_prob_fn(g):
    mean, cov, lo, hi = g['mean'], g['cov'], g['lo'], g['hi']

    def fn(x):
        x = np.atleast_2d(x)
        d = x - mean
        toret = -0.5 * np.sum(d * np.linalg.solve(cov, d.T).T, axis=-1)
        toret[((x < lo) | (x > hi)).any(axis=-1)] = -np.inf
        return toret
    return fn


@pytest.mark.parametrize('name', ['mh_blocks', 'mh_single', 'mh_scalar_blocks'])
def test_restatement_reproduces_the_reference_chain(name):
    g = load(name)
    rng = np.random.RandomState(int(g['seed']) + 1)
    draws = orc.MHNumpyDraws(rng, g['blocks'], g['oversample_factors'])
    scale = rng.uniform(0.6, 1.4)      # (the generator drew the scale of the proposal covariance from the same stream)
    assert np.array_equal(g['cov'] * scale, g['proposal_cov'])
    transforms = orc.mh_transforms(g['proposal_cov'], g['blocks'])
    chain, weight, logp, final = orc.mh_sample(log_prob_fn(g), g['start'], draws, transforms, proposal_scale=float(g['proposal_scale']), iterations=int(g['iterations']),
                                               thin_by=int(g['thin_by']), vectorize=int(g['vectorize']))
    assert chain.shape == g['chain'].shape
    assert np.array_equal(chain, g['chain'])
    assert np.array_equal(weight, g['weight'])
    assert np.array_equal(logp, g['logp'])
    assert np.array_equal(final[0], g['final_coords']) and final[1] == float(g['final_logp']) and final[2] == int(g['final_weight'])


def test_counter_based_draws_follow_the_reference_distributions():
    blocks, over = [3, 2, 1], [1, 2, 3]
    draws = orc.MHPhiloxDraws(seed=0x1234567890abcdef, chain=3, blocks=blocks, oversample_factors=over)
    nrep = len(draws.rep_block)
    assert nrep == 3 + 4 + 3
    # every cycle visits every entry of the cycler once (mcmc.py:150-155)
    for q in range(5):
        assert sorted(draws.permutation(q)) == list(range(nrep))
    assert draws.permutation(0) != draws.permutation(1)
    # the columns drawn over one rotation of a block are orthonormal, with determinant + 1 (special_ortho_group)
    for b, ib in [(3, 0), (2, 1)]:
        q = np.array([draws.rotation_column(7, ib, b, j) for j in range(b)]).T
        assert np.allclose(q.T.dot(q), np.eye(b), atol=1e-14)
        assert np.isclose(np.linalg.det(q), 1., atol=1e-13)
    # calls of a block walk through the columns in order and a new rotation starts every b calls
    seen = {}
    for call in range(4 * nrep):
        ib, direction = draws.direction(call)
        seen.setdefault(ib, []).append(direction)
    assert [len(seen[ib]) for ib in range(3)] == [4 * 3, 4 * 4, 4 * 3]
    for ib, b in [(0, 3), (1, 2)]:
        d = np.array(seen[ib][:b])
        d /= np.linalg.norm(d, axis=1)[:, None]
        assert np.allclose(d.dot(d.T), np.eye(b), atol=1e-13)
    # radial mixture (mcmc.py:176-183) and the Metropolis exponentials: moments over many calls
    big = orc.MHPhiloxDraws(seed=5, chain=0, blocks=[4])
    radii = np.array([np.linalg.norm(big.direction(call)[1]) for call in range(6000)])
    expected_mean = 0.33 * 1. + 0.67 * np.sqrt(np.pi / 2.)             # exponential | Rayleigh
    assert abs(radii.mean() - expected_mean) < 4. * radii.std() / np.sqrt(radii.size)
    expo = np.array([big.exponential(call) for call in range(6000)])
    assert abs(expo.mean() - 1.) < 0.06 and abs(expo.var() - 1.) < 0.15
    one = orc.MHPhiloxDraws(seed=6, chain=1, blocks=[1])
    steps = np.array([one.direction(call)[1][0] for call in range(6000)])
    assert abs(np.mean(steps > 0) - 0.5) < 0.03
    expected_abs = 0.33 * 1. + 0.67 * np.sqrt(2. / np.pi)              # exponential | half-normal
    assert abs(np.abs(steps).mean() - expected_abs) < 0.04
    # directions are isotropic: mean of the outer products -> identity / b
    dirs = np.array([big.direction(call)[1] for call in range(4000)])
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    assert np.allclose(dirs.T.dot(dirs) / len(dirs), np.eye(4) / 4., atol=0.03)


def test_counter_based_chain_samples_the_target():
    g = load('mh_blocks')
    fn = log_prob_fn(g)
    transforms = orc.mh_transforms(g['proposal_cov'], g['blocks'])
    draws = orc.MHPhiloxDraws(seed=99, chain=0, blocks=g['blocks'], oversample_factors=g['oversample_factors'])
    chain, weight, logp, final = orc.mh_sample(fn, g['start'], draws, transforms, ntries=2500, vectorize=2)
    assert weight.sum() + final[2] > 2500 * 2 * 0.5
    mean = np.average(chain, weights=weight, axis=0)
    std = np.sqrt(np.diag(g['cov']))
    assert np.all(np.abs(mean - g['mean']) < 0.35 * std)
    rate = len(weight) / weight.sum()
    assert 0.1 < rate < 0.6
