"""Golden chains of the reference's own Metropolis-Hastings sampler (desilike/samplers/mcmc.py: MHSampler + BlockProposer), run here with the reference's classes on
an analytic log-posterior (a correlated Gaussian inside a box, -inf outside):

    python tests/golden/make_mh_fixture.py        (build container only; writes tests/golden/mh_*.npz)

The restatement in oracle/np_oracle.py (mh_sample with MHNumpyDraws) must reproduce chain, weights and log-posteriors bit for bit from the same seed.
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402, F401  (puts the stand-in packages and /root/reference on the path)

from desilike.samplers.mcmc import MHSampler, BlockProposer   # noqa: E402


def target(ndim, seed):
    rng = np.random.RandomState(seed)
    a = rng.standard_normal((ndim, ndim))
    cov = a.dot(a.T) / ndim + np.eye(ndim)
    scale = np.linspace(0.5, 2., ndim)
    cov = cov * scale[:, None] * scale[None, :]
    mean = rng.uniform(-1., 1., ndim)
    half_width = 2.5 * np.sqrt(np.diag(cov))
    return mean, cov, mean - half_width, mean + half_width


def log_prob(x, mean, cov, lo, hi):
    x = np.atleast_2d(x)
    d = x - mean
    toret = -0.5 * np.sum(d * np.linalg.solve(cov, d.T).T, axis=-1)
    toret[((x < lo) | (x > hi)).any(axis=-1)] = -np.inf
    return toret


def dump(name, blocks, oversample_factors, vectorize, iterations, thin_by, seed, proposal_scale=2.4):
    ndim = int(sum(blocks))
    mean, cov, lo, hi = target(ndim, seed)
    rng = np.random.RandomState(seed + 1)
    proposer = BlockProposer(blocks=blocks, oversample_factors=oversample_factors, proposal_scale=proposal_scale, rng=rng)
    proposal_cov = cov * rng.uniform(0.6, 1.4)
    proposer.set_covariance(proposal_cov)
    ncalls = [0]

    def fn(x):
        ncalls[0] += len(np.atleast_2d(x))
        return log_prob(x, mean, cov, lo, hi) if np.ndim(x) > 1 else log_prob(x, mean, cov, lo, hi)[0]

    sampler = MHSampler(ndim, fn, propose=proposer, vectorize=vectorize, rng=rng)
    start = mean + 0.3 * np.sqrt(np.diag(cov))
    for _ in sampler.sample(start, iterations=iterations, thin_by=thin_by): pass
    chain, weight, logp = sampler.get_chain(), sampler.get_weight(), sampler.get_log_prob()
    out = dict(blocks=np.array(blocks), oversample_factors=np.array(oversample_factors), vectorize=vectorize, iterations=iterations, thin_by=thin_by, seed=seed,
               proposal_scale=proposal_scale, mean=mean, cov=cov, lo=lo, hi=hi, proposal_cov=proposal_cov, start=start, chain=chain, weight=weight, logp=logp,
               final_coords=sampler.state.coords, final_logp=sampler.state.log_prob, final_weight=sampler.state.weight, acceptance_rate=sampler.get_acceptance_rate(), ncalls=ncalls[0])
    np.savez(os.path.join(here, name + '.npz'), **out)
    print(name, 'chain', chain.shape, 'acceptance {:.3f}'.format(sampler.get_acceptance_rate()), 'calls', ncalls[0], 'outside the box', int(np.isinf(logp).sum()))


if __name__ == '__main__':
    dump('mh_blocks', blocks=[3, 2], oversample_factors=[1, 2], vectorize=3, iterations=400, thin_by=2, seed=11)
    dump('mh_single', blocks=[6], oversample_factors=[1], vectorize=1, iterations=500, thin_by=1, seed=12)
    dump('mh_scalar_blocks', blocks=[1, 1], oversample_factors=[1, 3], vectorize=2, iterations=300, thin_by=1, seed=13)
