"""CPU: the blocked Metropolis-Hastings sampler of the host package (desilike_amd/mcmc.py; reference desilike/samplers/mcmc.py) -- host driver against the oracle's
restatement of the reference (same counter-based draws -> same chain), the sampler surface on a toy likelihood, learning of the proposal covariance, checkpoints,
chains distributed over a gloo group of two ranks."""
import os

import numpy as np
import pytest

from oracle import np_oracle as orc
from test_samplers import ToyGaussianLikelihood
from test_oracle_mh import load, log_prob_fn


@pytest.mark.parametrize('name,vectorize,thin_by', [('mh_blocks', 3, 2), ('mh_single', 1, 1), ('mh_scalar_blocks', 2, 1)])
def test_host_driver_matches_the_oracle(name, vectorize, thin_by):
    from desilike_amd.mcmc import _HostMH
    g = load(name)
    fn = log_prob_fn(g)
    blocks, over = g['blocks'], g['oversample_factors']
    ndim = int(np.sum(blocks))
    order = np.random.RandomState(1).permutation(ndim)           # sorted position -> column
    inverse = np.argsort(order)

    def fn_columns(x):                                             # the likelihood sees its own column order
        return fn(np.atleast_2d(x)[:, order])

    seed, chain_ids, ntries = 0xfeedfacecafebeef, [4, 1], 300
    cov_sorted = g['proposal_cov']
    host = _HostMH(fn_columns, ndim, chain_ids, vectorize, blocks, over, order, 2.4, seed)
    host.set_covariance(np.linalg.cholesky(cov_sorted))
    starts_sorted = np.array([g['start'], g['start'] + 0.1 * np.sqrt(np.diag(g['cov']))])
    host.set_state(starts_sorted[:, inverse])
    records = host.run(ntries, thin_by=thin_by)
    state = host.get_state()
    for c, chain_id in enumerate(chain_ids):
        draws = orc.MHPhiloxDraws(seed, chain_id, blocks, over)
        chain, weight, logp, final = orc.mh_sample(fn, starts_sorted[c], draws, orc.mh_transforms(cov_sorted, blocks), ntries=ntries, thin_by=thin_by, vectorize=vectorize)
        assert len(weight) > 20
        assert np.array_equal(records[c][2], weight)
        assert np.allclose(records[c][0][:, order], chain, rtol=1e-12, atol=1e-14)
        assert np.allclose(records[c][1], logp, rtol=1e-12, atol=1e-12)
        assert np.allclose(state[0][c][order], final[0], rtol=1e-12, atol=1e-14) and state[2][c] == final[2]
    # two batches = one run
    again = _HostMH(fn_columns, ndim, chain_ids, vectorize, blocks, over, order, 2.4, seed)
    again.set_covariance(np.linalg.cholesky(cov_sorted))
    again.set_state(starts_sorted[:, inverse])
    first, second = again.run(120, thin_by=thin_by), again.run(180, thin_by=thin_by)
    for c in range(2):
        assert np.array_equal(np.concatenate([first[c][2], second[c][2]]), records[c][2])
        assert np.array_equal(np.concatenate([first[c][0], second[c][0]]), records[c][0])


def test_weighted_diagnostics_equal_the_expanded_chains():
    from desilike_amd import diagnostics as diag
    rng = np.random.RandomState(0)
    chains = [rng.standard_normal((200, 3)) + 0.1 * i for i in range(3)]
    weights = [rng.randint(1, 5, size=200) for _ in range(3)]
    expanded = [np.repeat(chain, weight, axis=0) for chain, weight in zip(chains, weights)]
    for method in ['eigen', 'diag']:
        assert np.allclose(diag.gelman_rubin(chains, method=method, weights=weights), diag.gelman_rubin(expanded, method=method), rtol=1e-12)
    # Geweke splits by samples, not by weight: compare with the explicit weighted formula
    gw = diag.geweke(chains, weights=weights)
    c, w = chains[0], weights[0]
    head, tail, wh, wt = c[:20], c[100:], w[:20], w[100:]
    expected = np.abs(np.average(head, weights=wh, axis=0) - np.average(tail, weights=wt, axis=0)) / (np.var(np.repeat(head, wh, axis=0), axis=0, ddof=1) + np.var(np.repeat(tail, wt, axis=0), axis=0, ddof=1))**0.5
    assert np.allclose(gw[:, 0], expected, rtol=1e-12)


def test_weighted_diagnostics_against_the_reference():
    """Gelman-Rubin / Geweke with frequency weights against the reference's own functions on chains of this sampler (tests/golden/validate_mh_chain.py)."""
    from desilike_amd import diagnostics as diag
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mh_weighted_diagnostics.npz')))
    x, w = [np.column_stack([g['a0'], g['b0']]), np.column_stack([g['a1'], g['b1']])], [g['w0'], g['w1']]
    assert np.allclose(diag.gelman_rubin(x, method='eigen', weights=w), g['eigen_gr'], rtol=1e-10)
    assert np.allclose(diag.gelman_rubin(x, method='diag', weights=w), g['diag_gr'], rtol=1e-10)
    assert np.allclose(diag.geweke(x, weights=w), g['geweke'], rtol=1e-10)


def test_sampler_recovers_the_toy_posterior(tmp_path):
    from desilike_amd.samplers import MCMCSampler
    like = ToyGaussianLikelihood()
    sampler = MCMCSampler(like, chains=4, vectorize=2, seed=5, covariance=np.diag([0.05, 0.05])**2, save_fn=str(tmp_path / 'chain_*.npy'))
    assert not sampler.device_resident and sampler.blocks == [2]
    chains = sampler.run(check_every=400, max_iterations=2400, check={'max_eigen_gr': 0.02, 'stable_over': 1}, min_iterations=800)
    assert len(chains) == 4 and all(set(chain) == {'a', 'b', 'fweight', 'logposterior'} for chain in chains)
    assert not np.allclose(sampler.covariance, np.diag([0.05, 0.05])**2)          # learnt from the chains
    assert 'eigen_gr' in sampler.diagnostics and sampler.diagnostics['eigen_gr'][-1] < 0.1
    x = np.concatenate([np.column_stack([chain['a'], chain['b']])[len(chain['a']) // 4:] for chain in chains])
    w = np.concatenate([chain['fweight'][len(chain['a']) // 4:] for chain in chains])
    mean = np.average(x, weights=w, axis=0)
    std = np.sqrt(np.average((x - mean)**2, weights=w, axis=0))
    assert np.allclose(mean, like.mean, atol=0.04)
    assert np.allclose(std, np.diag(like.cov)**0.5, rtol=0.15)
    assert np.all((sampler.acceptance_rate > 0.1) & (sampler.acceptance_rate < 0.8))
    # weights: every try is accounted for -- recorded weights + the weight of the current state + what the skipped start carried
    tries = sampler._tries
    for ichain, chain in enumerate(chains):
        assert chain['fweight'].sum() + sampler._state[ichain][2] <= tries * sampler.vectorize + 1
        assert chain['fweight'].min() >= 1
    # resume from the files: the continuation equals the uninterrupted run
    resumed = MCMCSampler(like, chains=[str(tmp_path / 'chain_{:d}.npy'.format(i)) for i in range(4)], seed=5, learn=False)
    assert resumed._tries == tries and resumed.counter_seed == sampler.counter_seed and resumed.vectorize == 2
    sampler.learn = False
    more_a = sampler.run(check_every=200, max_iterations=200)
    more_b = resumed.run(check_every=200, max_iterations=200)
    for a, b in zip(more_a, more_b):
        assert np.array_equal(a['a'], b['a']) and np.array_equal(a['fweight'], b['fweight'])


def test_blocks_and_arguments():
    from desilike_amd.samplers import MCMCSampler
    like = ToyGaussianLikelihood()
    sampler = MCMCSampler(like, blocks=[[3, ['a']], [1, ['b']]], seed=1)
    assert sampler.sorted_names == ['b', 'a'] and list(sampler.oversample_factors) == [1, 3] and list(sampler.order) == [1, 0] and sampler.blocks == [1, 1]
    # default covariance: the parameters' proposal scales
    assert np.allclose(sampler.covariance, np.diag([param.proposal**2 for param in like.varied_params]))
    chains = sampler.run(check_every=300, max_iterations=300)
    assert chains[0]['a'].size > 30
    with pytest.raises(ValueError): MCMCSampler(like, blocks=[[1, ['a']]])
    with pytest.raises(NotImplementedError): MCMCSampler(like, drag=True)
    with pytest.raises(ValueError): MCMCSampler(like, vectorize=100)
    with pytest.raises(np.linalg.LinAlgError): MCMCSampler(like, covariance=np.array([[1., 2.], [2., 1.]])).run(check_every=10, max_iterations=10)
    named = MCMCSampler(like, covariance=(['b'], np.array([[0.25]])), seed=2)
    assert named.covariance[1, 1] == 0.25 and named.covariance[0, 0] == like.varied_params[0].proposal**2
    # a start outside the prior has no finite log-posterior
    bad = MCMCSampler(like, seed=3)
    with pytest.raises(ValueError): bad.run(check_every=5, max_iterations=5, start=np.array([[9., 0.]]))


def _worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.samplers import MCMCSampler
    from desilike_amd.parallel import WalkerSharding
    like = ToyGaussianLikelihood()
    sampler = MCMCSampler(like, chains=3, vectorize=2, seed=9, sharding=WalkerSharding(min_shard_rows=0))
    assert sampler.chain_world == world and sampler.local_chains() == [c for c in range(3) if c % world == rank]
    chains = sampler.run(check_every=150, max_iterations=450, check={'max_eigen_gr': 1e-9})
    results[rank] = ([(chain['a'].copy(), chain['fweight'].copy()) for chain in chains], sampler.covariance.copy(), list(sampler.diagnostics['eigen_gr']), like.ncalls)
    dist.destroy_process_group()


def test_chains_over_two_ranks_equal_the_single_process_run():
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 35500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, results), nprocs=2, join=True)
    from desilike_amd.samplers import MCMCSampler
    single = MCMCSampler(ToyGaussianLikelihood(), chains=3, vectorize=2, seed=9)
    chains = single.run(check_every=150, max_iterations=450, check={'max_eigen_gr': 1e-9})
    for rank in range(2):
        got, cov, gr, ncalls = results[rank]
        for (a, w), chain in zip(got, chains):
            assert np.array_equal(a, chain['a']) and np.array_equal(w, chain['fweight'])      # every rank holds every chain; chains do not depend on the ranks
        assert np.allclose(cov, single.covariance, rtol=1e-12) and np.allclose(gr, single.diagnostics['eigen_gr'], rtol=1e-10)
    assert results[1][3] < results[0][3] or results[0][3] < 3 * 450 + 50                     # the ranks evaluated their own chains only


def test_learning_under_conditions():
    """``learn`` as a dictionary (mcmc.py:439-444, 467-483): the proposal covariance is updated only every so many samples and while Gelman-Rubin is inside the window."""
    from desilike_amd.samplers import MCMCSampler
    like = ToyGaussianLikelihood()
    start_cov = np.diag([0.3, 0.3])**2
    always = MCMCSampler(like, chains=3, vectorize=2, seed=6, covariance=start_cov, learn={'every': '5 * ndim', 'max_eigen_gr': 1e3, 'min_eigen_gr': -1., 'stable_over': 1})
    always.run(check_every=100, max_iterations=400)
    assert not np.allclose(always.covariance, start_cov) and 'eigen_gr' in always.learn_diagnostics
    never = MCMCSampler(like, chains=3, vectorize=2, seed=6, covariance=start_cov, learn={'max_eigen_gr': 1e-12, 'stable_over': 1})     # the window is never met
    never.run(check_every=100, max_iterations=400)
    assert np.allclose(never.covariance, start_cov)
    rare = MCMCSampler(like, chains=3, vectorize=2, seed=6, covariance=start_cov, learn={'every': 10**9, 'max_eigen_gr': 1e3, 'stable_over': 1})      # not enough new samples
    rare.run(check_every=100, max_iterations=300)
    assert np.allclose(rare.covariance, start_cov)
