"""Hamiltonian Monte Carlo on batches of chains (desilike_amd/hmc.py; the reference wraps blackjax.hmc, samplers/hmc.py): CPU tests on the toy Gaussian (finite-difference
gradient through the sampler's log-posterior), GPU tests with the analytic gradient of the device likelihood."""
import os

import numpy as np
import pytest

from test_samplers import ToyGaussianLikelihood


def test_hmc_recovers_the_toy_posterior(tmp_path):
    from desilike_amd.samplers import HMCSampler
    like = ToyGaussianLikelihood()
    sampler = HMCSampler(like, chains=64, seed=4, step_size=0.05, num_integration_steps=8, adaptation={'niterations': 150}, save_fn=str(tmp_path / 'hmc_*.npy'))
    chains = sampler.run(check_every=200, max_iterations=400, check={'max_eigen_gr': 0.05, 'stable_over': 1})
    assert len(chains) == 64 and chains[0]['a'].shape[0] in (200, 400)
    x = np.column_stack([np.concatenate([chain[name][50:] for chain in chains]) for name in ['a', 'b']])
    assert np.allclose(x.mean(axis=0), like.mean, atol=0.03)
    assert np.allclose(x.std(axis=0), np.diag(like.cov)**0.5, rtol=0.07)
    assert np.allclose(np.corrcoef(x.T)[0, 1], like.cov[0, 1] / np.sqrt(like.cov[0, 0] * like.cov[1, 1]), atol=0.07)
    # the warm-up found a step size that accepts most trajectories and a mass matrix of the posterior's scale
    assert 0.6 < sampler.acceptance_rate.mean() <= 1. and sampler.hyp['step_size'] > 0.05
    assert np.allclose(np.diag(sampler.hyp['inverse_mass_matrix']), np.diag(like.cov), rtol=0.6)
    assert sampler.diagnostics['eigen_gr'][-1] < 0.1 and (tmp_path / 'hmc_3.npy').exists()


def test_hmc_integrator_properties():
    """Without adaptation: a small step conserves the energy (every trajectory accepted), a huge one is flagged divergent and rejected; leaving the prior rejects."""
    import torch
    from desilike_amd.samplers import HMCSampler
    like = ToyGaussianLikelihood()
    sampler = HMCSampler(like, chains=8, seed=1, step_size=1e-3, num_integration_steps=5, adaptation=False, covariance=np.diag([0.04, 0.09]))
    sampler._generator = torch.Generator().manual_seed(3)
    q = torch.as_tensor(np.tile(like.mean, (8, 1)) + 0.05 * np.random.RandomState(0).standard_normal((8, 2)))
    lp, grad = sampler._value_and_grad(q)
    expected = -(q.numpy() - like.mean).dot(like.precision)
    expected[:, 1] -= q.numpy()[:, 1] / 100.                        # Gaussian prior of b (scale 10)
    assert np.allclose(grad.numpy(), expected, rtol=1e-5, atol=1e-7)
    minv, chol = sampler._mass(q.device)
    out = sampler._transition(q, lp, grad, 1e-3, minv, chol)
    assert bool(out[4].all()) and float(out[3].min()) > 0.999 and not bool(out[5].any())
    out = sampler._transition(q, lp, grad, 50., minv, chol)
    assert float(out[3].max()) < 1e-6 and torch.equal(out[0], q)
    # dense mass matrix: same target, momentum drawn with the Cholesky factor
    dense = HMCSampler(like, chains=8, seed=1, step_size=0.2, num_integration_steps=5, adaptation=False, covariance=like.cov)
    dense._generator = torch.Generator().manual_seed(3)
    minv, chol = dense._mass(q.device)
    assert minv.ndim == 2
    out = dense._transition(q, lp, grad, 0.2, minv, chol)
    assert float(out[3].mean()) > 0.9
    with pytest.raises(ValueError): HMCSampler(like, step_size=0.)
    with pytest.raises(ValueError): HMCSampler(like, gradient='jax')


@pytest.mark.gpu
def test_hmc_on_the_device_with_the_analytic_gradient():
    from golden_utils import load_golden
    from test_host_api import make_cfg2
    from desilike_amd.samplers import HMCSampler, EmceeSampler
    g, like = make_cfg2()
    names = like.varied_params.names()
    sampler = HMCSampler(like, chains=64, seed=2, num_integration_steps=12, adaptation={'niterations': 200}, gradient='analytic')
    chains = sampler.run(check_every=150, max_iterations=300)
    assert 0.5 < sampler.acceptance_rate.mean() <= 1.
    x = np.column_stack([np.concatenate([chain[name][100:] for chain in chains]) for name in names])
    ens = EmceeSampler(make_cfg2()[1], nwalkers=64, seed=3)
    chain = ens.run(niterations=1500)
    y = np.column_stack([chain[name][500:].ravel() for name in names])
    assert np.all(np.abs(x.mean(axis=0) - y.mean(axis=0)) < 0.25 * y.std(axis=0)), (x.mean(axis=0), y.mean(axis=0), y.std(axis=0))
    assert np.allclose(x.std(axis=0), y.std(axis=0), rtol=0.25)
    # central differences give the same trajectories to the accuracy of the differences
    finite = HMCSampler(make_cfg2()[1], chains=64, seed=2, num_integration_steps=12, adaptation=False, gradient='finite', step_size=sampler.step_size, covariance=sampler.inverse_mass_matrix)
    import torch
    q = torch.as_tensor(np.array([chain_[name][-1] for chain_ in chains for name in names]).reshape(64, len(names)), device='cuda:0')
    la, ga = sampler._value_and_grad(q)
    lf, gf = finite._value_and_grad(q)
    assert torch.allclose(la, lf, rtol=1e-12, atol=1e-9)
    scale = torch.as_tensor(np.sqrt(np.diag(sampler.inverse_mass_matrix)), device='cuda:0')
    assert float(((ga - gf) * scale).abs().max()) < 2e-2 * max(1., float((ga * scale).abs().max()))      # (the differences use the parameters' coarse `delta` steps)


def _worker(rank, world, port, results, nchains=6):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.samplers import HMCSampler
    from desilike_amd.parallel import WalkerSharding
    like = ToyGaussianLikelihood()
    sampler = HMCSampler(like, chains=nchains, seed=4, step_size=0.05, num_integration_steps=6, adaptation={'niterations': 60}, sharding=WalkerSharding(min_shard_rows=0))
    assert sampler.chain_world == world and sampler.local_chains() == [c for c in range(nchains) if c % world == rank]
    chains = sampler.run(check_every=80, max_iterations=160)
    chains = chains if isinstance(chains, list) else [chains]
    hyp = sampler.hyp
    results[rank] = (np.array([chain['a'] for chain in chains]), sampler.step_size, np.asarray(sampler.inverse_mass_matrix).copy(), sampler.acceptance_rate.copy(),
                     None if hyp is None else (hyp['step_size'], np.asarray(hyp['inverse_mass_matrix']).copy()))
    dist.destroy_process_group()


def test_hmc_chains_over_two_ranks():
    """Chains distributed over a gloo group of two: every rank ends with every chain, the same step size and mass matrix (averaged after the warm-up)."""
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 37500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, results), nprocs=2, join=True)
    a, b = results[0], results[1]
    assert a[0].shape == (6, 160) and np.array_equal(a[0], b[0])
    assert a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert 0.5 < a[3].mean() <= 1. and abs(a[0][:, 40:].mean() - 0.5) < 0.1


def test_hmc_fewer_chains_than_ranks():
    """One chain on two ranks: the rank without a chain adapts nothing, yet every rank ends with the chain and with the hyper-parameters the run USED (step size
    and mass matrix averaged over the ranks that have chains) -- also in ``hyp``, which ``save()`` writes into the chain attributes."""
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 39500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, results, 1), nprocs=2, join=True)
    a, b = results[0], results[1]
    assert a[0].shape == (1, 160) and np.array_equal(a[0], b[0])
    assert a[1] == b[1] and np.array_equal(a[2], b[2])
    for r in (a, b):
        assert r[4] is not None and r[4][0] == r[1] and np.array_equal(r[4][1], r[2])
