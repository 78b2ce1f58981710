"""CPU: sampler call surface and walker sharding (gloo, world_size 2) -- no GPU needed: a toy Gaussian likelihood object stands in.
Mirrors the reference's tests/test_samplers.py:9-52 (2-D Gaussian sampled with emcee; mean / std vs analytic)."""
import os

import numpy as np
import pytest

from desilike_amd import Parameter, ParameterCollection, Samples


class ToyGaussianLikelihood(object):
    """Minimal object with the likelihood surface the samplers use (varied_params, _param_*, _evaluate_dict)."""

    def __init__(self):
        self.varied_params = ParameterCollection([Parameter('a', prior=dict(limits=[-5., 5.]), ref=dict(limits=[-1., 1.])),
                                                  Parameter('b', prior=dict(dist='norm', loc=0., scale=10.), ref=dict(dist='norm', loc=0., scale=1.))])
        self.mean, self.cov = np.array([0.5, -0.3]), np.array([[0.04, 0.01], [0.01, 0.09]])
        self.precision = np.linalg.inv(self.cov)
        self._param_loglikelihood, self._param_logprior = Parameter('loglikelihood', derived=True), Parameter('logprior', derived=True)
        self.ncalls = 0

    def _evaluate_dict(self, flat, shape, errors='raise', return_derived=False):
        self.ncalls += 1
        x = np.column_stack([flat['a'], flat['b']]) - self.mean
        loglike = -0.5 * np.einsum('ij,jk,ik->i', x, self.precision, x)
        loglike[flat['a'] > 4.9] = np.nan                            # a NaN region, to exercise the NaN -> -inf convention
        logprior = sum(param.prior(flat[param.name]) for param in self.varied_params)
        derived = Samples()
        derived[self._param_loglikelihood], derived[self._param_logprior] = loglike.reshape(shape), logprior.reshape(shape)
        return ((loglike + logprior).reshape(shape), derived), {}


def test_logposterior_conventions():
    from desilike_amd.samplers import BasePosteriorSampler
    like = ToyGaussianLikelihood()
    sampler = BasePosteriorSampler(like, seed=1)
    values = np.array([[0.5, -0.3], [np.nan, 0.], [6., 0.], [4.95, 0.], [0., 0.]])
    lp = sampler.logposterior(values)
    assert np.isclose(lp[0], -0.5 * (0.3 / 10.)**2)          # likelihood maximum + Gaussian prior of b
    assert np.isneginf(lp[1]) and np.isneginf(lp[2]) and np.isneginf(lp[3]) and np.isfinite(lp[4])
    assert np.isclose(sampler.logposterior(values[0]), lp[0])   # 1-D input -> scalar
    calls = like.ncalls
    sampler.logposterior(np.array([[6., 0.], [-7., 1.]]))      # all rows outside the prior: the likelihood is not evaluated
    assert like.ncalls == calls


def test_stretch_move_recovers_gaussian():
    from desilike_amd.samplers import EmceeSampler
    like = ToyGaussianLikelihood()
    sampler = EmceeSampler(like, nwalkers=20, seed=42, use_emcee=False)
    sampler.run(niterations=400)
    chain = sampler.run(niterations=1600)
    assert chain['a'].shape == (2000, 20)
    samples = np.column_stack([chain[name][500:].ravel() for name in ['a', 'b']])
    assert np.allclose(samples.mean(axis=0), like.mean, atol=0.03)
    assert np.allclose(samples.std(axis=0), np.diag(like.cov)**0.5, rtol=0.1)
    assert 0.2 < sampler.acceptance_fraction.mean() < 0.9


def test_small_batches_are_not_sharded():
    """Below ``min_shard_rows`` every rank evaluates all rows (no exchange): the function sees the whole batch even when a process group is active."""
    from desilike_amd.parallel import WalkerSharding
    sharding = WalkerSharding()
    sharding.active, sharding.world, sharding.rank = True, 8, 3      # pretend: rank 3 of 8 (no collective may be called)
    seen = []
    values = np.arange(20.).reshape(10, 2)
    out = sharding.map(lambda rows: (seen.append(len(rows)), rows.sum(axis=1))[1], values)
    assert seen == [10] and np.array_equal(out, values.sum(axis=1))


def test_local_slice():
    from desilike_amd.parallel import local_slice
    for size in [0, 1, 7, 256, 1000]:
        for world in [1, 2, 3, 8]:
            slices = [local_slice(size, rank, world) for rank in range(world)]
            assert slices[0].start == 0 and slices[-1].stop == size
            assert all(a.stop == b.start for a, b in zip(slices[:-1], slices[1:]))


def _worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    like = ToyGaussianLikelihood()
    sharding = WalkerSharding(min_shard_rows=0)   # (the default keeps small batches un-sharded: here the exchange itself is under test)
    assert sharding.world == world and sharding.rank == rank
    rng = np.random.RandomState(3)
    values = rng.uniform(-1., 1., size=(13, 2))                  # ragged: 13 rows over 2 ranks
    values[4, 0] = np.nan
    sampler = EmceeSampler(like, nwalkers=12, seed=7, use_emcee=False, sharding=sharding)
    lp = sampler.logposterior(values)
    sampler.run(niterations=30)
    results[rank] = (lp, like.ncalls, sampler.chain['a'][-1].copy())
    dist.destroy_process_group()


def test_walker_sharding_gloo_world2():
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, results), nprocs=2, join=True)
    from desilike_amd.samplers import BasePosteriorSampler
    single = BasePosteriorSampler(ToyGaussianLikelihood())
    rng = np.random.RandomState(3)
    values = rng.uniform(-1., 1., size=(13, 2))
    values[4, 0] = np.nan
    expected = single.logposterior(values)
    for rank in range(2):
        assert np.allclose(results[rank][0], expected, equal_nan=True)     # every rank holds every walker's log-posterior
    assert np.allclose(results[0][2], results[1][2])                       # identical chains on all ranks


def _strong_worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    like = ToyGaussianLikelihood()
    sharding = WalkerSharding(min_shard_rows=0)                  # BASELINE configs[4] as written: every half-step is split over the ranks, whatever its size
    sampler = EmceeSampler(like, nwalkers=14, seed=11, use_emcee=False, sharding=sharding)   # half-steps of 7 proposals over 2 ranks: 4 + 3 rows (ragged)
    chain = sampler.run(niterations=40)
    results[rank] = (chain['a'].copy(), chain['b'].copy(), chain['logposterior'].copy(), like.ncalls, sampler.acceptance_fraction.copy())
    dist.destroy_process_group()


def test_strong_config5_path_ragged_walkers_gloo_world2():
    """The strong-scaling mode of BASELINE configs[4] (one ensemble, every half-step's proposals sharded over the ranks, log-posteriors all-gathered before the accept
    step) on two ranks with a walker count that does not divide: both ranks hold THE chain a single process produces from the same seed, and each rank evaluated
    only its share of the proposals."""
    import torch.multiprocessing as mp
    from desilike_amd.samplers import EmceeSampler
    manager = mp.Manager()
    results = manager.dict()
    port = 35500 + os.getpid() % 2000
    mp.spawn(_strong_worker, args=(2, port, results), nprocs=2, join=True)
    like = ToyGaussianLikelihood()
    single = EmceeSampler(like, nwalkers=14, seed=11, use_emcee=False)
    chain = single.run(niterations=40)
    for rank in range(2):
        a, b, logp, ncalls, acceptance = results[rank]
        assert np.array_equal(a, chain['a']) and np.array_equal(b, chain['b']) and np.array_equal(logp, chain['logposterior'])
        assert np.array_equal(acceptance, single.acceptance_fraction)
    assert results[0][3] > 0 and results[1][3] > 0


def _pipelined_worker(rank, world, port, results):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.parallel import PipelinedAllGather
    nloc = 5
    pipe = PipelinedAllGather((nloc,), torch.float64, 'cpu', nslots=2)
    out = []
    for step in range(6):   # two ensembles alternate: the gather of one is in flight while the other is "evaluated"
        slot = step % 2
        if pipe.pending(slot):
            out.append(pipe.result(slot).clone())
        local = torch.arange(nloc, dtype=torch.float64) + 100. * rank + 1000. * step
        pipe.submit(slot, local)
    for slot in [0, 1]:
        out.append(pipe.result(slot).clone())
    try:
        pipe.result(0)
        raised = False
    except RuntimeError:
        raised = True
    results[rank] = (torch.stack(out).numpy(), raised)
    dist.destroy_process_group()


def test_pipelined_allgather_gloo_world2():
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_pipelined_worker, args=(2, port, results), nprocs=2, join=True)
    expected = np.array([np.concatenate([np.arange(5) + 100. * rank + 1000. * step for rank in range(2)]) for step in range(6)])
    for rank in range(2):
        got, raised = results[rank]
        assert raised
        assert np.array_equal(got, expected)   # results come back in submission order, every rank sees every rank's rows


def _bucketed_worker(rank, world, port, results):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.parallel import BucketedAllGather
    nloc, k = 5, 3
    bucket = BucketedAllGather(nloc, torch.float64, 'cpu', steps_per_bucket=k)
    seen = []
    for step in range(8):   # 8 steps = 2 full buckets + one partial (flushed by results())
        out = bucket.slot()
        out.copy_(torch.arange(nloc, dtype=torch.float64) + 100. * rank + 1000. * step)
        bucket.advance()
        if step == 6:   # drain in the middle: everything submitted so far comes back
            seen.extend(t.clone() for t in bucket.results())
    seen.extend(t.clone() for t in bucket.results())
    results[rank] = [t.numpy() for t in seen]
    dist.destroy_process_group()


def test_bucketed_allgather_gloo_world2():
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 33500 + os.getpid() % 2000
    mp.spawn(_bucketed_worker, args=(2, port, results), nprocs=2, join=True)
    for rank in range(2):
        got = results[rank]
        assert all(g.shape == (2, 3, 5) for g in got)
        steps = sorted(set(int(v // 1000) for g in got for v in g.ravel() if v >= 0.))
        assert steps == list(range(8))   # every step's results arrived somewhere
        for g in got:
            for r in range(2):
                rows = g[r]
                live = [row for row in rows if not (row == 0.).all() or True]
                for row in rows:
                    step = int(row[0] // 1000)
                    if (row == 0.).all(): continue   # unused tail of the partial bucket
                    assert np.array_equal(row, np.arange(5) + 100. * r + 1000. * step)


class ToyWithFixed(ToyGaussianLikelihood):
    """+ ``all_params`` (one fixed parameter) for the grid / QMC samplers, which attach fixed parameters to their output (samplers/grid.py:112-113)."""

    def __init__(self):
        super(ToyWithFixed, self).__init__()
        self.all_params = ParameterCollection(list(self.varied_params) + [Parameter('c', value=3., fixed=True)])


def test_grid_sampler():
    from desilike_amd.samplers import GridSampler
    like = ToyWithFixed()
    sampler = GridSampler(like, size={'a': 5, 'b': 4}, ref_scale=0.5)
    samples = sampler.run()
    assert samples['a'].shape == (5, 4) and samples['loglikelihood'].shape == (5, 4)
    # 'a': uniform reference [-1, 1] around its centre 0, scaled by 0.5; 'b': value +- proposal (= reference std 1), even size drops the centre once (samplers/grid.py:80-93)
    assert np.allclose(sampler.grid[0], np.linspace(-0.5, 0.5, 5)) and np.allclose(sampler.grid[1], [-0.5, -0.25, 0.25, 0.5])
    x = np.stack([samples['a'], samples['b']], axis=-1) - like.mean
    assert np.allclose(samples['loglikelihood'], -0.5 * np.einsum('...j,jk,...k->...', x, like.precision, x))
    assert np.allclose(samples['logprior'], -0.5 * (samples['b'] / 10.)**2) and np.all(samples['c'] == 3.)
    assert like.ncalls == 1                                                      # the whole grid is ONE batched evaluation
    samples = sampler.run(grid={'a': [0.2, 0.1], 'b': [0.]})
    assert samples['a'].shape == (2, 1) and np.allclose(samples['a'][:, 0], [0.1, 0.2])


def test_qmc_sampler():
    from desilike_amd.samplers import QMCSampler, RQuasiRandomSequence
    seq = RQuasiRandomSequence(2).random(3)
    plastic = 1.32471795724474602596                                            # real root of x^3 = x + 1
    expected = (0.5 + np.arange(1, 4)[:, None] * np.array([1. / plastic, 1. / plastic**2])) % 1.
    assert np.allclose(seq, expected, rtol=0., atol=1e-12) and np.allclose(seq[0], [0.2548776662, 0.0698402910], atol=1e-9)
    like = ToyWithFixed()
    sampler = QMCSampler(like)
    samples = sampler.run(niterations=10)
    assert samples['a'].shape == (10,) and np.isfinite(samples['loglikelihood']).all()
    lower, upper = np.array([p.value - p.proposal for p in like.varied_params]), np.array([p.value + p.proposal for p in like.varied_params])
    unit = RQuasiRandomSequence(2).random(15)
    assert np.allclose(np.column_stack([samples['a'], samples['b']]), lower + unit[:10] * (upper - lower))
    samples = sampler.run(niterations=5)                                        # resumes after the 10 points already drawn
    assert samples['a'].shape == (15,) and np.allclose(np.column_stack([samples['a'], samples['b']]), lower + unit * (upper - lower))
    sobol = QMCSampler(like, engine='sobol', seed=3).run(niterations=8)
    assert sobol['a'].shape == (8,) and (np.abs(sobol['a']) <= upper[0]).all()


def test_counter_rng_known_answers():
    """Philox4x32-10 against the Random123 known-answer vectors; the derived draws."""
    from desilike_amd.samplers import CounterRNG

    def words(counter, key):
        return ['{:08x}'.format(int(v)) for v in CounterRNG.philox4x32(np.array([counter], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]]

    assert words([0, 0, 0, 0], [0, 0]) == ['6627e8d5', 'e169c58d', 'bc57ac4c', '9b00dbd8']
    assert words([0xffffffff] * 4, [0xffffffff] * 2) == ['408f276d', '41c83b0e', 'a20bc7c6', '6d5451fd']
    assert words([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == ['d16cfe09', '94fdcceb', '5001e420', '24126ea1']
    rng = CounterRNG(seed=123456789012345)
    perm = rng.permutation(5, 64)
    assert sorted(perm) == list(range(64)) and not np.array_equal(perm, rng.permutation(6, 64))
    # the split is a keyed bijection of [0, n) for every n (cycle-walking from the next power of two), and as a random split it is uniform: first-half membership,
    # co-membership of pairs and the walker in position 0, over 3000 iterations at n = 48, against the binomial / multinomial expectations
    for n in (2, 3, 5, 48, 100, 513):
        assert all(sorted(rng.permutation(it, n)) == list(range(n)) for it in range(20)), n
    n, T = 48, 3000
    first, co, pos0 = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    for it in range(T):
        p = rng.permutation(it, n)
        h = np.zeros(n); h[p[:n // 2]] = 1.
        first += h; co += np.outer(h, h); pos0[p[0]] += 1
    assert np.abs(first / T - 0.5).max() < 5. * np.sqrt(0.25 / T)
    pair = co[np.triu_indices(n, 1)] / T
    expected = (n / 2 - 1.) / (n - 1.) / 2.
    assert abs(pair.mean() - expected) < 1e-3 and np.abs(pair - expected).max() < 5. * np.sqrt(expected * (1. - expected) / T)
    assert ((pos0 - T / n)**2 / (T / n)).sum() / (n - 1) < 1.6       # chi2 / dof of the position-0 histogram
    u, partner = rng.move(5, 1, 32)
    assert ((u >= 0.) & (u < 1.)).all() and ((partner >= 0) & (partner < 32)).all()
    assert np.array_equal(u, CounterRNG(seed=123456789012345).move(5, 1, 32)[0])          # pure function of (seed, iteration, stream, index)
    big = np.concatenate([rng.accept(it, 0, 512) for it in range(40)])
    assert abs(big.mean() - 0.5) < 0.01 and abs(big.var() - 1. / 12.) < 0.005


def test_stretch_move_counter_rng_recovers_gaussian():
    from desilike_amd.samplers import BasePosteriorSampler, EnsembleStretchMove, CounterRNG
    like = ToyGaussianLikelihood()
    base = BasePosteriorSampler(like, seed=1)
    sampler = EnsembleStretchMove(20, 2, base.logposterior, rng=CounterRNG(7))
    start, logp = base._get_start(20)
    chain = []
    for it in range(2000):
        start, logp = sampler.step(start, logp)
        chain.append(start.copy())
    samples = np.concatenate(chain[500:])
    assert np.allclose(samples.mean(axis=0), like.mean, atol=0.03)
    assert np.allclose(samples.std(axis=0), np.diag(like.cov)**0.5, rtol=0.1)


def _unseeded_worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    like = ToyGaussianLikelihood()
    os.environ['DL_CHECK_SHARDING'] = '1'
    # no seed (rank 0's entropy is broadcast), then a user generator that differs per rank (rank 0's state is broadcast)
    chains = []
    for kwargs in [dict(), dict(rng=np.random.RandomState(100 + rank))]:
        sampler = EmceeSampler(like, nwalkers=12, use_emcee=False, sharding=WalkerSharding(min_shard_rows=0), **kwargs)
        sampler.run(niterations=10)
        chains.append(sampler.chain['a'].copy())
    results[rank] = chains
    dist.destroy_process_group()


def test_unseeded_samplers_share_one_random_stream_gloo_world2():
    """with a process group the ranks must propose the same walkers (seed / generator state broadcast from rank 0, samplers/base.py:213-217)."""
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 35500 + os.getpid() % 2000
    mp.spawn(_unseeded_worker, args=(2, port, results), nprocs=2, join=True)
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)


def test_bench_spawns_its_ranks():
    """``python bench.py --gpus 2`` with no launcher in the environment starts 2 ranks itself and reports n_gpus = 2 (--dry-run: the part that needs no GPU)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {key: value for key, value in os.environ.items() if key not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [line for line in out.stdout.decode().splitlines() if line.startswith('{')]
    assert len(lines) == 1                                                      # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['ranks'] == 2
    # under a launcher (WORLD_SIZE set) the process is one rank and spawns nothing
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--dry-run'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0 and json.loads(out.stdout.decode().strip().splitlines()[-1])['n_gpus'] == 1


def test_comm_bootstrap_under_torchrun_and_plain_launch(tmp_path):
    """The 128-byte RCCL id travels from rank 0 to the others over a TCP store: torchrun's own agent store (clients of MASTER_PORT) or, launched plainly
    (bench.py --gpus N), a store rank 0 hosts on MASTER_PORT + 1 -- the two ways the driver may start the ranks."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'exchange.py'
    script.write_text('''
import os, sys
sys.path.insert(0, {root!r})
from desilike_amd.parallel import _exchange_bytes
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
for it in range(2):
    payload = _exchange_bytes(bytes([it, 7, 9] * 40) if rank == 0 else None, rank, world)
    assert payload == bytes([it, 7, 9] * 40), (rank, payload)
# (in the product the ranks meet again in the collective dl_comm_create; here rank 0, which may host the store, must not leave before the others have read)
from desilike_amd.parallel import _stores
_stores[0].set('test/done/{{:d}}'.format(rank), '1')
_stores[0].wait(['test/done/{{:d}}'.format(r) for r in range(world)])
open(os.path.join({tmp!r}, 'ok_{{}}_{{:d}}'.format(os.environ.get('TORCHELASTIC_USE_AGENT_STORE'), rank)), 'w').write('ok')
'''.format(root=root, tmp=str(tmp_path)))
    env = {key: value for key, value in os.environ.items() if key not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_USE_AGENT_STORE')}
    def free_port():
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            return s.getsockname()[1]

    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(free_port()), str(script)],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and all((tmp_path / 'ok_True_{:d}'.format(rank)).exists() for rank in range(2)), out.stdout.decode()[-2000:]
    port, comm_port = free_port(), free_port()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DL_COMM_PORT=str(comm_port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for rank in range(2)]
    for rank, proc in enumerate(procs):
        text = proc.communicate(timeout=300)[0].decode()
        assert proc.returncode == 0 and (tmp_path / 'ok_None_{:d}'.format(rank)).exists(), text[-2000:]


# ---- chains as the unit of parallelism (samplers/base.py:409-502, utils.py:1040-1148) ---------------------------------------------------------------------------
def _chains_reference(nchains=3, niterations=40, nwalkers=12, seed=11):
    """Single process: ``nchains`` chains of the toy posterior through the host-driven counter-based stretch move."""
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    sampler = EmceeSampler(ToyGaussianLikelihood(), nwalkers=nwalkers, chains=nchains, seed=seed, use_emcee=False, sharding=WalkerSharding(group=False))
    sampler.run(niterations=niterations)
    return sampler


def test_chains_are_independent_and_reproducible():
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    sampler = _chains_reference()
    assert len(sampler.chains) == 3 and all(chain['a'].shape == (40, 12) for chain in sampler.chains)
    assert len(set(sampler.counter_seeds)) == 3
    assert not np.array_equal(sampler.chains[0]['a'], sampler.chains[1]['a'])
    # chain c == the single-chain run with chain c's key and start
    for ichain in [0, 2]:
        start = np.column_stack([sampler.chains[ichain][name][0] for name in ['a', 'b']])   # (the state after the first update: continue from there)
        single = EmceeSampler(ToyGaussianLikelihood(), nwalkers=12, chains=1, seed=99, use_emcee=False, counter_seeds=[sampler.counter_seeds[ichain]], sharding=WalkerSharding(group=False))
        single._state[0] = (start, sampler.chains[ichain]['logposterior'][0].copy())
        single._iterations[0] = 1
        single.run(niterations=39)
        for name in ['a', 'b', 'logposterior']:
            assert np.array_equal(single.chain[name], sampler.chains[ichain][name][1:]), (ichain, name)
    # batches of check_every updates give the same chains as one run
    batched = EmceeSampler(ToyGaussianLikelihood(), nwalkers=12, chains=3, seed=11, use_emcee=False, sharding=WalkerSharding(group=False))
    batched.run(niterations=40, check_every=15, check=False)
    for c0, c1 in zip(sampler.chains, batched.chains):
        assert all(np.array_equal(c0[name], c1[name]) for name in c0)
    assert np.array_equal(batched.acceptance_fraction, sampler.acceptance_fraction) and batched.acceptance_fraction.shape == (3, 12)


def test_chain_checkpoint_resume(tmp_path):
    """run -> save -> load into a NEW sampler -> run == the uninterrupted run (key, counter, positions and accepted counts travel in the file; reference layout)."""
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    from desilike_amd.io import ChainFile
    full = _chains_reference(nchains=2, niterations=30)
    first = EmceeSampler(ToyGaussianLikelihood(), nwalkers=12, chains=2, seed=11, use_emcee=False, sharding=WalkerSharding(group=False), save_fn=str(tmp_path / 'chain_*.npz'))
    first.run(niterations=18)
    saved = ChainFile.load(str(tmp_path / 'chain_1.npz'))
    assert saved.attrs['iteration'] == 18 and saved.attrs['counter_seed'] == first.counter_seeds[1] and saved.arrays['a'].shape == (18, 12)
    resumed = EmceeSampler(ToyGaussianLikelihood(), nwalkers=12, chains=[str(tmp_path / 'chain_0.npz'), str(tmp_path / 'chain_1.npz')], seed=5, use_emcee=False, sharding=WalkerSharding(group=False))
    resumed.run(niterations=12)
    for c0, c1 in zip(full.chains, resumed.chains):
        assert all(np.array_equal(c0[name], c1[name]) for name in c0)
    assert np.array_equal(resumed.acceptance_fraction, full.acceptance_fraction)
    # load() on an existing sampler that already ran something else
    other = EmceeSampler(ToyGaussianLikelihood(), nwalkers=12, chains=2, seed=77, use_emcee=False, sharding=WalkerSharding(group=False))
    other.run(niterations=3)
    other.load(str(tmp_path / 'chain_*.npz'))
    other.run(niterations=12)
    for c0, c1 in zip(full.chains, other.chains):
        assert all(np.array_equal(c0[name], c1[name]) for name in c0)


def test_convergence_check_stops_the_run():
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    sampler = EmceeSampler(ToyGaussianLikelihood(), nwalkers=16, chains=4, seed=3, use_emcee=False, sharding=WalkerSharding(group=False))
    sampler.run(min_iterations=200, max_iterations=4000, check_every=200, check={'max_eigen_gr': 0.05, 'stable_over': 2})
    niterations = sampler.chains[0]['a'].shape[0]
    assert 400 <= niterations < 4000 and niterations % 200 == 0            # stopped by the criterion, not by max_iterations
    d = sampler.diagnostics
    assert len(d['eigen_gr']) == niterations // 200 and all(d['eigen_gr_test'][-2:]) and d['eigen_gr'][-1] < 0.05
    assert set(['diag_gr', 'geweke', 'geweke_pvalue', 'iact', 'iterations_over_iact']) <= set(d)
    samples = np.column_stack([np.concatenate([chain[name][niterations // 2:].ravel() for chain in sampler.chains]) for name in ['a', 'b']])
    like = ToyGaussianLikelihood()
    assert np.allclose(samples.mean(axis=0), like.mean, atol=0.03) and np.allclose(samples.std(axis=0), np.diag(like.cov)**0.5, rtol=0.1)


def _chains_worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from desilike_amd.samplers import EmceeSampler
    like = ToyGaussianLikelihood()
    sampler = EmceeSampler(like, nwalkers=12, chains=3, seed=11 + 100 * rank, use_emcee=False)   # (different seeds on the ranks: rank 0's is broadcast)
    assert sampler.chain_world == world and sampler.local_chains() == ([0, 2] if rank == 0 else [1])
    sampler.run(niterations=40, check_every=15, check=False)
    converged = sampler.check(max_eigen_gr=10.)
    results[rank] = ([{name: value.copy() for name, value in chain.items()} for chain in sampler.chains], like.ncalls, sampler.counter_seeds, sampler.acceptance_fraction,
                     dict(sampler.diagnostics), converged)
    dist.destroy_process_group()


def test_chain_parallel_gloo_world2():
    """3 chains over 2 ranks (rank 0: chains 0 and 2, rank 1: chain 1): every rank ends up with all chains, bit-identical to the single-process run, having evaluated
    only its own; the convergence statistics agree on all ranks."""
    import torch.multiprocessing as mp
    manager = mp.Manager()
    results = manager.dict()
    port = 35500 + os.getpid() % 2000
    mp.spawn(_chains_worker, args=(2, port, results), nprocs=2, join=True)
    ref = _chains_reference()
    for rank in range(2):
        chains, ncalls, keys, accepted, diag, converged = results[rank]
        assert keys == ref.counter_seeds
        for c0, c1 in zip(ref.chains, chains):
            assert all(np.array_equal(c0[name], c1[name]) for name in c0)
        assert np.array_equal(accepted, ref.acceptance_fraction)
    assert results[0][1] > results[1][1]                       # rank 0 ran two chains, rank 1 one
    assert results[0][4]['eigen_gr'] == results[1][4]['eigen_gr'] and results[0][5] == results[1][5]
