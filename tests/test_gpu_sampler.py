"""GPU (-m gpu): BASELINE config 5 geometry -- two tracers, joint covariance, walkers evaluated as one batch; sampler call surface."""
import os

import numpy as np
import pytest

from golden_utils import load_golden
from test_host_api import make_cfg5
from desilike_amd._lib import refresh_options as _refresh_options   # the library reads its DL_* switches once per process

pytestmark = pytest.mark.gpu


def test_two_tracers_vs_reference():
    from desilike_amd import vmap
    g, like = make_cfg5()
    rnames = [str(n) for n in g['names']]
    points = {name: g['theta'][:, i] for i, name in enumerate(rnames)}
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)(points)
    assert errors == {}
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(derived[like._param_loglikelihood] - g['loglikelihood']) <= tol).all()
    assert np.allclose(derived[like._param_logprior], g['logprior'], rtol=1e-13, atol=1e-13)
    like._evaluate_dict({name: np.atleast_1d(values[0]) for name, values in points.items()}, (), errors='return', return_flattheory=True)
    assert np.allclose(like.flattheory, g['flattheory'][0], rtol=1e-11, atol=1e-8)


def test_ensemble_sampler_on_gpu():
    from desilike_amd.samplers import EmceeSampler
    g, like = make_cfg5()
    sampler = EmceeSampler(like, nwalkers=64, seed=42, use_emcee=False, device_resident=False)   # host driver (NumPy stretch move) around the GPU likelihood
    chain = sampler.run(niterations=40)
    assert chain['logposterior'].shape == (40, 64) and np.isfinite(chain['logposterior']).all()
    # the ensemble climbs towards the posterior mode
    assert chain['logposterior'][-1].mean() > chain['logposterior'][0].mean()
    # fast path: torch tensors resident on the GPU give the same numbers as the dict surface
    import torch
    theta = np.column_stack([chain[param.name][-1] for param in like.varied_params])
    ll, lp, st = like.evaluate_batch(torch.as_tensor(theta, dtype=torch.float64, device='cuda:0'))
    torch.cuda.synchronize()
    assert np.allclose((ll + lp).cpu().numpy(), sampler.logposterior(theta), rtol=1e-12, atol=1e-10)


def test_grid_and_qmc_samplers_on_gpu():
    """samplers/grid.py, samplers/qmc.py: the whole grid / sequence is one batched evaluation; same numbers as the vmap surface, marginalised parameters attached."""
    from desilike_amd import vmap
    from desilike_amd.samplers import GridSampler, QMCSampler
    g, like = make_cfg5()
    names = like.varied_params.names()
    size = {name: 1 for name in names}
    size.update({'LRG.b1': 5, 'qpar': 3})
    samples = GridSampler(like, size=size).run()
    shape = tuple(size[name] for name in names)
    assert samples['LRG.b1'].shape == shape
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: samples[name] for name in names})
    assert errors == {} and np.array_equal(samples['loglikelihood'], derived['loglikelihood']) and np.array_equal(samples['logprior'], derived['logprior'])
    qmc = QMCSampler(like).run(niterations=64)
    assert qmc['loglikelihood'].shape == (64,) and np.isfinite(qmc['loglikelihood']).all()


def test_sampler_fast_path_matches_call_surface():
    """BasePosteriorSampler.logposterior through ONE dl_eval_logposterior_host call vs the vmap(likelihood) route (samplers/base.py:144-200 conventions: NaN rows, rows outside
    the prior, scalar input)."""
    from desilike_amd.samplers import BasePosteriorSampler
    g, like = make_cfg5()
    fast, slow = BasePosteriorSampler(like, seed=1), BasePosteriorSampler(like, seed=1)
    slow.fast = False
    rng = np.random.RandomState(3)
    values = np.column_stack([param.ref.sample(size=37, random_state=rng) for param in like.varied_params])
    values[3, 0] = np.nan
    values[5, 1] = 7.      # qper outside its prior
    a, b = fast.logposterior(values), slow.logposterior(values)
    assert np.isneginf(a[3]) and np.isneginf(a[5]) and np.isneginf(b[3]) and np.isneginf(b[5])
    ok = np.isfinite(b)
    assert ok.sum() == 35 and np.array_equal(np.isfinite(a), ok) and np.allclose(a[ok], b[ok], rtol=1e-13, atol=1e-10)
    assert np.isclose(fast.logposterior(values[0]), b[0], rtol=1e-13, atol=1e-10)


def test_device_resident_ensemble_matches_numpy_driver():
    """dl_ensemble_*: proposals, log-posterior, accept / reject and the counter-based generator on the device reproduce, bit for bit, the chain of the NumPy
    ``EnsembleStretchMove`` driven by the same generator (emcee's stretch move, samplers/emcee.py:69-111; conventions of samplers/base.py:144-200)."""
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    g, like = make_cfg5()
    nwalkers, niterations = 64, 30
    sampler = EmceeSampler(like, nwalkers=nwalkers, seed=42)
    assert sampler.device_resident
    start, logp0 = sampler._get_start(nwalkers)
    chain = sampler.run(niterations=niterations, start=start)
    chain = sampler.run(niterations=10)                                           # resumes from the device state
    assert chain['logposterior'].shape == (niterations + 10, nwalkers)
    host = EnsembleStretchMove(nwalkers, len(like.varied_params), sampler.logposterior, rng=CounterRNG(sampler.counter_seed))
    coords, logp = start.copy(), sampler.logposterior(start)
    for it in range(niterations + 10):
        coords, logp = host.step(coords, logp)
        got = np.column_stack([chain[param.name][it] for param in like.varied_params])
        assert np.array_equal(got, coords), it
        assert np.array_equal(chain['logposterior'][it], logp), it
    assert np.array_equal(sampler.acceptance_fraction, host.acceptance_fraction)
    assert 0.1 < sampler.acceptance_fraction.mean() < 0.9
    # thinning keeps every other ensemble
    again = EmceeSampler(like, nwalkers=nwalkers, seed=42)
    thin = again.run(niterations=5, thin_by=2, start=start)
    assert thin['logposterior'].shape == (5, nwalkers)


def test_device_ensemble_global_memory_variant():
    """The step kernel with the state in global memory (ensembles beyond the LDS of one CU; DL_ENS_GLOBAL=1 forces it) and the 1024-thread instantiation of the
    LDS kernel (more than 512 walkers) give the chains of the default kernel / of the NumPy driver, bit for bit."""
    import os
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    g, like = make_cfg5()
    nwalkers, niterations = 64, 12
    ref = EmceeSampler(like, nwalkers=nwalkers, seed=7)
    start, _ = ref._get_start(nwalkers)
    chain_ref = ref.run(niterations=niterations, start=start)
    os.environ['DL_ENS_GLOBAL'] = '1'; _refresh_options()
    try:
        other = EmceeSampler(like, nwalkers=nwalkers, seed=7)
        other._get_start(nwalkers)                                                 # (the device key is the generator's next draw: same history, same key)
        chain = other.run(niterations=niterations, start=start)
    finally:
        del os.environ['DL_ENS_GLOBAL']; _refresh_options()
    for name in chain_ref.keys() if hasattr(chain_ref, 'keys') else ['logposterior']:
        assert np.array_equal(np.asarray(chain[name]), np.asarray(chain_ref[name])), name
    assert np.array_equal(other.acceptance_fraction, ref.acceptance_fraction)
    # 640 walkers: the 1024-thread kernel; 1024 walkers: the same with the partial chi2 left in global memory (positions + proposals alone take 107 KB of LDS);
    # 2048 walkers: beyond the LDS, the global-memory kernel chosen by size -- each against the NumPy driver
    for nwalkers, niterations in [(640, 4), (1024, 3), (2048, 2)]:
        _check_against_numpy_driver(like, nwalkers, niterations)


def _check_against_numpy_driver(like, nwalkers, niterations):
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    big = EmceeSampler(like, nwalkers=nwalkers, seed=11)
    start, _ = big._get_start(nwalkers)
    chain = big.run(niterations=niterations, start=start)
    host = EnsembleStretchMove(nwalkers, len(like.varied_params), big.logposterior, rng=CounterRNG(big.counter_seed))
    coords, logp = start.copy(), big.logposterior(start)
    for it in range(niterations):
        coords, logp = host.step(coords, logp)
        assert np.array_equal(np.column_stack([chain[param.name][it] for param in like.varied_params]), coords), it
        assert np.array_equal(chain['logposterior'][it], logp), it


def test_device_ensemble_separate_finalize_path():
    """DL_ENS_NO_DEFER=1: the proposals' log-posteriors come from the finalize kernel (the path every rank of a sharded run takes: the all-gather ships
    log-posteriors) instead of being finished inside the step kernel from the partial chi2 -- same chain, bit for bit, at 64 and at 640 walkers."""
    from desilike_amd.samplers import EmceeSampler
    g, like = make_cfg5()
    for nwalkers, niterations in [(64, 10), (640, 3)]:
        ref = EmceeSampler(like, nwalkers=nwalkers, seed=9)
        start, _ = ref._get_start(nwalkers)
        chain_ref = ref.run(niterations=niterations, start=start)
        os.environ['DL_ENS_NO_DEFER'] = '1'; _refresh_options()
        try:
            other = EmceeSampler(like, nwalkers=nwalkers, seed=9)
            other._get_start(nwalkers)
            chain = other.run(niterations=niterations, start=start)
        finally:
            del os.environ['DL_ENS_NO_DEFER']; _refresh_options()
        for param in like.varied_params:
            assert np.array_equal(chain[param.name], chain_ref[param.name]), (nwalkers, param.name)
        assert np.array_equal(chain['logposterior'], chain_ref['logposterior'])
        assert np.array_equal(other.acceptance_fraction, ref.acceptance_fraction)


def test_device_ensemble_odd_shapes():
    """Ensemble shapes that are not multiples of anything: 7 parameters (config 2) x 18 or 22 walkers -- 9 / 11 proposals per half-step, an odd number of doubles in
    the proposal block and a last LDS-DMA chunk that is only half there -- against the NumPy driver, bit for bit; and the same on the two-tracer likelihood with 30
    walkers (15 proposals per half-step: one workgroup of the 16-row GEMM tile, partly filled)."""
    from test_host_api import make_cfg2
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    for like, nwalkers in [(make_cfg2()[1], 18), (make_cfg2()[1], 22), (make_cfg5()[1], 30)]:
        ndim, niterations = len(like.varied_params), 8
        assert nwalkers >= 2 * ndim
        sampler = EmceeSampler(like, nwalkers=nwalkers, seed=5)
        assert sampler.device_resident
        start, _ = sampler._get_start(nwalkers)
        chain = sampler.run(niterations=niterations, start=start)
        host = EnsembleStretchMove(nwalkers, ndim, sampler.logposterior, rng=CounterRNG(sampler.counter_seed))
        coords, logp = start.copy(), sampler.logposterior(start)
        for it in range(niterations):
            coords, logp = host.step(coords, logp)
            assert np.array_equal(np.column_stack([chain[param.name][it] for param in like.varied_params]), coords), (nwalkers, it)
            assert np.array_equal(chain['logposterior'][it], logp), (nwalkers, it)
        assert np.array_equal(sampler.acceptance_fraction, host.acceptance_fraction)


def test_device_ensemble_out_of_prior_and_nan_start():
    """Walkers proposed outside the prior are rejected (log-posterior -inf on the device); the chain never leaves the prior."""
    from desilike_amd.samplers import EmceeSampler
    g, like = make_cfg5()
    sampler = EmceeSampler(like, nwalkers=32, seed=3, ref_scale=4.)             # wide start: many proposals land outside qpar / qper limits
    chain = sampler.run(niterations=50)
    assert np.isfinite(chain['logposterior']).all()
    for param in like.varied_params:
        lo, hi = param.prior.limits
        assert (chain[param.name] >= lo).all() and (chain[param.name] <= hi).all()


def test_rccl_group_single_rank():
    """dl_comm_*: the RCCL binding of the C ABI with one rank (the multi-rank exchange needs one GPU per rank: the driver's multi-GPU run)."""
    import torch
    from desilike_amd.parallel import RcclGroup, WalkerSharding, BucketedAllGather
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29400 + os.getpid() % 500))
    group = RcclGroup(0, rank=0, world=1)
    assert group.rccl_version > 20000
    send = torch.arange(7, dtype=torch.float64, device='cuda:0')
    recv = torch.zeros(7, dtype=torch.float64, device='cuda:0')
    group.allgather_into(recv, send)
    group.barrier()
    assert torch.equal(recv, send)
    assert np.array_equal(group.broadcast(np.arange(5.)), np.arange(5.))
    assert group.max(3.5) == 3.5
    # sharded log-posterior through the device-resident exchange = the plain host call
    g, like = make_cfg5()
    ctx, offset = like._get_posterior_context()
    rng = np.random.RandomState(3)
    values = np.column_stack([param.ref.sample(size=37, random_state=rng) for param in like.varied_params])
    sharding = WalkerSharding(group=group, min_shard_rows=0)
    sharding.world = 1
    expected = ctx.eval_logposterior_host(values)[0] + offset
    assert np.array_equal(sharding.map_logposterior(ctx, values, offset=offset), expected)
    # the in-place path itself (world forced to look sharded)
    sharding.sharded = lambda size: True
    assert np.array_equal(sharding.map_logposterior(ctx, values, offset=offset), expected)
    # bucketed asynchronous all-gather on the side stream
    bucket = BucketedAllGather(5, torch.float64, torch.device('cuda:0'), steps_per_bucket=3, group=group, force_collective=True)
    for step in range(7):
        bucket.slot().copy_(torch.arange(5, dtype=torch.float64, device='cuda:0') + 10. * step)
        bucket.advance()
    out = [t.cpu().numpy() for t in bucket.results()]
    torch.cuda.synchronize()
    steps = sorted(int(row[0] // 10) for t in out for row in t[0] if not (row == 0.).all() or True)
    assert set(range(7)) <= set(steps)
    # the device-resident ensemble with the in-stream all-gather of the proposals' log-posteriors per half-step (DL_ENS_FORCE_COMM: the sharded path on one rank)
    # gives the chain of the single-rank path
    from desilike_amd.samplers import EmceeSampler
    ref = EmceeSampler(like, nwalkers=64, seed=13)
    start, _ = ref._get_start(64)
    chain_ref = ref.run(niterations=6, start=start)
    os.environ['DL_ENS_FORCE_COMM'] = '1'; _refresh_options()
    try:
        sharded = EmceeSampler(like, nwalkers=64, seed=13, sharding=WalkerSharding(group=group, min_shard_rows=0), device_resident=True)
        sharded._get_start(64)
        chain = sharded.run(niterations=6, start=start)
        assert sharded._get_ensemble().info('rows_per_rank') == 32
    finally:
        del os.environ['DL_ENS_FORCE_COMM']; _refresh_options()
    assert np.array_equal(chain['logposterior'], chain_ref['logposterior'])
    for param in like.varied_params: assert np.array_equal(chain[param.name], chain_ref[param.name])
    group.close()


def test_bench_two_ranks_on_one_gpu():
    """``python bench.py --gpus 2`` without a launcher starts its own ranks (host-side gloo exchange because both share this box's single GPU) and reports n_gpus = 2."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DL_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '16', '--warmup', '4', '--prewarm-ms', '50', '--config5-iterations', '4', '--chains-iterations', '6',
                          '--sustained-seconds', '0.2', '--no-other-configs', '--no-cpu-baseline'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = out.stdout.decode().strip().splitlines()
    line, complete = json.loads(lines[-1]), json.loads(lines[-2])     # the compact line the driver keeps (< 4 KB), and the complete record before it
    assert complete['record'] == 'complete' and complete['value'] == pytest.approx(line['value'], rel=1e-5) and len(lines[-1]) < 4096
    assert line['n_gpus'] == 2 and line['config']['ranks'] == 2 and line['value'] > 0.
    # the exchange checked end to end inside the run: every rank holds the same gathered log-posteriors, each rank's slice agrees with the oracle (bench.py::gathered_check)
    assert line['gathered_check']['ranks'] == 2 and line['gathered_check']['identical_on_every_rank'] and line['gathered_check']['max_rel_err_vs_oracle'] <= 1e-10
    assert line['config5_strong']['n_gpus'] == 2
    assert line['chains_weak']['n_gpus'] == 2 and [entry['chains'] for entry in line['chains_weak']['per_k']] == [2, 4, 8]      # chain-parallel sampler: K chains per rank
    assert complete['sustained']['steps'] >= 256


def test_config5_as_stated_device_vs_host_driver():
    """BASELINE configs[4] as written: 512 walkers on TWO config-2 tracers (dense 120 x 1200 windows, n = 240, block-diagonal precision: the chi2 GEMM skips the
    other tracer's K panels, one theory launch for both observables, the step kernel finishes the proposals itself).  The device-resident ensemble and the
    host-driven stretch move with the same counter-based generator give the same chain, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_likelihood_config5
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    like = make_likelihood_config5(0)
    assert like.size == 240 and len(like.varied_params) == 8
    ctx = like._get_context()
    assert ctx.info('N_pad') == 256 and ctx.info('K_pad') == 2432
    sampler = EmceeSampler(like, nwalkers=512, seed=42)
    start, logp0 = sampler._get_start(512)
    chain = sampler.run(niterations=12, start=start)
    host = EnsembleStretchMove(512, 8, sampler.logposterior, rng=CounterRNG(sampler.counter_seed))
    coords, logp = start.copy(), sampler.logposterior(start)
    for it in range(12):
        coords, logp = host.step(coords, logp)
        assert np.array_equal(np.column_stack([chain[param.name][it] for param in like.varied_params]), coords), it
        assert np.array_equal(chain['logposterior'][it], logp), it
    # the merged theory launch / panel skipping / aligned rows against the straightforward paths of the same library
    import torch
    theta = torch.as_tensor(coords, dtype=torch.float64, device='cuda:0').contiguous()
    ref = torch.empty(512, dtype=torch.float64, device='cuda:0')
    ctx.eval_logposterior(theta, ref)
    torch.cuda.synchronize()
    for flag in ['DL_NO_MERGED_THEORY', 'DL_NO_PANEL_SKIP', 'DL_NO_ROW_ALIGN']:
        os.environ[flag] = '1'; _refresh_options()
    try:
        from desilike_amd._lib import Context
        plain = Context(like._spec({}, like._flatdata_list(), like.precision), device=0)
        out = torch.empty(512, dtype=torch.float64, device='cuda:0')
        plain.eval_logposterior(theta, out)
        torch.cuda.synchronize()
        plain.close()
    finally:
        for flag in ['DL_NO_MERGED_THEORY', 'DL_NO_PANEL_SKIP', 'DL_NO_ROW_ALIGN']:
            del os.environ[flag]; _refresh_options()
    assert torch.allclose(out, ref, rtol=1e-13, atol=1e-10)
    # the chi2 GEMM's two row tiles (16 rows per workgroup below 129 workgroups of 32 rows: the 256-point half-steps; 32 otherwise): the partial sums of a row do
    # not depend on the tile, so the results are the same bit for bit
    half = theta[:256].contiguous()
    out16, out32 = torch.empty(256, dtype=torch.float64, device='cuda:0'), torch.empty(256, dtype=torch.float64, device='cuda:0')
    ctx.eval_logposterior(half, out16)
    torch.cuda.synchronize()
    os.environ['DL_CG_MT'] = '32'; _refresh_options()
    try:
        ctx.eval_logposterior(half, out32)
        torch.cuda.synchronize()
    finally:
        del os.environ['DL_CG_MT']; _refresh_options()
    assert torch.equal(out16, out32) and torch.equal(out16, ref[:256])


# ---- chains as the unit of parallelism: one device-resident ensemble per chain, several chains of a process on separate streams -------------------------------------
def test_two_chains_on_two_streams_are_the_single_chain_runs():
    """chains = 2 on ONE GPU (two ensembles, two HIP streams, enqueued back to back) == two single-chain runs with the same keys and starts, bit for bit."""
    from desilike_amd.samplers import EmceeSampler
    g, like = make_cfg5()
    nwalkers, niterations = 64, 25
    both = EmceeSampler(like, nwalkers=nwalkers, chains=2, seed=42)
    assert both.device_resident and both.chain_parallel and both.local_chains() == [0, 1]
    both._starts()
    starts = np.array([both._state[ichain][0] for ichain in range(2)])
    chains = both.run(niterations=niterations, check_every=10, check=False)          # three batches: 10 + 10 + 5
    assert both._runners[0].stream is not None and both._runners[0].stream.cuda_stream != both._runners[1].stream.cuda_stream
    assert chains[0]['logposterior'].shape == (niterations, nwalkers)
    assert not np.array_equal(chains[0]['qpar'], chains[1]['qpar'])
    for ichain in range(2):
        single = EmceeSampler(like, nwalkers=nwalkers, seed=1, counter_seeds=[both.counter_seeds[ichain]])
        chain = single.run(niterations=niterations, start=starts[ichain])
        for name in chain:
            assert np.array_equal(chain[name], chains[ichain][name]), (ichain, name)
        assert np.array_equal(single.acceptance_fraction, both.acceptance_fraction[ichain])
    assert both.check(max_eigen_gr=100.) in (True, False) and len(both.diagnostics['eigen_gr']) == 1


def test_device_resident_resume_from_checkpoint(tmp_path):
    """load() into a fresh sampler, then run() -- the new ensemble must take over positions, log-posteriors, key and counter of the file (it used to
    sample from uninitialised device memory): run -> save -> load -> run == the uninterrupted run."""
    from desilike_amd.samplers import EmceeSampler
    g, like = make_cfg5()
    nwalkers = 64
    full = EmceeSampler(like, nwalkers=nwalkers, seed=9)
    start, _ = full._get_start(nwalkers)
    full.run(niterations=35, start=start)
    first = EmceeSampler(like, nwalkers=nwalkers, seed=9, counter_seeds=[full.counter_seed])
    first.run(niterations=20, start=start)
    fn = str(tmp_path / 'chain.npz')
    first.save(fn)
    fresh = EmceeSampler(like, nwalkers=nwalkers, seed=1234)
    fresh.load(fn)
    chain = fresh.run(niterations=15)
    assert chain['logposterior'].shape == (35, nwalkers)
    for name in chain:
        assert np.array_equal(chain[name], full.chain[name]), name
    assert np.array_equal(fresh.acceptance_fraction, full.acceptance_fraction)
    # the same through the constructor (chains = files), and load() on a sampler whose ensemble already exists
    again = EmceeSampler(like, nwalkers=nwalkers, chains=[fn])
    assert np.array_equal(again.run(niterations=15)['qpar'], full.chain['qpar'])
    used = EmceeSampler(like, nwalkers=nwalkers, seed=3)
    used.run(niterations=4)
    used.load(fn)
    assert np.array_equal(used.run(niterations=15)['logposterior'], full.chain['logposterior'])


def test_device_ensemble_follows_parameter_changes():
    """a change of ``likelihood.all_params`` after the first run must reach the device-resident ensemble (it used to keep the old context)."""
    from desilike_amd.samplers import EmceeSampler
    g, like = make_cfg5()
    sampler = EmceeSampler(like, nwalkers=32, seed=5)
    sampler.run(niterations=5)
    like.all_params['LRG.b1'].update(prior=dict(dist='norm', loc=2., scale=0.01))
    chain = sampler.run(niterations=5)
    coords = np.column_stack([chain[param.name][-1] for param in like.varied_params])
    assert np.allclose(chain['logposterior'][-1], sampler.logposterior(coords), rtol=1e-12, atol=1e-9)   # (the host route re-reads the parameters at every call)


def test_importance_sampler(tmp_path):
    """desilike/samplers/importance.py: chains of one likelihood re-weighted by another (here: the same observable with twice the covariance), every chain point
    evaluated in one batch; the weights, the replaced loglikelihood / logprior, the attributes and the chain-file round trip."""
    from desilike_amd import vmap
    from desilike_amd.io import ChainFile
    from desilike_amd.samplers import EmceeSampler, ImportanceSampler
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood

    def make(scale):
        theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'))
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0., 0.2, 21), ells=(0, 2), wmatrix={'resolution': 2}, theory=theory, shotnoise=1e4)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=scale * 4e5 * np.eye(40))

    like_a, like_b = make(1.), make(2.)
    sampler = EmceeSampler(like_a, nwalkers=32, seed=3)
    sampler.run(niterations=12)
    chain = ChainFile.from_sampler(sampler)
    fn_in, fn_out = str(tmp_path / 'in.npz'), str(tmp_path / 'out_*.npy')
    chain.save(fn_in)
    with ImportanceSampler(like_b, fn_in, save_fn=fn_out) as importance:
        out, = importance.run(subtract_input=True)
    names = like_b.varied_params.names()
    points = {name: np.ravel(chain.arrays[name]) for name in names}
    (logpost_b, derived_b), errors = vmap(like_b, errors='return', return_derived=True)(points)
    assert errors == {}
    shape = chain.shape
    assert np.array_equal(out.arrays['loglikelihood'], np.asarray(derived_b['loglikelihood']).reshape(shape)) and np.array_equal(out.arrays['logprior'], np.asarray(derived_b['logprior']).reshape(shape))
    lp_a, lp_b = chain.arrays['logposterior'], np.asarray(logpost_b).reshape(shape)
    expected = np.exp(lp_b - lp_b[np.isfinite(lp_b)].max()) / np.exp(lp_a - lp_a[np.isfinite(lp_a)].max())
    assert np.allclose(out.arrays['aweight'], expected, rtol=1e-12, atol=0.)
    assert np.array_equal(out.arrays['logposterior'], lp_a)                       # the stored log-posterior stays the input's (as in the reference)
    assert out.attrs['size'] == 40 and out.attrs['nvaried'] == len(names) and out.attrs['ndof'] == 40 - len(names)
    back = ChainFile.load(fn_out.replace('*', '0'))
    assert np.array_equal(back.arrays['aweight'], out.arrays['aweight']) and sorted(back.arrays) == sorted(out.arrays)
    # halving the precision flattens the posterior: the weights grow away from the input chain's best point
    ibest = np.unravel_index(np.argmax(lp_a), shape)
    assert out.arrays['aweight'][ibest] <= np.median(out.arrays['aweight']) * 1.0000001 or out.arrays['aweight'].max() > out.arrays['aweight'][ibest]
