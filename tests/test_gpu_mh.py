"""GPU (-m gpu): the device-resident blocked Metropolis-Hastings sampler (dl_mh_*; reference desilike/samplers/mcmc.py) against the oracle's restatement of the
reference's MHSampler + BlockProposer (pinned bit for bit on chains of the reference itself, tests/test_oracle_mh.py) driven by the same counter-based draws."""
import numpy as np
import pytest

from test_host_api import make_cfg5

pytestmark = pytest.mark.gpu


def _setup(nchains, seed=3):
    g, like = make_cfg5()
    ctx, offset = like._get_posterior_context()
    P = ctx.n_params
    rng = np.random.RandomState(seed)
    center = np.array([param.value for param in like.varied_params], dtype='f8')
    sigma = np.array([0.02 * max(abs(v), 0.5) for v in center])
    a = rng.standard_normal((P, P))
    corr = a.dot(a.T) / P + 2. * np.eye(P)
    d = np.sqrt(np.diag(corr))
    cov = corr / d[:, None] / d[None, :] * sigma[:, None] * sigma[None, :]
    start = center + 0.5 * sigma * rng.standard_normal((nchains, P))
    return like, ctx, offset, cov, start


def _oracle_chain(ctx, offset, order, blocks, oversample, cov_sorted, start_ctx, seed, chain_id, ntries, vectorize, thin_by, scale=2.4):
    from oracle import np_oracle as orc
    order = np.asarray(order)

    def log_prob_fn(x_sorted):
        x_sorted = np.atleast_2d(x_sorted)
        x = np.empty_like(x_sorted)
        x[:, order] = x_sorted
        lp = ctx.eval_logposterior_host(x)[0]
        lp = np.where(np.isnan(lp), -np.inf, lp) + offset
        return lp

    draws = orc.MHPhiloxDraws(seed, chain_id, blocks, oversample)
    transforms = orc.mh_transforms(cov_sorted, blocks)
    chain, weight, logp, final = orc.mh_sample(log_prob_fn, start_ctx[order], draws, transforms, proposal_scale=scale, ntries=ntries, thin_by=thin_by, vectorize=vectorize)
    out = np.empty_like(chain)
    out[:, order] = chain
    fin = np.empty_like(final[0])
    fin[order] = final[0]
    return out, weight, logp, (fin, final[1], final[2])


@pytest.mark.parametrize('blocks,oversample,vectorize,thin_by', [([5, 3], [1, 2], 4, 2), ([8], [1], 1, 1), ([1, 6, 1], [1, 1, 3], 3, 1)])
def test_device_chains_match_the_oracle(blocks, oversample, vectorize, thin_by):
    import torch
    from desilike_amd._lib import DeviceMH
    nchains, ntries, seed = 3, 50, 0x9e3779b97f4a7c15
    like, ctx, offset, cov, start = _setup(nchains)
    P = ctx.n_params
    order = np.random.RandomState(5).permutation(P)
    cov_sorted = cov[np.ix_(order, order)]
    chain_ids = [7, 0, 12]
    mh = DeviceMH(ctx, nchains, vectorize=vectorize, blocks=blocks, oversample=oversample, order=order, chain_ids=chain_ids, seed=seed, offset=offset)
    assert mh.info('cycle') == int(np.sum(np.array(blocks) * np.array(oversample)))
    mh.set_covariance(np.linalg.cholesky(cov_sorted))
    mh.set_state(start)
    coords, logp, weight, count = mh.run(ntries, thin_by=thin_by)
    torch.cuda.synchronize()
    fcoords, flogp, fweight, fnacc, fails = mh.get_state()
    coords, logp, weight, count = coords.cpu().numpy(), logp.cpu().numpy(), weight.cpu().numpy(), count.cpu().numpy()
    assert mh.info('tries') == ntries
    for c in range(nchains):
        ochain, oweight, ologp, ofinal = _oracle_chain(ctx, offset, order, blocks, oversample, cov_sorted, start[c], seed, chain_ids[c], ntries, vectorize, thin_by)
        n = int(count[c])
        assert n == len(oweight) and n > 3, (n, len(oweight))
        assert np.array_equal(weight[c, :n], oweight)
        assert np.allclose(coords[c, :n], ochain, rtol=1e-11, atol=1e-13)
        assert np.allclose(logp[c, :n], ologp, rtol=1e-10, atol=1e-9)
        assert np.allclose(fcoords[c], ofinal[0], rtol=1e-11, atol=1e-13) and np.isclose(flogp[c], ofinal[1], rtol=1e-10, atol=1e-9) and fweight[c] == ofinal[2]
        assert fails[c] < 20
    # continuing the run is the same chain as one longer run (nothing but positions and counters is carried over)
    coords2, logp2, weight2, count2 = mh.run(20, thin_by=thin_by)
    torch.cuda.synchronize()
    c = 1
    ochain, oweight, ologp, ofinal = _oracle_chain(ctx, offset, order, blocks, oversample, cov_sorted, start[c], seed, chain_ids[c], ntries + 20, vectorize, thin_by)
    n1, n2 = int(count[c]), int(count2[c].item())
    assert n1 + n2 == len(oweight)
    assert np.array_equal(np.concatenate([weight[c, :n1], weight2[c, :n2].cpu().numpy()]), oweight)
    assert np.allclose(coords2[c, :n2].cpu().numpy(), ochain[n1:], rtol=1e-11, atol=1e-13)


def test_device_state_round_trip_and_errors():
    import torch
    from desilike_amd._lib import DeviceMH, LibraryError
    like, ctx, offset, cov, start = _setup(4)
    P = ctx.n_params
    L = np.linalg.cholesky(cov)
    a = DeviceMH(ctx, 4, vectorize=2, seed=11, offset=offset)
    a.set_covariance(L)
    a.set_state(start)
    ra = a.run(30)
    torch.cuda.synchronize()
    sa = a.get_state()
    # resume in a new object from (positions, log-posteriors, weights, counters) after 12 tries
    b = DeviceMH(ctx, 4, vectorize=2, seed=11, offset=offset)
    b.set_covariance(L)
    b.set_state(start)
    rb1 = b.run(12)
    torch.cuda.synchronize()
    coords, logp, weight, nacc, fails = b.get_state()
    c = DeviceMH(ctx, 4, vectorize=2, seed=11, offset=offset)
    c.set_covariance(L)
    c.set_state(coords, logposterior=logp, weight=weight, naccepted=nacc, tries=12)
    rc = c.run(18)
    torch.cuda.synchronize()
    sc = c.get_state()
    for x, y in zip(sa, sc): assert np.array_equal(x, y)
    for ich in range(4):
        n = int(ra[3][ich].item()); n1 = int(rb1[3][ich].item()); n2 = int(rc[3][ich].item())
        assert n == n1 + n2
        assert torch.equal(ra[0][ich, :n], torch.cat([rb1[0][ich, :n1], rc[0][ich, :n2]]))
        assert torch.equal(ra[2][ich, :n], torch.cat([rb1[2][ich, :n1], rc[2][ich, :n2]]))
    with pytest.raises(LibraryError): DeviceMH(ctx, 2, vectorize=65)
    with pytest.raises(LibraryError): DeviceMH(ctx, 2, blocks=[3, 3])
    with pytest.raises(LibraryError): DeviceMH(ctx, 2, order=np.zeros(P))
    d = DeviceMH(ctx, 2, seed=1)
    with pytest.raises(LibraryError): d.run(5)                 # no covariance
    with pytest.raises(LibraryError): d.set_covariance(L.T)     # upper-triangular
    d.set_covariance(L)
    bad = start[:2].copy(); bad[0, 1] = 1e3                      # outside the prior: no finite starting log-posterior
    d.set_state(bad)
    with pytest.raises(LibraryError): d.run(5)


def test_sampler_on_the_device_equals_the_host_driver():
    """MCMCSampler: the device-resident chains (dl_mh_*) and the host driver (NumPy proposals around the same GPU likelihood) give the same chains from the same seed."""
    from desilike_amd.samplers import MCMCSampler
    g, like = make_cfg5()
    names = like.varied_params.names()
    blocks = [[1, names[:5]], [2, names[5:]]]
    like2 = make_cfg5()[1]
    _, _, _, cov, start = _setup(3)
    kw = dict(blocks=blocks, chains=3, vectorize=4, seed=3, learn=False, covariance=cov)
    dev, host = MCMCSampler(like, **kw), MCMCSampler(like2, device_resident=False, **kw)
    assert dev.device_resident and not host.device_resident and dev.counter_seed == host.counter_seed
    cd = dev.run(check_every=30, max_iterations=60, start=start)
    ch = host.run(check_every=30, max_iterations=60, start=start)
    for a, b in zip(cd, ch):
        assert a['fweight'].size > 5 and np.array_equal(a['fweight'], b['fweight'])
        for name in names: assert np.allclose(a[name], b[name], rtol=1e-11, atol=1e-13)
        assert np.allclose(a['logposterior'], b['logposterior'], rtol=1e-10, atol=1e-9)


def test_sampler_learns_and_converges_on_the_device(tmp_path):
    from desilike_amd.samplers import MCMCSampler, EmceeSampler
    g, like = make_cfg5()
    names = like.varied_params.names()
    sampler = MCMCSampler(like, chains=8, seed=1, save_fn=str(tmp_path / 'mh_*.npy'))
    assert sampler.device_resident and sampler.vectorize == 32
    chains = sampler.run(check_every=300, min_iterations=600, max_iterations=3000, check={'max_eigen_gr': 0.05, 'stable_over': 1})
    assert sampler.diagnostics['eigen_gr'][-1] < 0.3
    rate = sampler.acceptance_rate
    assert np.all(rate > 0.02) and np.all(rate < 0.9)
    x = np.concatenate([np.column_stack([chain[name] for name in names])[chain['fweight'].size // 2:] for chain in chains])
    w = np.concatenate([chain['fweight'][chain['fweight'].size // 2:] for chain in chains])
    mean = np.average(x, weights=w, axis=0)
    std = np.sqrt(np.average((x - mean)**2, weights=w, axis=0))
    # the same posterior through the ensemble sampler
    ens = EmceeSampler(make_cfg5()[1], nwalkers=64, seed=2)
    chain = ens.run(niterations=1500)
    y = np.column_stack([chain[name][500:].ravel() for name in names])
    assert np.all(np.abs(mean - y.mean(axis=0)) < 0.35 * y.std(axis=0)), (mean, y.mean(axis=0), y.std(axis=0))
    assert np.allclose(std, y.std(axis=0), rtol=0.35)
    assert all(np.isfinite(chain['logposterior']).all() for chain in chains)
    assert (tmp_path / 'mh_0.npy').exists()


def _many_parameter_likelihood(ntemplates=30):
    """config 2 with ``ntemplates`` systematic templates as sampled pass-through parameters: 6 + ntemplates dimensions."""
    from golden_utils import load_golden
    from test_window_extras import KLIM
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg2_fc_syst')
    kedges = np.linspace(0., 0.2, 41)

    def template(i):
        return lambda ell, k: 2e2 * (ell == 2 * (i % 3)) / (1. + (k / (0.01 + 0.005 * i))**2)

    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, k=(kedges[:-1] + kedges[1:]) / 2., klim=KLIM, ells=(0, 2, 4),
                                                  wmatrix={'resolution': 2}, theory=theory, shotnoise=1e4, systematic_templates=[template(i) for i in range(ntemplates)])
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    for param in like.all_params.select(basename='syst_*'):
        param.update(prior=dict(dist='norm', loc=0., scale=2.), ref=dict(dist='norm', loc=0., scale=0.05), proposal=0.05)
    like._invalidate()
    return like


@pytest.mark.parametrize('split', [None, 20, 33])
def test_wide_blocks_on_the_device_equal_the_host_driver(split):
    """Blocks of more than 16 / more than 32 parameters (the rotation column from LDS rows in several passes / with the Gaussians drawn reflection by reflection)."""
    from desilike_amd.samplers import MCMCSampler
    names = _many_parameter_likelihood().varied_params.names()
    assert len(names) == 36
    blocks = None if split is None else [[1, names[:split]], [2, names[split:]]]
    kw = dict(blocks=blocks, chains=2, vectorize=3, seed=17, learn=False, proposal_scale=0.6)
    dev, host = MCMCSampler(_many_parameter_likelihood(), **kw), MCMCSampler(_many_parameter_likelihood(), device_resident=False, **kw)
    assert dev.device_resident and dev.blocks == ([36] if split is None else [split, 36 - split])
    start = dev._get_start(2)[0]
    cd, ch = dev.run(check_every=40, max_iterations=40, start=start), host.run(check_every=40, max_iterations=40, start=start)
    for a, b in zip(cd, ch):
        assert a['fweight'].size > 3 and np.array_equal(a['fweight'], b['fweight'])
        for name in names: assert np.allclose(a[name], b[name], rtol=1e-10, atol=1e-12), name
        assert np.allclose(a['logposterior'], b['logposterior'], rtol=1e-10, atol=1e-8)


def test_stream_groups_do_not_change_the_chains():
    """Groups of chains on HIP streams of their own (separate device contexts, concurrent tries): the chains are those of the single-stream run."""
    from desilike_amd.samplers import MCMCSampler
    _, _, _, cov, start = _setup(6)
    kw = dict(chains=6, vectorize=4, seed=21, learn=False, covariance=cov)
    one, three = MCMCSampler(make_cfg5()[1], streams=1, **kw), MCMCSampler(make_cfg5()[1], streams=3, **kw)
    assert one.streams == 1 and three.streams == 3
    ca = one.run(check_every=30, max_iterations=60, start=start)
    cb = three.run(check_every=30, max_iterations=60, start=start)
    for a, b in zip(ca, cb):
        assert a['fweight'].size > 5 and np.array_equal(a['fweight'], b['fweight']) and np.array_equal(a['logposterior'], b['logposterior'])
        assert np.array_equal(a['qpar'], b['qpar'])
    assert [state[3] for state in one._state] == [state[3] for state in three._state]


def _few_parameter_likelihood(keep):
    g, like = make_cfg5()
    like.all_params = {param.name: {'fixed': True} for param in like.varied_params if param.name not in keep}
    return like


@pytest.mark.parametrize('keep,blocks,vectorize', [(['qpar'], None, 64), (['qpar', 'qper'], [[1, ['qpar']], [4, ['qper']]], 5), (['qpar', 'qper', 'df'], None, 1)])
def test_edge_shapes_on_the_device_equal_the_host_driver(keep, blocks, vectorize):
    """One parameter (a block of one: sign x radius, no rotation), two one-parameter blocks with oversampling (the cycler keeps its order for two entries or fewer; here five
    entries), the largest number of speculative proposals, a single chain."""
    from desilike_amd.samplers import MCMCSampler
    kw = dict(blocks=blocks, chains=1, vectorize=vectorize, seed=5, learn=False)
    dev, host = MCMCSampler(_few_parameter_likelihood(keep), **kw), MCMCSampler(_few_parameter_likelihood(keep), device_resident=False, **kw)
    assert dev.device_resident and len(dev.varied_params) == len(keep)
    start = dev._get_start(1)[0]
    cd, ch = dev.run(check_every=40, max_iterations=80, start=start), host.run(check_every=40, max_iterations=80, start=start)
    a, b = cd[0], ch[0]
    assert a['fweight'].size > 5 and np.array_equal(a['fweight'], b['fweight'])
    for name in keep: assert np.allclose(a[name], b[name], rtol=1e-11, atol=1e-13)
    assert np.allclose(a['logposterior'], b['logposterior'], rtol=1e-10, atol=1e-9)


def test_sampler_resumes_device_chains_from_files(tmp_path):
    """save -> a NEW sampler built from the files -> run: the continuation is what the uninterrupted run gives (positions, weights, counters and the key of the
    counter-based draws travel in the files; the device sampler receives them before its first try)."""
    from desilike_amd.samplers import MCMCSampler
    _, _, _, cov, start = _setup(3)
    kw = dict(vectorize=4, seed=8, learn=False, covariance=cov)
    a = MCMCSampler(make_cfg5()[1], chains=3, save_fn=str(tmp_path / 'mh_*.npy'), **kw)
    a.run(check_every=40, max_iterations=40, start=start)
    b = MCMCSampler(make_cfg5()[1], chains=[str(tmp_path / 'mh_{:d}.npy'.format(i)) for i in range(3)], learn=False)
    assert b.device_resident and b._tries == 40 and b.counter_seed == a.counter_seed and np.allclose(b.covariance, cov)
    a.save_fn = None
    ca, cb = a.run(check_every=30, max_iterations=30), b.run(check_every=30, max_iterations=30)
    for x, y in zip(ca, cb):
        assert x['fweight'].size > 8 and np.array_equal(x['fweight'], y['fweight']) and np.array_equal(x['qpar'], y['qpar']) and np.array_equal(x['logposterior'], y['logposterior'])


def test_marginalised_likelihood_on_the_device_equals_the_host_driver():
    """Analytically marginalised shot-noise terms: the device chains sample the marginalised posterior (the constant of the marginalisation travels as the offset)."""
    from desilike_amd.samplers import MCMCSampler

    def build():
        g, like = make_cfg5()
        like.all_params = {'*.sn0': {'derived': '.marg'}}
        return like

    a = build()
    assert len(a.solved_params) == 2 and len(a.varied_params) == 6
    kw = dict(chains=2, vectorize=3, seed=12, learn=False)
    dev, host = MCMCSampler(a, **kw), MCMCSampler(build(), device_resident=False, **kw)
    assert dev.device_resident
    start = dev._get_start(2)[0]
    cd, ch = dev.run(check_every=40, max_iterations=40, start=start), host.run(check_every=40, max_iterations=40, start=start)
    for x, y in zip(cd, ch):
        assert x['fweight'].size > 4 and np.array_equal(x['fweight'], y['fweight'])
        assert np.allclose(x['logposterior'], y['logposterior'], rtol=1e-10, atol=1e-8)
        assert np.allclose(x['qpar'], y['qpar'], rtol=1e-11, atol=1e-13)
    # the recorded log-posteriors are those of the sampler's own call surface
    names = a.varied_params.names()
    points = np.column_stack([cd[0][name] for name in names])
    assert np.allclose(dev.logposterior(points), cd[0]['logposterior'], rtol=1e-10, atol=1e-8)


def test_a_thousand_chains_sample_the_same_posterior_as_the_ensemble_sampler():
    """Full-size batch (1024 chains = 1024 rows per try): after burn-in the weighted ensemble of the chains' states has the moments the stretch-move sampler finds."""
    from desilike_amd.samplers import MCMCSampler, EmceeSampler
    g, like = make_cfg5()
    names = like.varied_params.names()
    sampler = MCMCSampler(like, chains=1024, vectorize=1, seed=31)
    sampler.run(check_every=250, max_iterations=2500)            # the proposal covariance is learnt from the pooled chains after each batch
    chains = sampler.chains
    x = np.concatenate([np.column_stack([chain[name] for name in names])[chain['fweight'].size // 2:] for chain in chains if chain is not None])
    w = np.concatenate([chain['fweight'][chain['fweight'].size // 2:] for chain in chains if chain is not None])
    assert x.shape[0] > 20000
    mean = np.average(x, weights=w, axis=0)
    std = np.sqrt(np.average((x - mean)**2, weights=w, axis=0))
    ens = EmceeSampler(make_cfg5()[1], nwalkers=64, seed=2)
    chain = ens.run(niterations=1500)
    y = np.column_stack([chain[name][500:].ravel() for name in names])
    assert np.all(np.abs(mean - y.mean(axis=0)) < 0.2 * y.std(axis=0)), (mean - y.mean(axis=0)) / y.std(axis=0)
    assert np.allclose(std, y.std(axis=0), rtol=0.15), std / y.std(axis=0)
    assert 0.1 < np.nanmean(sampler.acceptance_rate) < 0.6
