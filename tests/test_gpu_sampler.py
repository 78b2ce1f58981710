"""GPU (-m gpu): BASELINE config 5 geometry -- two tracers, joint covariance, walkers evaluated as one batch; sampler call surface."""
import numpy as np
import pytest

from golden_utils import load_golden
from test_host_api import make_cfg5

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
    sampler = EmceeSampler(like, nwalkers=64, seed=42, use_emcee=False)
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
