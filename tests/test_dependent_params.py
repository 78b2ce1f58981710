"""Parameters derived from others by an expression, ``derived='{a} * {b}'`` (desilike/parameter.py:758-807, 1872-1897; pipeline side base.py:533-545).
CPU: the parameter algebra and the host-side expansion of the device theta; GPU: the reference's own test of the feature (desilike/tests/test_base.py:158-165)."""
import numpy as np
import pytest

from test_host_api import make_cfg2


def test_parameter_expression():
    from desilike_amd import Parameter, ParameterCollection
    from desilike_amd.parameter import ParameterError
    param = Parameter('b1', derived='{b1s8} / {sigma8} + 0. * {b1s8}', prior=None)
    assert param.depends == ['b1s8', 'sigma8'] and param.varied and not param.solved and param.derived is not False
    assert np.allclose(param.eval(b1s8=np.array([1.6, 2.4]), sigma8=0.8), [2., 3.])
    with pytest.raises(ParameterError):
        param.eval(b1s8=1.)
    with pytest.raises(ParameterError):
        Parameter('x', derived='a + b')                       # neither a solved mode nor an expression of parameters
    assert Parameter('y', derived='np.sqrt({x})').eval(x=4.) == 2.
    params = ParameterCollection({'a': dict(prior=dict(limits=[0., 4.])), 'b': dict(value=3., fixed=True), 'c': dict(derived='{a} + {b}', prior=dict(limits=[0., 6.])),
                                  'd': dict(derived='2 * {c}')})
    assert params.eval(a=2., b=3.) == {'a': 2., 'b': 3., 'c': 5., 'd': 10.}     # parameter.py:1872-1887; d depends on a dependent
    assert params.prior(a=2., b=3.) == 0. and np.isneginf(params.prior(a=3.5, b=3.))   # c = 6.5 is outside its own prior (parameter.py:1894)
    assert params.names(varied=True, derived=False) == ['a']


def test_spec_and_expand():
    g, like = make_cfg2(dense=True)
    like.initialize()
    names0 = like.varied_params.names()
    spec0 = like._spec({}, like._flatdata_list(), like.precision)
    assert '_expand' not in spec0
    like.all_params['b1'].update(derived='{b}**2', prior=None)
    like.all_params['b'] = {'prior': {'limits': [0., 2.]}}
    names = like.varied_params.names()
    assert 'b1' not in names and names[-1] == 'b' and like.dependent_params.names() == ['b1']
    spec = like._spec({}, like._flatdata_list(), like.precision)
    # device columns: sampled parameters, then dependents; the theory reads b1 from its own column
    assert int(spec['n_params'][0]) == len(names) + 1 and spec['priors'].shape == (len(names) + 1, 5)
    assert spec['observables'][0]['inputs']['b1X'][0] == len(names)
    expand, nvaried = spec['_expand']
    assert nvaried == len(names)
    theta = np.random.RandomState(0).uniform(0.5, 1.5, (7, len(names)))
    full = expand(theta)
    assert full.shape == (7, len(names) + 1) and np.array_equal(full[:, :-1], theta) and np.allclose(full[:, -1], theta[:, -1]**2, rtol=0, atol=0)
    import torch
    assert np.array_equal(expand(torch.as_tensor(theta)).numpy(), full)
    # '_expand' is a host-side entry: it does not reach the C-ABI config
    from desilike_amd._lib import fill_config
    keys = []
    fill_config(spec, lambda key, array: keys.append(key), lambda key, array: keys.append(key))
    assert not any(key.startswith('_') for key in keys) and 'n_params' in keys
    # a dependent of a parameter nobody defines is an error when the context is compiled
    from desilike_amd.base import PipelineError
    like.all_params['b1'].update(derived='{nope}**2')
    with pytest.raises(PipelineError):
        like._spec({}, like._flatdata_list(), like.precision)
    # a fixed parameter enters expressions as a constant, overridden per call like any fixed parameter
    like.all_params['b1'].update(derived='{b} * {scale}')
    like.all_params['scale'] = dict(value=2., fixed=True)
    assert np.allclose(like._spec({}, like._flatdata_list(), like.precision)['_expand'][0](theta)[:, -1], 2. * theta[:, -1])
    assert np.allclose(like._spec({'scale': 3.}, like._flatdata_list(), like.precision)['_expand'][0](theta)[:, -1], 3. * theta[:, -1])
    del names0


@pytest.mark.gpu
def test_reference_identity():
    # desilike/tests/test_base.py:155-165: the likelihood in terms of b, b1 = b**2, equals the likelihood in terms of b1
    g, like = make_cfg2(dense=True)
    like.all_params['sn0'].update(derived='.marg')
    like(b1=1.5)
    bak = like.loglikelihood
    ctx0 = like._get_context()
    like.all_params['b1'].update(derived='{b}**2', prior=None)
    like.all_params['b'] = {'prior': {'limits': [0., 2.]}}
    assert 'b' in like.varied_params and 'b1' not in like.varied_params
    value, derived = like(b=1.5**0.5, return_derived=True)
    assert like._get_context() is not ctx0                       # the compiled context followed the change of parameters
    assert np.allclose(like.loglikelihood, bak, rtol=1e-12, atol=1e-10)
    assert np.allclose(derived['b1'], 1.5)                        # dependents are reported with the derived parameters (base.py:541-545)
    assert np.isneginf(like(b=2.5))                               # prior of the sampled parameter
    # batched surfaces: vmap, device tensors, samplers
    from desilike_amd import vmap
    import torch
    rng = np.random.RandomState(3)
    names = like.varied_params.names()
    theta = np.array([[rng.uniform(*np.clip(param.prior.limits, -10., 10.)) if param.prior.dist == 'uniform' else param.value for param in like.varied_params] for _ in range(33)])
    theta[:, names.index('b')] = rng.uniform(0.8, 1.6, 33)
    for name in ('qpar', 'qper', 'df', 'dm'):
        if name in names: theta[:, names.index(name)] = rng.uniform(0.95, 1.05, 33) if name != 'dm' else rng.uniform(-0.05, 0.05, 33)
    post = vmap(like)({name: theta[:, i] for i, name in enumerate(names)})
    dev = torch.as_tensor(theta, device='cuda')
    ll, lp = torch.empty(33, dtype=torch.float64, device='cuda'), torch.empty(33, dtype=torch.float64, device='cuda')
    like.evaluate_batch(dev, loglike=ll, logprior=lp)
    torch.cuda.synchronize()
    assert np.allclose((ll + lp).cpu().numpy(), post, rtol=1e-12, atol=1e-9)
    from desilike_amd.samplers import EmceeSampler
    sampler = EmceeSampler(like, nwalkers=16, seed=4)
    assert not sampler.device_resident                            # the device-resident ensemble does not see host-side expressions
    assert np.allclose(sampler.logposterior(theta), post, rtol=1e-12, atol=1e-9)
    # ... and all of it equals the likelihood in terms of b1 at b1 = b**2
    g2, ref = make_cfg2(dense=True)
    ref.all_params['sn0'].update(derived='.marg')
    names_ref = ref.varied_params.names()
    values = {name: theta[:, names.index(name)] for name in names_ref if name != 'b1'}
    values['b1'] = theta[:, names.index('b')]**2
    assert np.allclose(vmap(ref)(values), post, rtol=1e-12, atol=1e-9)
    chain = sampler.run(niterations=5)
    assert np.isfinite(chain['logposterior']).all() and chain['b'].shape == (5, 16)
