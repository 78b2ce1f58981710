"""CPU: on-disk formats (desilike_amd/io.py).  The chain layout is pinned by the reference: tests/golden/validate_chain_io.py made the
reference load files written by ``ChainFile.save`` and run ``Chain.sample_solved`` (samples/chain.py:229-263) on them; its outputs and a chain file written by the
reference's own ``Chain.save`` are the fixtures used here."""
import importlib.util
import os

import numpy as np
import pytest

from desilike_amd import io

HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_chain():
    spec = importlib.util.spec_from_file_location('validate_chain_io', os.path.join(HERE, 'golden', 'validate_chain_io.py'))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module.synthetic_chain()


def test_hessian_packing():
    rng = np.random.RandomState(0)
    A = rng.standard_normal((5, 3, 3)); hessian = A + np.swapaxes(A, -1, -2)
    value = rng.standard_normal(5)
    packed = io.pack_hessian(value, hessian)
    assert packed.shape == (5, 7) and io.solved_derivs(['a', 'b', 'c']) == [(), ('a', 'a'), ('a', 'b'), ('a', 'c'), ('b', 'b'), ('b', 'c'), ('c', 'c')]
    assert np.array_equal(packed[:, 3], hessian[:, 0, 2]) and np.array_equal(packed[:, 4], hessian[:, 1, 1])
    v, h = io.unpack_hessian(packed, 3)
    assert np.array_equal(v, value) and np.array_equal(h, hessian)


def test_chain_written_here_is_what_the_reference_consumed(tmp_path):
    g = dict(np.load(os.path.join(HERE, 'golden', 'chain_io.npz'), allow_pickle=False))
    chain, solved = synthetic_chain()
    assert list(g['solved']) == solved
    for name, value in chain.arrays.items():
        assert np.array_equal(g['input.' + name], value), name          # the very arrays the reference loaded from our file
    for ext in ['npz', 'npy']:
        fn = str(tmp_path / ('chain.' + ext))
        chain.save(fn)
        if ext == 'npz':
            raw = dict(np.load(fn, allow_pickle=True))
            assert sorted(raw) == list(g['file_keys']) and sorted(raw['others'][()]) == list(g['others_keys'])
            assert sorted(raw['params'][()][0]['param']) == list(g['param_state_keys'])     # the reference's Parameter.__getstate__ key set
            assert tuple(raw['__class__']) == ('desilike.samples.chain.Chain',)
        back = io.ChainFile.load(fn)
        assert list(back.arrays) == list(chain.arrays) and back.derivs == chain.derivs and back.shape == chain.shape
        for name in chain.arrays:
            assert np.array_equal(back.arrays[name], chain.arrays[name])
        assert back.params['alpha0'].derived == '.marg' and back.params['LRG.b1'].namespace == 'LRG' and back.params['sn0'].prior.scale == 2.
    # what Chain.sample_solved made of the file (samples/chain.py:229-263), recomputed from the packed Hessians: the layout means what the reference reads in it
    ll, hl = io.unpack_hessian(chain.arrays['loglikelihood'], 2)
    lp, hp = io.unpack_hessian(chain.arrays['logprior'], 2)
    hessian = hl + hp
    covariance = np.linalg.inv(-hessian)
    covariance = 0.5 * (covariance + np.swapaxes(covariance, -1, -2))
    L = np.moveaxis(np.linalg.cholesky(covariance), (-2, -1), (0, 1))
    noise = np.random.RandomState(seed=42).standard_normal((2,) + chain.shape + (1,))
    values = np.sum(noise[None, ...] * L[..., None], axis=1)
    for ip, name in enumerate(solved):
        assert np.allclose(g['sample_solved.' + name], chain.arrays[name] + values[ip].reshape(chain.shape), rtol=1e-12, atol=1e-12)
    dlog = {}
    for key, base, hess in [('loglikelihood', ll, hl), ('logprior', lp, hp)]:
        quad = 0.5 * np.einsum('i...,...ij,j...->...', values[..., 0], hess, values[..., 0])
        dlog[key] = quad
        extra = 0.5 * np.linalg.slogdet(-hessian)[1] if key == 'loglikelihood' else 0.          # both solved parameters are marginalised ('.marg', '.auto')
        assert np.allclose(g['sample_solved.' + key], base + quad + extra, rtol=1e-12, atol=1e-10), key
    assert np.allclose(g['sample_solved.logposterior'], chain.arrays['logposterior'] + dlog['loglikelihood'] + dlog['logprior'] + 0.5 * np.linalg.slogdet(-hessian)[1],
                       rtol=1e-12, atol=1e-10)


def test_reader_loads_a_file_written_by_the_reference():
    chain, solved = synthetic_chain()
    back = io.ChainFile.load(os.path.join(HERE, 'golden', 'chain_reference_written.npz'))
    assert sorted(back.arrays) == sorted(chain.arrays) and back.shape == chain.shape
    for name in chain.arrays:
        assert np.array_equal(back.arrays[name], chain.arrays[name]), name
    assert back.derivs['loglikelihood'] == io.solved_derivs(solved) and back.derivs['logprior'] == io.solved_derivs(solved)
    assert back.params['alpha0'].solved and back.params['alpha0'].prior.scale == 12.5 and not back.params['qpar'].solved
    samples = back.to_samples()
    assert samples['loglikelihood'].shape == chain.shape and np.array_equal(samples['qpar'], chain.arrays['qpar'])


def test_window_and_data_containers(tmp_path):
    rng = np.random.RandomState(1)
    k = [np.linspace(0.01, 0.2, 20), np.linspace(0.01, 0.15, 15)]
    kin = np.linspace(0.001, 0.3, 60)
    matrix = rng.standard_normal((35, 3 * 60))
    fn = str(tmp_path / 'window.npz')
    io.save_window(fn, matrix, kin, (0, 2, 4), k, (0, 2), wshotnoise=np.r_[np.ones(20), np.zeros(15)])
    kw = io.load_window(fn)
    assert np.array_equal(kw['wmatrix'], matrix) and kw['ellsin'] == (0, 2, 4) and kw['ells'] == (0, 2) and np.array_equal(kw['k'][1], k[1]) and kw['wshotnoise'].sum() == 20.
    # the keyword arguments drive the window calculator as a dense matrix would
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import WindowedPowerSpectrumMultipoles
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'))
    window = WindowedPowerSpectrumMultipoles(theory=theory, shotnoise=1e3, **kw)
    window.initialize()
    assert window.matrix_full.shape == (35, 180) and window.size == 35 and np.array_equal(window.shotnoisein, np.zeros(3))
    # pypower BaseMatrix legacy state (.npy): value [n_in, n_out], one wide-angle projection dropped
    state = dict(value=np.vstack([matrix.T, rng.standard_normal((60, 35))]), xin=[kin] * 4, xout=k, projsin=[dict(ell=0, wa_order=0), dict(ell=2, wa_order=0), dict(ell=4, wa_order=0), dict(ell=1, wa_order=1)],
                 projsout=[dict(ell=0, wa_order=None), dict(ell=2, wa_order=None)], weightsin=None, weightsout=None, attrs={})
    fn = str(tmp_path / 'legacy.npy')
    np.save(fn, state, allow_pickle=True)
    kw2 = io.load_window(fn)
    assert np.array_equal(kw2['wmatrix'], matrix) and kw2['ellsin'] == (0, 2, 4) and kw2['ells'] == (0, 2)
    fn = str(tmp_path / 'data.npz')
    cov = np.eye(35)
    io.save_data(fn, k, (0, 2), rng.standard_normal(35), covariance=cov, shotnoise=1e3)
    kd = io.load_data(fn)
    assert kd['ells'] == (0, 2) and kd['data'].shape == (35,) and np.array_equal(kd['covariance'], cov) and kd['shotnoise'] == 1e3


def test_chain_statistics_with_weights():
    """ChainFile.mean / std / covariance / remove_burnin / concatenate with frequency weights: the values the reference's Chain gives for the chains of the
    Metropolis-Hastings sampler (tests/golden/validate_mh_chain.py checks mean and covariance against the reference itself) = statistics of the expanded samples."""
    import numpy as np
    from desilike_amd.io import ChainFile
    rng = np.random.RandomState(0)
    a, b, w = rng.standard_normal(300), rng.standard_normal(300) * 2. + 1., rng.randint(1, 6, size=300)
    chain = ChainFile({'a': a, 'b': b, 'fweight': w, 'logposterior': -0.5 * a**2})
    ea, eb = np.repeat(a, w), np.repeat(b, w)
    assert np.isclose(chain.mean('a'), ea.mean()) and np.isclose(chain.std('b'), eb.std(ddof=1))
    assert np.allclose(chain.covariance(['a', 'b']), np.cov(np.column_stack([ea, eb]), rowvar=False, ddof=1))
    cut = chain.remove_burnin(0.5)
    assert cut.shape == (150,) and np.isclose(cut.mean('a'), np.repeat(a[150:], w[150:]).mean())
    both = ChainFile.concatenate([cut, cut])
    assert both.shape == (300,) and np.isclose(both.mean('b'), cut.mean('b'))
    plain = ChainFile({'a': a.reshape(100, 3)})
    assert np.isclose(plain.mean('a'), a.mean()) and np.isclose(plain.std('a'), a.std(ddof=1))
