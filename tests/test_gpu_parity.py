"""GPU (-m gpu): the HIP path, called through the C ABI, against golden vectors captured from the reference
and against the NumPy oracle on seeded inputs.  Tolerances: 1e-10 on logL (north star; relative above |logL| = 1),
1e-11 relative on intermediates."""
import numpy as np
import pytest

from golden_utils import load_golden, spec_from_golden, observable_constants, prior_list

pytestmark = pytest.mark.gpu

GOLDEN = ['cfg1_kaiser_nowindow', 'cfg2_shapefit_window', 'cfg2_shapefit_window_dense', 'cfg2v_eft_damping_qisoqap']


@pytest.fixture(scope='module')
def contexts():
    from desilike_amd._lib import Context
    cache = {}

    def get(name):
        if name not in cache:
            g = load_golden(name)
            cache[name] = (g, Context(spec_from_golden(g), device=0))
        return cache[name]

    yield get
    for g, ctx in cache.values():
        ctx.close()


@pytest.mark.parametrize('name', GOLDEN)
def test_theory_tables_vs_reference(contexts, name):
    g, ctx = contexts(name)
    nint = g['int_power'].shape[0]
    power, tables = ctx.eval_theory_host(g['theta'][:nint], iobs=0, return_tables=True)
    for i, key in enumerate(['pk_dd', 'pk_dt', 'pk_tt']):
        ref = g['int_' + key][:, 0]
        assert np.allclose(tables[:, i], ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max()), key
    ref = g['int_power'][:, 0]
    assert np.allclose(power, ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize('name', GOLDEN)
def test_loglikelihood_vs_reference(contexts, name):
    g, ctx = contexts(name)
    loglike, logprior, status, flat = ctx.eval_batch_host(g['theta'], return_flattheory=True)
    assert np.allclose(flat, g['flattheory'], rtol=1e-11, atol=1e-8)
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(loglike - g['loglikelihood']) <= tol).all(), np.abs(loglike - g['loglikelihood']).max()
    finite = np.isfinite(g['logprior'])
    assert np.allclose(logprior[finite], g['logprior'][finite], rtol=1e-13, atol=1e-13)
    assert np.array_equal(np.isneginf(logprior), np.isneginf(g['logprior']))
    assert np.array_equal(status == 1, np.isneginf(g['logprior']))
    assert (status[finite] == 0).all()


def test_vs_oracle_seeded_batch(contexts):
    """Same seeded inputs through the HIP path and the NumPy oracle, at a size the oracle finishes in seconds (B = 257: ragged tile)."""
    from oracle import np_oracle as orc
    g, ctx = contexts('cfg2_shapefit_window_dense')
    c, priors = observable_constants(g), prior_list(g)
    rng = np.random.RandomState(123)
    lo = np.array([0.9, 0.9, -0.5, 0.5, 0.5, -3.])
    hi = np.array([1.1, 1.1, 0.5, 1.5, 3.5, 3.])
    theta = rng.uniform(lo, hi, size=(257, 6))
    loglike, logprior, status = ctx.eval_batch_host(theta)
    names = [str(n) for n in g['names']]
    ref = []
    for row in theta:
        p = dict(zip(names, row)); p['b1'] = (p['b1'], p['b1'])
        out = orc.fullshape_observable(c, p)
        ref.append(orc.gaussian_loglikelihood(out['flattheory'], c['flatdata'], g['precision'])[0])
    ref = np.array(ref)
    assert (np.abs(loglike - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), np.abs((loglike - ref) / ref).max()
    assert np.allclose(logprior, orc.logprior(theta, priors), rtol=1e-13, atol=1e-13)
    assert (status == 0).all()


def test_edge_cases(contexts):
    g, ctx = contexts('cfg2_shapefit_window')
    theta = g['theta'][:5].copy()
    theta[1, 2] = np.nan
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert status[1] == 3 and status[0] == 0
    # B = 1 (scalar likelihood() call surface) and empty batch
    l1, p1, s1 = ctx.eval_batch_host(g['theta'][:1])
    assert np.isclose(l1[0], g['loglikelihood'][0], rtol=1e-12, atol=1e-10)
    l0, p0, s0 = ctx.eval_batch_host(np.zeros((0, ctx.n_params)))
    assert l0.size == 0
    # data generated from theory => logL(fiducial) = 0 (likelihoods/tests/test_galaxy_clustering.py:6-16)
    names = [str(n) for n in g['names']]
    fid = np.array([[1., 1., 0., 1., 2., 0.]])
    assert names == ['qpar', 'qper', 'dm', 'df', 'b1', 'sn0']
    lf, _, _ = ctx.eval_batch_host(fid)
    assert abs(lf[0]) < 1e-12


def test_large_batch_path_agrees(contexts):
    """> 2048 rows per pass go through the split-K GEMM + slab finalize, smaller passes through the chi2 GEMM: same points, same answers (ragged sizes)."""
    g, ctx = contexts('cfg2_shapefit_window_dense')
    rng = np.random.RandomState(5)
    lo = np.array([0.9, 0.9, -0.5, 0.5, 0.5, -3.])
    hi = np.array([1.1, 1.1, 0.5, 1.5, 3.5, 3.])
    theta = rng.uniform(lo, hi, size=(2500 + 37, 6))
    theta[7, 0] = 0.1   # outside the prior
    big = ctx.eval_batch_host(theta)
    parts = [ctx.eval_batch_host(theta[i:i + 1000]) for i in range(0, len(theta), 1000)]
    small = [np.concatenate([p[k] for p in parts]) for k in range(3)]
    assert (np.abs(big[0] - small[0]) <= 1e-10 * np.maximum(1., np.abs(small[0]))).all()
    assert np.array_equal(big[1], small[1]) and np.array_equal(big[2], small[2])
    assert big[2][7] == 1


def test_logposterior_entry_point(contexts):
    """dl_eval_logposterior = loglikelihood + logprior with the samplers' -inf conventions (samplers/base.py:185-191), torch tensors, asynchronous."""
    import torch
    g, ctx = contexts('cfg2_shapefit_window')
    theta = g['theta'].copy()
    theta[2, 1] = np.nan
    loglike, logprior, status = ctx.eval_batch_host(theta)
    expected = np.where(status == 0, loglike + logprior, -np.inf)
    th = torch.as_tensor(theta, dtype=torch.float64, device='cuda').contiguous()
    out = torch.empty(len(theta), dtype=torch.float64, device='cuda')
    st = torch.empty(len(theta), dtype=torch.int32, device='cuda')
    ctx.eval_logposterior(th, out, status=st)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(st.cpu().numpy(), status)
    assert np.array_equal(np.isneginf(got), np.isneginf(expected)) and np.isneginf(got[2])
    finite = np.isfinite(expected)
    assert np.allclose(got[finite], expected[finite], rtol=1e-15, atol=0.)


def test_repeatable_bitwise(contexts):
    """The chi2 GEMM stages its operands by LDS-DMA ordered only by counted waits and barriers: a misplaced wait shows up as run-to-run differences.
    300 evaluations of one ragged batch must be bit-identical (fixed summation orders everywhere: no atomics on the data path)."""
    import torch
    g, ctx = contexts('cfg2_shapefit_window_dense')
    rng = np.random.RandomState(11)
    lo = np.array([0.9, 0.9, -0.5, 0.5, 0.5, -3.])
    hi = np.array([1.1, 1.1, 0.5, 1.5, 3.5, 3.])
    theta = torch.as_tensor(rng.uniform(lo, hi, size=(1000 + 13, 6)), dtype=torch.float64, device='cuda').contiguous()
    first, out = None, torch.empty(len(theta), dtype=torch.float64, device='cuda')
    for it in range(300):
        ctx.eval_batch(theta, loglike=out)
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(first, out), it
    torch.cuda.synchronize()
    assert torch.isfinite(first).all()
