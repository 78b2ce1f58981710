"""GPU (-m gpu): the batched device FFTLog (dl_fftlog_*, csrc/dl_fftlog.hip) through the C ABI, against
 (i) the host restatement of the same algorithm with numpy.fft (oracle/np_fftlog.py),
 (ii) the oracle's independent implementation through scipy.fft.fht (oracle/np_oracle.py FFTLogPowerToCorrelation),
 (iii) the Hankel operator the BAO xi_ell path folds into the window, built by one device batch vs built on the host.
The reference's transform is third-party (cosmoprimo, unpinned): "parity unpinned" for row a11, see oracle/np_oracle.py.
Tolerance: 1e-13 of the largest |xi s^{3/2}| of the row (an FFT of 4096 points in float64; measured ~3e-16)."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden

pytestmark = pytest.mark.gpu


def _scaled(s, xi):
    return xi * s**1.5


@pytest.mark.parametrize('n', [2048, 1000, 300, 37, 8])
def test_device_fftlog_vs_host_restatement(n):
    from desilike_amd.fftlog import PowerToCorrelation
    ells = (0, 2, 4)
    k = np.logspace(-4., 3., n)
    rng = np.random.RandomState(n)
    B = 5
    fun = rng.standard_normal((B, len(ells), n)) * (k / 0.1)**-1.2 * np.exp(-(np.log(k / 0.05) / 3.)**2)
    from oracle.np_fftlog import PowerToCorrelation as HostPowerToCorrelation
    host, dev = HostPowerToCorrelation(k, ell=ells), PowerToCorrelation(k, ell=ells, device=0)
    s, xi = dev(fun)
    assert xi.shape == fun.shape
    for b in range(B):
        sh, xh = host(fun[b])
        assert np.array_equal(s, sh)
        a, r = _scaled(s, xi[b]), _scaled(sh, xh)
        assert (np.abs(a - r).max(axis=-1) <= 1e-13 * np.abs(r).max(axis=-1)).all(), (n, b, np.abs(a - r).max(axis=-1) / np.abs(r).max(axis=-1))
    # linearity and batch independence (size-independent properties): transform of a sum, permuted batch
    s2, x2 = dev(fun[::-1] + 2. * fun)
    assert np.allclose(_scaled(s, x2), _scaled(s, xi[::-1] + 2. * xi), rtol=0., atol=1e-12 * np.abs(_scaled(s, xi)).max())


def test_device_fftlog_vs_oracle_on_reference_grid():
    """The grid of the reference's get_corr (tgc/base.py:62-77: 2048 points, npad = 4096) on the fixture's P_ell; resident batch, explicit stream, B = 0."""
    import torch
    from desilike_amd.fftlog import PowerToCorrelation
    g = load_golden('cfg4_bao_xi')
    c = g['obs0']
    k = np.logspace(-4., 3., 2048)
    power = g['wiggle_power']                                                   # [B, n_ell, n_kin]
    pk = np.array([[np.interp(np.log10(k), np.log10(c['kin']), p, right=0.) for p in point] for point in power])
    dev, oracle = PowerToCorrelation(k, ell=(0, 2), engine='hip', device=0), orc.FFTLogPowerToCorrelation(k, ell=(0, 2))
    stream = torch.cuda.Stream(device=0)
    fun = torch.as_tensor(pk, dtype=torch.float64, device='cuda:0').contiguous()
    out = torch.full_like(fun, np.nan)
    with torch.cuda.stream(stream):
        dev.apply_device(fun, out=out, stream=stream.cuda_stream)
    stream.synchronize()
    xi = out.cpu().numpy()
    for b in range(len(pk)):
        so, xo = oracle(pk[b])
        mask = (so[0] > 20.) & (so[0] < 200.)
        assert np.allclose(xi[b][:, mask], xo[:, mask], rtol=1e-12, atol=0.)     # identical to rounding where the data live
    empty = dev.apply_device(torch.empty((0, 2, 2048), dtype=torch.float64, device='cuda:0'))
    assert empty.shape == (0, 2, 2048)
    # bitwise repeatability
    again = dev.apply_device(fun)
    torch.cuda.synchronize()
    assert torch.equal(again, out)


def test_hankel_operator_device_vs_host():
    from desilike_amd.fftlog import hankel_operator
    g = load_golden('cfg4_bao_xi')
    c = g['obs0']
    from oracle.np_fftlog import hankel_operator as host_hankel_operator
    Hd, Hh = hankel_operator(c['kin'], c['s'], (0, 2), device=0), host_hankel_operator(c['kin'], c['s'], (0, 2))
    # entries far below the largest one carry the FFT's rounding noise (relative to the transform's maximum, in either implementation)
    assert np.allclose(Hd, Hh, rtol=1e-9, atol=1e-13 * np.abs(Hh).max())
    ref = orc.get_corr(g['wiggle_power'][0], c['kin'], c['s'], (0, 2))
    mine = np.einsum('lsk,lk->ls', Hd, g['wiggle_power'][0])
    assert np.allclose(mine, ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())


def test_plan_argument_errors():
    from desilike_amd._lib import FFTLogPlan, LibraryError
    with pytest.raises(LibraryError):
        FFTLogPlan(10, 24, np.ones(10), np.ones((1, 13, 2)), np.ones((1, 10)), device=0)       # npad not a power of two
    with pytest.raises(LibraryError):
        FFTLogPlan(10, 16384, np.ones(10), np.ones((1, 8193, 2)), np.ones((1, 10)), device=0)  # npad too large for LDS


def test_device_fftlog_against_analytic_hankel_pairs():
    """The device transform against closed-form pairs (Gaussian-damped power laws, Gradshteyn & Ryzhik 6.631.4) -- independent of scipy's and of the host
    restatement's conventions (see tests/test_oracle_bao.py::test_fftlog_oracle_against_analytic_hankel_pairs)."""
    from desilike_amd.fftlog import PowerToCorrelation
    k = np.logspace(-4., 3., 2048)
    ells = (0, 2, 4)
    dev = PowerToCorrelation(k, ell=ells, engine='hip', device=0)
    sigmas = (6., 12., 25.)
    fun = np.array([[k**ell * np.exp(-0.5 * (k * sigma)**2) for ell in ells] for sigma in sigmas])
    s, xi = dev(fun)
    for isig, sigma in enumerate(sigmas):
        for ill, ell in enumerate(ells):
            analytic = (-1.)**(ell // 2) / (2. * np.pi**2) * np.sqrt(np.pi / 2.) * s[ill]**ell * sigma**(-(2 * ell + 3)) * np.exp(-0.5 * (s[ill] / sigma)**2)
            mask = (s[ill] > 1.) & (s[ill] < 200.)
            assert np.abs(xi[isig, ill][mask] - analytic[mask]).max() <= (1e-8 if ell == 0 else 1e-12) * np.abs(analytic[mask]).max(), (sigma, ell)
