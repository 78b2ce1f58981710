"""CPU: pin the BAO restatement (bao.py:117-151, 495-534, 881-905; theories/galaxy_clustering/base.py:127-136) to fixtures captured from the
reference, and the FFTLog implementations (third-party in the reference: parity unpinned) to each other and to the brute-force integral."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, prior_list
from bench_configs import bao_point   # noqa: E402,F401  (shared with bench.py)


@pytest.mark.parametrize('space', ['xi', 'pk'])
def test_bao_chain_vs_reference(space):
    g = load_golden('cfg4_bao_' + space)
    c = g['obs0']
    priors = prior_list(g)
    ells = tuple(int(ell) for ell in c['ells'])
    for i, row in enumerate(g['theta']):
        power, broadband = bao_point(g, row)
        assert np.allclose(power, g['wiggle_power'][i], rtol=1e-11, atol=1e-12 * np.abs(g['wiggle_power'][i]).max())
        if space == 'xi':
            theory = orc.get_corr(power, c['kin'], c['s'], ells) + broadband
            flat = np.ravel(theory)
        else:
            theory = power + broadband
            flat = orc.window_apply(theory, matrix_full=c['matrix_full'], shotnoisein=c['shotnoisein'], shotnoiseout=c['shotnoiseout'])
        assert np.allclose(theory, g['theory'][i], rtol=1e-11, atol=1e-13 * np.abs(g['theory'][i]).max())
        logl = orc.gaussian_loglikelihood(flat, c['flatdata'], g['precision'])[0]
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))
        assert np.isclose(orc.logprior(row, priors), g['logprior'][i], rtol=1e-13, atol=1e-13)


def test_fftlog_implementations_and_bruteforce():
    from oracle.np_fftlog import PowerToCorrelation, hankel_operator
    g = load_golden('cfg4_bao_xi')
    c = g['obs0']
    k = np.logspace(-4., 3., 2048)
    power = g['wiggle_power'][0]
    # same construction as get_corr on the FFTLog grid
    ref = orc.get_corr(power, c['kin'], c['s'], (0, 2))
    H = hankel_operator(c['kin'], c['s'], (0, 2))
    mine = np.einsum('lsk,lk->ls', H, power)
    assert np.allclose(mine, ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())     # two implementations of Hamilton's algorithm
    # physical check: brute-force integral (theories/galaxy_clustering/base.py:163-168), agreement limited by truncation / damping of the tail
    brute = orc.bruteforce_correlation(c['kin'], power, c['s'], (0, 2))
    assert np.allclose(mine[0], brute[0], rtol=0.02, atol=2e-5)
    a, b = PowerToCorrelation(k, ell=(0, 2)), orc.FFTLogPowerToCorrelation(k, ell=(0, 2))
    pk = np.array([np.interp(np.log10(k), np.log10(c['kin']), p, right=0.) for p in power])
    (sa, xa), (sb, xb) = a(pk), b(pk)
    mask = (sa[0] > 20.) & (sa[0] < 200.)
    assert np.allclose(sa, sb, rtol=1e-14) and np.allclose(xa[:, mask], xb[:, mask], rtol=1e-12, atol=0.)   # identical to rounding where the data live


def test_fftlog_oracle_against_analytic_hankel_pairs():
    """Independent pin of the FFTLog oracle: Gaussian-damped power laws have closed-form transforms,
        int_0^inf dk k^(l+2) exp(-k^2 sigma^2 / 2) j_l(k s) = sqrt(pi / 2) s^l sigma^-(2l+3) exp(-s^2 / (2 sigma^2))     (Gradshteyn & Ryzhik 6.631.4),
    so xi_l(s) = (-1)^(l/2) / (2 pi^2) x that.  Checks normalisation ((2 pi)^-3/2 s^-3/2 post-factor), phase (-1)^(l/2), output grid and low-ringing offset of
    ``FFTLogPowerToCorrelation`` on the reference's grid (tgc/base.py:62-77: 2048 points on [1e-4, 1e3], zero padding to 4096) without going through scipy's or our own
    conventions twice."""
    k = np.logspace(-4., 3., 2048)
    ells = (0, 2, 4)
    fftlog = orc.FFTLogPowerToCorrelation(k, ell=ells, q=0, lowring=True)
    for sigma in (6., 12., 25.):
        s, xi = fftlog(np.array([k**ell * np.exp(-0.5 * (k * sigma)**2) for ell in ells]))
        for ill, ell in enumerate(ells):
            analytic = (-1.)**(ell // 2) / (2. * np.pi**2) * np.sqrt(np.pi / 2.) * s[ill]**ell * sigma**(-(2 * ell + 3)) * np.exp(-0.5 * (s[ill] / sigma)**2)
            mask = (s[ill] > 1.) & (s[ill] < 200.)
            # ell = 0: the integrand does not vanish at the lower edge of the grid (P -> 1): an absolute aliasing floor (1e-10 .. 4e-9 of the peak for sigma = 6 .. 25); higher multipoles reach rounding
            assert np.abs(xi[ill][mask] - analytic[mask]).max() <= (1e-8 if ell == 0 else 1e-12) * np.abs(analytic[mask]).max(), (sigma, ell)
