"""The xi_ell path pinned a second, independent way (tests/golden/make_xi_pin_fixture.py): the reference's ``get_corr`` / window / likelihood ran on
top of ``scipy.fft.fht`` called directly (no class of oracle/ in between), and its brute-force integral (tgc/base.py:163-168) was recorded beside it.
CPU: the oracle's restatement against both.  GPU: the device path (own FFTLog operator folded into the window) against the fixture.  cosmoprimo itself is absent from
the image: agreement with ITS transform stays unclaimed."""
import os

import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))


def load_pin(name):
    return np.load(os.path.join(HERE, 'golden', 'xi_pin_{}.npz'.format(name)))


def test_oracle_get_corr_vs_scipy_fht_pipeline():
    """The oracle's get_corr (prologue, Hankel step, epilogue) on the reference's P_ell reproduces the reference's xi_ell obtained with scipy.fft.fht in the Hankel slot."""
    for name in ['kaiser', 'bao']:
        g = load_pin(name)
        ells = tuple(int(ell) for ell in g['ells'])
        if name == 'bao': continue     # (broadband terms are added after the transform: covered through the likelihood values on the GPU)
        for power, corr in zip(g['power'], g['corr']):
            got = orc.get_corr(power, g['kin'], g['s'], ells)
            assert np.allclose(got, corr, rtol=1e-12, atol=1e-14 * np.abs(corr).max())


def test_bruteforce_integral_level():
    """tgc/base.py:163-168: the trapezoid integral of k^3 P_ell j_ell(k s) over the theory grid (k <= 0.6 h/Mpc, cut sharply) against the FFTLog result (tail extrapolated
    and damped).  The oracle's restatement of the formula is exact; the two methods agree to the truncation of the k range: 3 % of max |xi_ell| at s in [22.5, 167.5] Mpc/h
    (the reference's docstring: 'difference ... comes from the effect of truncation / damping')."""
    g = load_pin('kaiser')
    ells = tuple(int(ell) for ell in g['ells'])
    for power, corr, brute in zip(g['power'], g['corr'], g['bruteforce']):
        assert np.allclose(orc.bruteforce_correlation(g['kin'], power, g['s'], ells), brute, rtol=1e-13, atol=1e-16)
        scale = np.abs(corr).max(axis=-1, keepdims=True)
        assert (np.abs(corr - brute) <= 0.03 * scale).all()
        assert (np.abs(corr - brute)[0] <= 0.015 * scale[0]).all()      # monopole: 1.2 %


@pytest.mark.gpu
def test_device_xi_vs_scipy_fht_pipeline():
    """Device path against the reference-with-scipy.fft.fht fixtures: log-likelihoods to 1e-10, xi_ell to 1e-10 of its maximum."""
    from desilike_amd import vmap
    from test_gpu_kaiser_xi import make_kaiser_xi
    from test_host_api import make_cfg4
    for name, (g0, like) in [('kaiser', make_kaiser_xi('kaiser_xi')), ('bao', make_cfg4('xi'))]:
        g = load_pin(name)
        names = [str(n) for n in g['names']]
        assert like.varied_params.names() == names
        (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({n: g['theta'][:, i] for i, n in enumerate(names)})
        assert errors == {}
        assert (np.abs(derived['loglikelihood'] - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all()
        flat = like._get_context().eval_batch_host(g['theta'], return_flattheory=True)[3]
        corr = g['corr'].reshape(len(g['theta']), -1)
        assert np.allclose(flat, corr, rtol=1e-9, atol=1e-10 * np.abs(corr).max())
