"""CPU: convergence diagnostics against the reference's own outputs on seeded chains (tests/golden/diagnostics.npz <- tests/golden/make_diagnostics_fixture.py)."""
import numpy as np

from desilike_amd import diagnostics


def load():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'diagnostics.npz'))


def test_gelman_rubin_geweke_iact_vs_reference():
    g = load()
    chains = list(g['chains'])                                     # [nchains][niterations, nwalkers, ndim]
    assert np.allclose(diagnostics.gelman_rubin(chains, method='eigen', check_valid='ignore'), g['eigen_gr_full'], rtol=1e-10)
    assert np.allclose(diagnostics.gelman_rubin(chains, method='diag'), g['diag_gr_full'], rtol=1e-12)
    burnin, nsplits, lensplits = int(g['check_burnin']), int(g['check_nsplits']), int(g['check_lensplits'])
    split = [chain[burnin + islab * lensplits:burnin + (islab + 1) * lensplits] for islab in range(nsplits) for chain in chains]
    assert np.allclose(diagnostics.gelman_rubin(split, method='eigen', check_valid='ignore'), g['eigen_gr'], rtol=1e-10)
    assert np.allclose(diagnostics.gelman_rubin(split, method='diag'), g['diag_gr'], rtol=1e-12)
    assert np.allclose(diagnostics.geweke(split, first=0.1, last=0.5), g['geweke'], rtol=1e-10)
    walkers = np.concatenate([np.moveaxis(chain[burnin:], 1, 0) for chain in chains])     # one series per walker: [nchains * nwalkers, nsamples, ndim]
    assert np.allclose(diagnostics.integrated_autocorrelation_time(walkers), g['iact'], rtol=1e-10)


def test_diagnostics_history():
    d = diagnostics.Diagnostics()
    assert d.add_test('x', 'x', 0.5, limits=(None, 1.), stable_over=2) is False     # first call: not yet stable
    assert d.add_test('x', 'x', 0.4, limits=(None, 1.), stable_over=2) is True
    assert d.add_test('x', 'x', 2., limits=(None, 1.), stable_over=2) is False
    assert d['x'] == [0.5, 0.4, 2.] and d['x_test'] == [True, True, False]
    assert d.add_test('y', 'y', np.nan, limits=(None, 1.)) is False and d.add_test('z', 'z', 3.) is True
