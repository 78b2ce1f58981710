"""The RR-count window of correlation function multipoles (desilike_amd/observables/galaxy_clustering/correlation_function.py::window_matrix_RR; reference
window.py:71-138) against the matrix of the reference's own function on synthetic pair counts (tests/golden/make_rr_window_fixture.py), and through the window class."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_matrix_against_the_reference():
    from desilike_amd.observables.galaxy_clustering import window_matrix_RR
    g = np.load(os.path.join(HERE, 'golden', 'rr_window.npz'))
    sin, matrix = window_matrix_RR({0: g['sout0'], 2: g['sout2']}, g['sedges'], g['muedges'], g['wcounts'], ellsin=(0, 2, 4), resolution=2)
    assert np.array_equal(sin, g['sin']) and matrix.shape == g['matrix'].shape
    assert np.allclose(matrix, g['matrix'], rtol=1e-13, atol=1e-15)
    # isotropic, separable counts: the estimator is unbiased -- the monopole passes through (rows sum to one), the quadrupole does not leak into it
    counts = ((g['sedges'][:-1] + g['sedges'][1:]) / 2.)[:, None]**2 * np.ones((1, g['muedges'].size - 1))
    sin, matrix = window_matrix_RR({0: g['sout0']}, g['sedges'], g['muedges'], counts, ellsin=(0, 2), resolution=1)
    n = sin.size
    assert np.allclose(matrix[:n].sum(axis=0), 1., rtol=1e-12) and np.abs(matrix[n:]).max() < 1e-12


def test_window_class_takes_the_counts():
    from desilike_amd.observables.galaxy_clustering import WindowedCorrelationFunctionMultipoles
    from desilike_amd.theories.galaxy_clustering import KaiserTracerCorrelationFunctionMultipoles, StandardPowerSpectrumTemplate
    g = np.load(os.path.join(HERE, 'golden', 'rr_window.npz'))
    theory = KaiserTracerCorrelationFunctionMultipoles(template=StandardPowerSpectrumTemplate(z=0.5, fiducial='synthetic'))
    window = WindowedCorrelationFunctionMultipoles(sedges=[g['sout0'], g['sout2']], ells=(0, 2), ellsin=(0, 2, 4), theory=theory,
                                                   wmatrix={'sedges': g['sedges'], 'muedges': g['muedges'], 'wcounts': g['wcounts'], 'resolution': 2})
    window.initialize()
    assert window.ellsin == (0, 2, 4) and np.array_equal(window.sin, g['sin'])
    assert np.allclose(window.matrix_full, g['matrix'].T, rtol=1e-13, atol=1e-15)
    assert tuple(theory.ells) == (0, 2, 4) and np.array_equal(theory.s, g['sin'])
