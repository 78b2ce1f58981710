"""Fiducial-cosmology providers for the power-spectrum templates.

The reference takes its fiducial linear power spectrum from the third-party Boltzmann wrapper ``cosmoprimo``
(power_template.py:49-66), which is out of scope (CPU, external).  The hot path only
needs the fiducial *tables*: ``pk_dd(k)``, optionally the no-wiggle ``pknow_dd(k)``, and the growth rate ``f``.
"""
import numpy as np


class TabulatedFiducial(object):
    """Fiducial given as tables (e.g. exported once from cosmoprimo / CLASS / CAMB on a CPU node)."""

    def __init__(self, k, pk_dd, f, pknow_dd=None, sigma8=None, rs_drag=None, h=None, A_s=None, n_s=None, k_pivot=0.05):
        self.h, self.A_s, self.n_s, self.k_pivot = h, A_s, n_s, k_pivot   # primordial spectrum A_s (k / k_pivot)^(n_s - 1) and h: only the PNG theory needs them
        self.k = np.asarray(k, dtype='f8')
        self._logpk = np.log(np.asarray(pk_dd, dtype='f8'))
        self._logpknow = None if pknow_dd is None else np.log(np.asarray(pknow_dd, dtype='f8'))
        self.f = float(f)
        self.sigma8 = sigma8
        self.rs_drag = rs_drag   # sound horizon of the fiducial cosmology [Mpc/h]: sets the default broadband pivot 2 pi / r_d (bao.py:488)

    def _interp(self, k, table):
        from scipy import interpolate
        k = np.asarray(k, dtype='f8')
        if k.shape == self.k.shape and np.array_equal(k, self.k):
            return np.exp(table)
        return np.exp(interpolate.CubicSpline(np.log(self.k), table)(np.log(k)))

    def pk_dd(self, k):
        return self._interp(k, self._logpk)

    def pknow_dd(self, k):
        if self._logpknow is None:
            raise ValueError('no-wiggle table not provided')
        return self._interp(k, self._logpknow)

    def pk_prim(self, k):
        """Dimensionless primordial spectrum (what ``cosmo.get_primordial(mode='scalar').pk_interpolator()`` returns, primordial_non_gaussianity.py:83)."""
        if self.A_s is None or self.n_s is None or self.h is None:
            raise ValueError('the PNG theory needs the primordial spectrum: pass h, A_s, n_s (and k_pivot, in the units of k) to TabulatedFiducial')
        return self.A_s * (np.asarray(k, dtype='f8') / self.k_pivot)**(self.n_s - 1.)


class SyntheticFiducial(object):
    r"""Analytic synthetic cosmology used for benchmarks and fixtures:
    :math:`P(k) = A (k / 0.05)^{n_s} T_\mathrm{BBKS}(k / k_\mathrm{eq})^2 (1 + w \sin(k r_s) e^{-(8 k)^2})`, ``f = 0.8``."""

    def __init__(self, A=2.5e4, n_s=0.965, keq=0.015, rs=100., wiggle=0.05, f=0.8):
        self.A, self.n_s, self.keq, self.rs, self.wiggle, self.f = A, n_s, keq, rs, wiggle, f
        self.rs_drag = rs
        self.h, self.A_s, self.k_pivot = 0.7, 2.1e-9, 0.05

    def pk_prim(self, k):
        return self.A_s * (np.asarray(k, dtype='f8') / self.k_pivot)**(self.n_s - 1.)

    def _pk(self, k, wiggle):
        k = np.asarray(k, dtype='f8')
        q = k / self.keq
        T = np.log(1. + 2.34 * q) / (2.34 * q) * (1. + 3.89 * q + (16.1 * q)**2 + (5.46 * q)**3 + (6.71 * q)**4)**(-0.25)
        return self.A * (k / 0.05)**self.n_s * T**2 * (1. + wiggle * np.sin(k * self.rs) * np.exp(-(8. * k)**2))

    def pk_dd(self, k):
        return self._pk(k, self.wiggle)

    def pknow_dd(self, k):
        return self._pk(k, 0.)


class FiducialWarning(UserWarning):
    """The synthetic stand-in cosmology was used where the reference would have used a real one."""


def get_fiducial(fiducial):
    """``TabulatedFiducial`` / ``SyntheticFiducial`` instance, dict of ``TabulatedFiducial`` arguments, or 'synthetic' (explicit opt-in: benchmarks, fixtures).
    The reference's default 'DESI' (a ``cosmoprimo.fiducial`` entry, power_template.py:49-66) cannot be honoured without cosmoprimo: the synthetic
    analytic cosmology stands in WITH A WARNING -- its P(k), growth rate and sound horizon are not DESI's; pass tables exported from a Boltzmann code
    for a real analysis."""
    if isinstance(fiducial, str) and fiducial.lower() == 'synthetic':
        return SyntheticFiducial()
    if fiducial is None or (isinstance(fiducial, str) and fiducial.lower() == 'desi'):
        import warnings
        warnings.warn("fiducial={!r}: no Boltzmann code on this path, the SYNTHETIC analytic cosmology (BBKS-like P(k), f = 0.8, r_d = 100 Mpc/h) stands in; pass "
                      "fiducial=TabulatedFiducial(k, pk_dd, f, pknow_dd=..., rs_drag=...) (or a dict of these) for real data, or fiducial='synthetic' to "
                      "silence this warning".format(fiducial), FiducialWarning, stacklevel=3)
        return SyntheticFiducial()
    if isinstance(fiducial, str):
        raise ValueError('unknown fiducial {!r}'.format(fiducial))
    if isinstance(fiducial, dict):
        return TabulatedFiducial(**fiducial)
    return fiducial
