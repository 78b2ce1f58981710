"""Stand-in for the un-vendored third-party package ``cosmoprimo``.

TEST INFRASTRUCTURE ONLY.  Written from scratch for this repo: it provides a *synthetic analytic
cosmology* (BBKS-like P(k) with a damped sinusoidal BAO wiggle) with just enough surface for the
reference (``/root/reference``, never shipped) to import and run its own arithmetic downstream of
``pk_dd_fid[k]`` / ``f_fid`` unmodified, so that ``make_golden.py`` can capture golden vectors.
Nothing here is cosmoprimo's algorithm; the numbers are inputs, not results under test.
"""
import numpy as np

from . import constants
from .cosmology import Cosmology, CosmologyError, BaseEngine, BaseSection
from .interpolator import PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D


class _Fiducial(object):

    @staticmethod
    def DESI(**kwargs):
        return Cosmology(**kwargs)


fiducial = _Fiducial()


class PowerSpectrumBAOFilter(object):
    """Returns the wiggle-free version of the synthetic spectrum."""

    def __init__(self, pk_interpolator, engine=None, cosmo=None, cosmo_fid=None, **kwargs):
        self(pk_interpolator, cosmo=cosmo)

    def __call__(self, pk_interpolator, cosmo=None):
        self.pk_interpolator = pk_interpolator
        return self

    def smooth_pk_interpolator(self):
        return self.pk_interpolator.clone(wiggle=0.)


class PowerToCorrelation(object):
    """Placeholder: replaced at golden-generation time by the build's own FFTLog oracle (parity unpinned)."""

    def __init__(self, k, ell=0, q=0, lowring=True, **kwargs):
        import sys, os
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', '..', '..'))
        from oracle.np_oracle import FFTLogPowerToCorrelation
        self._impl = FFTLogPowerToCorrelation(k, ell=ell, q=q, lowring=lowring)

    def __call__(self, fun):
        return self._impl(np.asarray(fun))
