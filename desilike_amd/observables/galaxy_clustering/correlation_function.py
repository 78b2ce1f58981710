"""Correlation function multipoles observable and its window (reference: desilike/observables/galaxy_clustering/window.py:536-793,
correlation_function.py:20-120, 380-381)."""
import numpy as np

from ...base import BaseCalculator
from ... import utils
from ...utils import window_matrix_bininteg
from .window import SystematicTemplatePowerSpectrumMultipoles, _append_systematic_templates
from ._binning import MultipoleBins


class SystematicTemplateCorrelationFunctionMultipoles(SystematicTemplatePowerSpectrumMultipoles):
    """Systematic templates for correlation function multipoles (window.py:1363-1384): ``flatcorr += sum_i syst_i template_i``."""
    _xname = 's'

    def _default_x(self):
        return np.linspace(20., 200, 101)


def _legendre_pair_primitive(*ells):
    """Primitive F(mu) of the product of the Legendre polynomials of orders ``ells`` (numpy polynomial, highest power first)."""
    from numpy.polynomial import legendre as npleg
    poly = np.array([1.])
    for ell in ells:
        coeffs = np.zeros(int(ell) + 1); coeffs[-1] = 1.
        poly = np.polymul(poly, npleg.leg2poly(coeffs)[::-1])
    return np.polyint(poly)


def _integral_beyond(primitive, mu_min):
    """Integral over |mu| > mu_min, mu in [-1, 1], of the polynomial whose primitive is ``primitive``."""
    F = lambda mu: np.polyval(primitive, mu)
    return F(1.) - F(mu_min) + F(-mu_min) - F(-1.)


def window_matrix_RR(soutedges, sedges, muedges, wcounts, ellsin=(0, 2, 4), resolution=1):
    r"""Window matrix of correlation function multipoles estimated with random-random pair counts ``wcounts [n_s, n_mu]`` on the fine grid ``sedges`` x ``muedges``
    (reference: window.py:71-138).  Within an output bin (``factor`` fine bins) the estimator weights every fine separation bin by its share of the pairs at each mu, so the
    measured multipole :math:`\ell` receives the theory multipole :math:`\ell'` through

    .. math:: (2 \ell + 1) \sum_{\mu\,bins} \frac{RR(s, \mu)}{\sum_{s' \in bin} RR(s', \mu)} \int_{\mu\,bin} L_\ell L_{\ell'} d\mu \Big/ \sum_{non-empty} \Delta\mu ,

    followed by the volume-weighted integration of the theory over the fine bins (:func:`window_matrix_bininteg`).  ``soutedges``: dict multipole -> output edges ((low, high) pairs as the
    reference carries them, or 1D; each starting on a fine edge, widths a multiple of the fine width).  Returns (sin, matrix [n_ellin * n_sin, n_out]) with the theory separations nothing depends on removed."""
    def pairs(edges):   # (low, high) per bin, as the reference carries edges; a 1D array of edges is accepted
        edges = np.asarray(edges, dtype='f8')
        return np.column_stack([edges[:-1], edges[1:]]) if edges.ndim == 1 else edges

    sedges, muedges, wcounts = pairs(sedges), pairs(muedges), np.asarray(wcounts, dtype='f8')
    sin, binmatrix = window_matrix_bininteg([sedges], resolution=resolution)          # [n_sin, n_fine]
    dmu = muedges[:, 1] - muedges[:, 0]
    lines, used = [], np.zeros(sin.size, dtype='?')
    for ellout, edges in soutedges.items():
        edges = pairs(edges)
        first = np.flatnonzero(sedges[:, 0] == edges[0, 0])
        if not first.size: raise ValueError('output edges {} not found in RR s-edges {}'.format(edges, sedges))
        first = int(first[0])
        factor = int(np.rint((edges[0, 1] - edges[0, 0]) / (sedges[first, 1] - sedges[first, 0])))
        if factor == 0: raise ValueError('s-resolution of RR counts is larger than required output s-binning')
        line = []
        for ellin in ellsin:
            primitive = _legendre_pair_primitive(ellout, ellin)
            integral = np.polyval(primitive, muedges[:, 1]) - np.polyval(primitive, muedges[:, 0])
            fine = np.zeros((len(sedges), len(edges)), dtype='f8')
            for iout in range(len(edges)):
                rows = slice(first + factor * iout, first + factor * (iout + 1))
                counts = wcounts[rows]
                total = counts.sum(axis=0)
                filled = total != 0.
                share = counts / np.where(filled, total, 1.)
                fine[rows, iout] = (2. * ellout + 1.) * np.sum(share * filled * integral, axis=-1) / np.sum(filled * dmu)
            block = binmatrix.dot(fine)                                                 # [n_sin, n_out of this multipole]
            used |= np.any(block != 0., axis=1)
            line.append(block.T)
        lines.append(line)
    matrix = np.block([[block[:, used] for block in line] for line in lines])          # [n_out, n_ellin * n_sin kept]
    return sin[used], matrix.T


class TopHatFiberCollisionsCorrelationFunctionMultipoles(BaseCalculator):
    r"""Fiber collisions in configuration space (Hahn et al. 2016, arXiv:1609.01714; reference: window.py:1192-1250): a fraction ``fs`` of the pairs with transverse
    separation below ``Dfc`` is lost -- at separation s these are the pairs with :math:`|\mu| > \mu_{min}(s) = \sqrt{1 - (D_{fc} / s)^2}`.  The multipoles mix at fixed s:

    .. math:: \xi^{fc}_\ell(s) = \sum_{\ell'} \Big[\delta_{\ell\ell'} - f_s \frac{2 \ell + 1}{2} \int_{|\mu| > \mu_{min}} L_\ell L_{\ell'} d\mu\Big] \xi_{\ell'}(s)
              - f_s \frac{2 \ell + 1}{2} \int_{|\mu| > \mu_{min}} L_\ell d\mu

    Init-time constants only: ``kernel_correlated [n_ell, n_ellin, n_s]``, ``kernel_uncorrelated [n_ell, n_s]`` (exact polynomial integrals); the window folds them into its
    matrix and offset (window.py:688-706).  ``mu_range_cut``: divide by the uncut range of mu (window.py:1243-1245)."""

    def _segments(self):
        """[(weight, transverse scale)]: the pair-loss function as a sum of top hats; here one."""
        return [(self.fs, np.array([0., self.Dfc]))]

    def _read_kernel(self, init):
        self.fs, self.Dfc = float(init.get('fs', 1.)), float(init.get('Dfc', 0.))
        self.mu_range_cut = bool(init.get('mu_range_cut', False))

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        self.ells = tuple(init.get('ells', (0, 2, 4)))
        theory = init.get('theory', None)
        if theory is None:
            from ...theories.galaxy_clustering import KaiserTracerCorrelationFunctionMultipoles
            theory = KaiserTracerCorrelationFunctionMultipoles()
        self.theory = theory
        if init.get('s', None) is not None: theory.init.update(s=np.asarray(init['s'], dtype='f8'))
        theory.initialize()
        self.s, self.ellsin = np.array(theory.s, dtype='f8'), tuple(theory.ells)
        self.sin = self.s
        self.with_uncorrelated = bool(init.get('with_uncorrelated', True))
        self._read_kernel(init)

        def lost(*ells):   # sum over the top hats of weight x [pairs beyond mu_min(outer edge) - pairs beyond mu_min(inner edge)]
            primitive, total = _legendre_pair_primitive(*ells), 0.
            for weight, edges in self._segments():
                mu_min = np.sqrt(np.clip(1. - (edges[:, None] / self.s)**2, 0., None))
                beyond = _integral_beyond(primitive, mu_min)
                total = total + weight * (beyond[1] - beyond[0])
            return total

        self.kernel_uncorrelated = -np.array([(2. * ell + 1.) / 2. * lost(ell) for ell in self.ells])
        kernels = np.empty((len(self.ells), len(self.ellsin), self.s.size), dtype='f8')
        for iout, ellout in enumerate(self.ells):
            for iin, ellin in enumerate(self.ellsin):
                kernels[iout, iin] = float(ellin == ellout) - (2. * ellout + 1.) / 2. * lost(ellout, ellin)
                if getattr(self, 'mu_range_cut', False):
                    mu_min = np.sqrt(np.clip(1. - (self.Dfc / self.s)**2, 0., None))
                    kernels[iout, iin][mu_min > 0.] /= mu_min[mu_min > 0.]
        self.kernel_correlated = kernels
        self._initialized = True
        return self


class FiberCollisionsCorrelationFunctionMultipoles(TopHatFiberCollisionsCorrelationFunctionMultipoles):
    """The same with a pair-loss function tabulated in transverse separation: ``kernel`` (the fraction of pairs lost) at ``sep``, read as a sum of top hats -- one per
    interval of ``sep``, at the mean of the two tabulated values (reference: window.py:1132-1189).  As in the reference, a table that does not start at ``sep = 0`` is
    extended down to 0 with its first value, and the number of intervals used stays that of the table as given (window.py:865-875, 1162-1166)."""

    def _read_kernel(self, init):
        sep, kernel = np.array(init['sep'], dtype='f8'), np.array(init['kernel'], dtype='f8')
        self._nsegments = sep.size - 1
        if kernel.size == 1: kernel = np.full_like(sep, kernel.flat[0])
        if sep[0] > 0.: sep, kernel = np.insert(sep, 0, 0.), np.insert(kernel, 0, kernel[0])
        self.sep, self.kernel = sep, kernel
        self.mu_range_cut = False

    def _segments(self):
        return [(0.5 * (self.kernel[i] + self.kernel[i + 1]), self.sep[i:i + 2]) for i in range(self._nsegments)]

    def to_tophat(self):
        """The top hat with the same lost fraction and first moment (window.py:1186-1189)."""
        fs = np.trapezoid(self.kernel, x=self.sep) / np.trapezoid(self.sep, x=self.sep)
        Dfc = 2. * np.trapezoid(self.sep * self.kernel, x=self.sep) / np.trapezoid(self.kernel, x=self.sep)
        return TopHatFiberCollisionsCorrelationFunctionMultipoles(s=self.s, ells=self.ells, theory=self.theory, fs=fs, Dfc=Dfc)


class WindowedCorrelationFunctionMultipoles(BaseCalculator):
    """Window (binning) effect on correlation function multipoles: ``slim``, ``s``, ``sedges``, ``ells``,
    ``wmatrix`` (None, ``{'resolution': n}`` or 2D array with ``sin``, ``ellsin``), ``theory`` -- same meaning as the reference's."""

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        wmatrix, sin, ellsin = init.get('wmatrix', None), init.get('sin', None), init.get('ellsin', None)
        # output binning (window.py:583-640): rules in _binning.MultipoleBins
        bins = MultipoleBins.resolve(x=init.get('s', None), edges=init.get('sedges', None), lim=init.get('slim', None), ells=init.get('ells', None),
                                     default_step=5., default_edges=np.arange(17.5, 155., 5.), label='s', lim_from_edges=True)
        self.ells, self.s, self.sedges, self.smasklim = bins.ells, bins.x, bins.edges, bins.masklim
        theory = init.get('theory', None)
        if theory is None:
            raise ValueError('provide theory (e.g. DampedBAOWigglesTracerCorrelationFunctionMultipoles)')
        self.theory = self._require(theory)
        self.matrix_full, self.smask, self.offset = None, None, None
        if wmatrix is None:   # window.py:649-656
            self.ellsin = tuple(self.ells)
            self.sin, self.smask = bins.input_grid()
        elif isinstance(wmatrix, dict):
            if 'wcounts' in wmatrix:   # window.py:659-662: Legendre mixing from the mu-distribution of the RR counts
                self.ellsin = tuple(ellsin or self.ells)
                self.sin, matrix_full = window_matrix_RR({ell: self.sedges[ill] for ill, ell in enumerate(self.ells)}, ellsin=self.ellsin, **wmatrix)
            else:
                self.ellsin = tuple(self.ells)
                self.sin, matrix_full = window_matrix_bininteg(self.sedges, **wmatrix)
            self.matrix_full = matrix_full.T
        elif isinstance(wmatrix, np.ndarray):   # window.py:667-681: the reference takes the matrix as [input, output] here (transposed w.r.t. P_ell)
            self.ellsin = tuple(ellsin or self.ells)
            self.sin = np.ravel(np.asarray(sin, dtype='f8')).copy()
            self.matrix_full = np.array(wmatrix, dtype='f8').T
        else:
            raise ValueError('unrecognized wmatrix {}'.format(wmatrix))
        fiber_collisions = init.get('fiber_collisions', None)
        if fiber_collisions is not None:   # window.py:688-706: xi -> K xi + u at every input separation, then the window
            self.theory.init.update(s=self.sin, ells=self.ellsin)
            fiber_collisions.init.update(ells=self.ellsin, theory=self.theory, s=self.sin)
            fiber_collisions.initialize()
            kc = fiber_collisions.kernel_correlated
            kernel = np.block([[np.diag(kc[iout, iin]) for iin in range(kc.shape[1])] for iout in range(kc.shape[0])])
            uncorrelated = fiber_collisions.kernel_uncorrelated.ravel() if fiber_collisions.with_uncorrelated else None
            if self.matrix_full is None:
                self.offset, self.matrix_full = uncorrelated, kernel
            else:
                if uncorrelated is not None: self.offset = self.matrix_full.dot(uncorrelated)
                self.matrix_full = self.matrix_full.dot(kernel)
        self.fiber_collisions = fiber_collisions
        systematic_templates = init.get('systematic_templates', None)
        if systematic_templates is not None:   # window.py:697-701
            if not isinstance(systematic_templates, SystematicTemplateCorrelationFunctionMultipoles):
                systematic_templates = SystematicTemplateCorrelationFunctionMultipoles(templates=systematic_templates)
            systematic_templates.init.update(s=self.s, ells=self.ells)
            systematic_templates.initialize()
        self.systematic_templates = systematic_templates
        self.theory.init.update(s=self.sin, ells=self.ellsin)
        self.theory.initialize()
        self.shotnoise = 0.
        self._initialized = True
        return self

    kmask = property(lambda self: self.smask)

    def _window_spec(self):
        self.initialize()
        fold = self.theory._fold()                                       # theory vector = fold . [device output, broadband parameters]
        wmatrix = fold if self.matrix_full is None else self.matrix_full.dot(fold)
        wmatrix = _append_systematic_templates(wmatrix, self.systematic_templates, self.smask, wmatrix.shape[1])
        return dict(wmatrix=wmatrix, kmask=None if self.smask is None else np.asarray(self.smask, dtype='i4'), offset=self.offset)

    def _pass_params(self):
        self.initialize()
        return list(self.systematic_templates.templates) if self.systematic_templates is not None else []

    def _extra_params(self):
        self.initialize()
        return list(self.systematic_templates.params) if self.systematic_templates is not None else []

    @property
    def size(self):
        self.initialize()
        return sum(len(ss) for ss in self.s)


class TracerCorrelationFunctionMultipolesObservable(BaseCalculator):
    """Correlation function multipoles observable (correlation_function.py:20-120): ``data`` (flat array or dict of parameters),
    ``covariance``, ``slim`` / ``s`` / ``sedges`` / ``ells`` / ``wmatrix`` / ``theory`` forwarded to the window."""
    name = 'correlation2poles'

    def initialize(self):
        if self._initialized:
            return self
        init = dict(self.init)
        data, covariance, wmatrix = init.pop('data', None), init.pop('covariance', None), init.pop('wmatrix', None)
        self.name = init.pop('name', self.name)
        self.covariance = None if covariance is None else np.asarray(covariance, dtype='f8')
        self.nobs = init.pop('nobs', None)
        self.transform = None
        if isinstance(wmatrix, WindowedCorrelationFunctionMultipoles):
            self.wmatrix = wmatrix
        else:
            self.wmatrix = WindowedCorrelationFunctionMultipoles()
            if wmatrix is not None: self.wmatrix.init.update(wmatrix=wmatrix)
        self._require(self.wmatrix)
        self.wmatrix.init.update(init)
        self.wmatrix.initialize()
        for name in ['s', 'ells', 'sedges']:
            setattr(self, name, getattr(self.wmatrix, name))
        self._data_params = None
        if isinstance(data, dict):
            self._data_params, self.flatdata = dict(data), None
        elif data is None:
            raise ValueError('provide data (flat array or dict of parameters to generate it from theory)')
        else:
            self.flatdata = np.ravel(np.asarray(data, dtype='f8'))
            if self.flatdata.size != self.wmatrix.size:
                raise ValueError('data size {:d} does not match the window output size {:d}'.format(self.flatdata.size, self.wmatrix.size))
        self._initialized = True
        return self

    def _standalone_pipeline(self):
        from ...likelihoods import ObservablesGaussianLikelihood
        self.initialize()
        return ObservablesGaussianLikelihood(observables=[self], precision=np.ones(self.wmatrix.size)), []

    def _standalone_products(self, likelihood):
        """``flattheory`` and ``theory`` (one array per multipole) at the last call (power_spectrum.py:400-404 / correlation_function.py:380-381)."""
        self.flattheory = np.array(likelihood.observable_flattheory(0))
        sizes = [len(xx) for xx in self.wmatrix.s]
        self.theory = [self.flattheory[sum(sizes[:ill]):sum(sizes[:ill + 1])] for ill in range(len(sizes))]

    def _observable_spec(self, flatdata=None):
        self.initialize()
        spec = self.wmatrix.theory._theory_spec()
        spec.update(self.wmatrix._window_spec())
        spec['transform'] = np.array([0], dtype='i4')
        spec['flatdata'] = flatdata if flatdata is not None else self.flatdata
        return spec
