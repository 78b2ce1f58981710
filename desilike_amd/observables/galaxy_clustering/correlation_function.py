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
            if 'wcounts' in wmatrix:
                raise NotImplementedError('RR-count window matrices are out of scope')
            self.ellsin = tuple(self.ells)
            self.sin, matrix_full = window_matrix_bininteg(self.sedges, **wmatrix)
            self.matrix_full = matrix_full.T
        elif isinstance(wmatrix, np.ndarray):   # window.py:667-681: the reference takes the matrix as [input, output] here (transposed w.r.t. P_ell)
            self.ellsin = tuple(ellsin or self.ells)
            self.sin = np.ravel(np.asarray(sin, dtype='f8')).copy()
            self.matrix_full = np.array(wmatrix, dtype='f8').T
        else:
            raise ValueError('unrecognized wmatrix {}'.format(wmatrix))
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
