"""Tracer power spectrum multipoles observable (reference: desilike/observables/galaxy_clustering/power_spectrum.py:20-121, 400-404)."""
import numpy as np

from ...base import BaseCalculator
from .window import WindowedPowerSpectrumMultipoles
from . import _containers


class TracerPowerSpectrumMultipolesObservable(BaseCalculator):
    """
    Compare a power spectrum measurement to theory.

    Parameters
    ----------
    data : array, dict
        Flat data vector (all multipoles concatenated), or dict of parameter values used to generate a
        mock measurement from the theory itself (power_spectrum.py:86-88).
    covariance : array, default=None
        Covariance of this observable (used when the likelihood is not given a global one).
    klim, kedges, k, ells, wmatrix, shotnoise, theory, ... :
        Forwarded to :class:`WindowedPowerSpectrumMultipoles`.
    transform : str, default=None
        ``None`` or 'cubic' (power_spectrum.py:400-404).
    ``data`` may also be a measurement container or a list of mocks (duck-typed: ``.ells``, ``.get(ells=ell)`` -> ``.coords('k')``, ``.edges('k')``, ``.value()``;
    see :mod:`._containers`): the binning, the shot noise and -- from several mocks -- the covariance default to theirs.
    """
    name = 'spectrum2poles'

    def initialize(self):
        if self._initialized:
            return self
        init = dict(self.init)
        data, covariance = init.pop('data', None), init.pop('covariance', None)
        wmatrix, transform = init.pop('wmatrix', None), init.pop('transform', None)
        self.name = init.pop('name', self.name)
        # ``covariance`` (power_spectrum.py:64-75): a 2-D array, a covariance-matrix container (matched to the observable at the likelihood level), or a LIST OF MOCK
        # MEASUREMENTS whose sample covariance is taken (read below, once the binning is known); mocks passed as ``data`` only give the data vector (their mean)
        self.covariance, self.mocks, self._covariance_container = None, None, None
        cov_items = list(covariance) if isinstance(covariance, (list, tuple)) else None
        cov_mocks = cov_items if cov_items and all(_containers.is_measurement(item) for item in cov_items) else None
        if cov_mocks is None and covariance is not None:
            if _containers.is_matrix_container(covariance): self._covariance_container = covariance
            else: self.covariance = np.asarray(covariance, dtype='f8')
        self.nobs = init.pop('nobs', None)
        if isinstance(wmatrix, WindowedPowerSpectrumMultipoles):
            self.wmatrix = wmatrix
        else:
            self.wmatrix = WindowedPowerSpectrumMultipoles()
            if wmatrix is not None:
                self.wmatrix.init.update(wmatrix=wmatrix)
        self._require(self.wmatrix)
        # measurements given as (lsstypes-like, duck-typed) containers: one object or a list of mocks (power_spectrum.py:123-233) -> flat data vector = their mean;
        # binning and shot noise default to the containers' own
        items = list(data) if isinstance(data, (list, tuple)) else [data]
        if items and all(_containers.is_measurement(item) for item in items):
            klim = init.get('klim', None)
            read = [_containers.read_measurement(item, lim=klim if isinstance(klim, dict) else None, coord='k') for item in items]
            ells, list_k, list_edges = read[0][0], read[0][1], read[0][2]
            for other in read[1:]:
                if other[0] != ells or not all(np.allclose(a, b, rtol=1e-3, atol=0.) for a, b in zip(other[1], list_k)):
                    raise ValueError('the mocks do not share the multipoles / k-bins of the first one')
            if init.get('k', None) is None and init.get('kedges', None) is None:
                init['ells'] = ells
                if all(edges is not None for edges in list_edges): init['kedges'] = list_edges
                else: init['k'] = list_k
                init.pop('klim', None)
            if init.get('shotnoise', None) is None and read[0][4] is not None: init['shotnoise'] = float(np.mean([r[4] for r in read]))
            data = np.array([np.concatenate(r[3]) for r in read]).mean(axis=0)      # power_spectrum.py:218-227: the mean of the measurements
        if cov_mocks is not None:   # power_spectrum.py:69-75: mocks -> sample covariance (ddof = 1); the likelihood takes the number of observations from them (likelihoods/base.py:541-544)
            klim = init.get('klim', None)
            read = [_containers.read_measurement(item, lim=klim if isinstance(klim, dict) else None, coord='k') for item in cov_mocks]
            for other in read[1:]:
                if other[0] != read[0][0] or not all(np.allclose(a, b, rtol=1e-3, atol=0.) for a, b in zip(other[1], read[0][1])):
                    raise ValueError('the mocks do not share the multipoles / k-bins of the first one')
            if init.get('k', None) is None and init.get('kedges', None) is None and init.get('ells', None) is None:     # binning from the mocks when nothing else gives it
                init['ells'] = read[0][0]
                if all(edges is not None for edges in read[0][2]): init['kedges'] = read[0][2]
                else: init['k'] = read[0][1]
                init.pop('klim', None)
            if init.get('shotnoise', None) is None and read[0][4] is not None: init['shotnoise'] = float(np.mean([r[4] for r in read]))     # power_spectrum.py:228-232
            self.mocks = np.array([np.concatenate(r[3]) for r in read])
            self.covariance = np.cov(self.mocks, rowvar=False, ddof=1)
            if self.nobs is None: self.nobs = len(read)
        self.wmatrix.init.update(init)
        self.wmatrix.initialize()
        for name in ['k', 'ells', 'kedges']:
            setattr(self, name, getattr(self.wmatrix, name))
        self.shotnoise = self.wmatrix.shotnoise
        if self._covariance_container is not None:      # power_spectrum.py:115-117: the rows / columns of this observable's bins out of a covariance-matrix container
            container = self._covariance_container
            if getattr(container.observable, 'observables', None): self.covariance = _containers.read_covariance(container, [self])
            else:
                index = _containers._rows_of(container.observable, self.ells, self.k, coord='k')
                self.covariance = np.asarray(container.value(), dtype='f8')[np.ix_(index, index)]
            if self.nobs is None: self.nobs = getattr(container, 'nobs', None)
        self.transform = transform
        if self.transform not in [None, 'cubic']:
            raise ValueError('transform must be one of {}'.format([None, 'cubic']))
        self._data_params = None
        if isinstance(data, dict):
            self._data_params = dict(data)
            self.flatdata = None  # generated from the theory at compile time
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
        sizes = [len(xx) for xx in self.wmatrix.k]
        self.theory = [self.flattheory[sum(sizes[:ill]):sum(sizes[:ill + 1])] for ill in range(len(sizes))]

    @property
    def theory_calculator(self):
        self.initialize()
        return self.wmatrix.theory

    def _observable_spec(self, flatdata=None):
        self.initialize()
        theory = self.wmatrix.theory
        spec = theory._theory_spec()
        spec.update(self.wmatrix._window_spec())
        spec['transform'] = np.array([1 if self.transform == 'cubic' else 0], dtype='i4')
        spec['flatdata'] = flatdata if flatdata is not None else self.flatdata
        return spec
