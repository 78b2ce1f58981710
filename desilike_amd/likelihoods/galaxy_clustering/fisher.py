"""Likelihood for Fisher forecasts (reference: desilike/likelihoods/galaxy_clustering/fisher.py:10-71): the anisotropic power spectrum P(k, mu) of each theory on a
linear grid of wavenumbers x Gauss-Legendre cosines against a fiducial one, with the diagonal precision of the Gaussian covariance of P(k, mu),

    precision(k, mu) = 4 pi / (2 (2 pi)^3) w_mu V k^2 dk / (P(k, mu) + 1 / nbar)^2 .

On the device this is the ordinary path with a special "window": P(k_i, mu_j) = sum_ell P_ell(k_i) L_ell(mu_j) is LINEAR in the theory multipoles, so the synthesis
matrix [n_k n_mu, n_ell n_k] stands where an observable's window matrix stands, the precision is its diagonal form, and everything downstream (chi2, priors,
Fisher matrices, gradients, samplers) is the code of every other likelihood."""
import types

import numpy as np

from ...base import BaseCalculator
from ... import utils
from ..base import ObservablesGaussianLikelihood


class _SignalToNoiseObservable(BaseCalculator):
    """P(k, mu) of one theory, flattened k-major as the reference's ``pkmu.ravel()`` (fisher.py:58-66); plays the observable's part for the Gaussian likelihood."""
    name = 'spectrum2d'

    def initialize(self):
        if self._initialized: return self
        init = self.init
        self.theory, self.mu = init['theory'], np.asarray(init['mu'], dtype='f8')
        self._require(self.theory)
        k = init.get('k', None)
        if k is not None: self.theory.init.update(k=np.asarray(k, dtype='f8'))
        self.theory.initialize()
        self.k, self.ells = np.asarray(self.theory.k, dtype='f8'), tuple(self.theory.ells)
        nk, nmu = self.k.size, self.mu.size
        synthesis = np.zeros((nk, nmu, len(self.ells), nk), dtype='f8')           # fisher.py:60-64
        for ill, ell in enumerate(self.ells):
            synthesis[np.arange(nk), :, ill, np.arange(nk)] = utils.legendre(ell, self.mu)[None, :]
        self.matrix = synthesis.reshape(nk * nmu, len(self.ells) * nk)
        self.wmatrix = types.SimpleNamespace(size=nk * nmu, theory=self.theory, k=[self.k] * nmu)
        self.covariance, self.nobs, self.transform = None, None, None
        data = init.get('data', None)
        self._data_params, self.flatdata = (dict(data), None) if isinstance(data, dict) or data is None else (None, np.ravel(np.asarray(data, dtype='f8')))
        if self._data_params is None and self.flatdata.size != nk * nmu: raise ValueError('data size {:d} does not match {:d} x {:d}'.format(self.flatdata.size, nk, nmu))
        self._initialized = True
        return self

    @property
    def theory_calculator(self):
        self.initialize()
        return self.theory

    def _observable_spec(self, flatdata=None):
        self.initialize()
        spec = self.theory._theory_spec()
        spec.update(dict(wmatrix=self.matrix, kmask=None, offset=None, shotnoise_in=None, shotnoise_out=np.zeros(self.matrix.shape[0], dtype='f8')))
        spec['transform'] = np.array([0], dtype='i4')
        spec['flatdata'] = flatdata if flatdata is not None else self.flatdata
        return spec


class SNWeightedPowerSpectrumLikelihood(ObservablesGaussianLikelihood):
    """
    Likelihood for Fisher forecasts, integrating the anisotropic signal-to-noise over the cosine angle to the line of sight and the wavenumber.

    Parameters (the reference's, fisher.py:15-36)
    ----------
    theories : list, theory calculator
    data : dict, default=None
        Parameters passed to ``theories`` to generate the fiducial measurement.
    covariance : dict, default=None
        Parameters passed to ``theories`` to generate the fiducial covariance; defaults to ``data``.
    footprints : list, BaseFootprint
        (One or a list of) footprints: ``volume`` and ``shotnoise`` are used.
    klim : tuple
        Wavenumber range: 500 linearly spaced wavenumbers (fisher.py:44-46).  Required: the reference's trapezoidal weights are those of this grid (fisher.py:55).
    mu : int, default=20
        Number of Gauss-Legendre cosines in [0, 1].
    """

    def initialize(self):
        if self._initialized: return self
        init = self.init
        theories, footprints = init['theories'], init.get('footprints', None)
        if not isinstance(theories, (list, tuple)): theories = [theories]
        if not isinstance(footprints, (list, tuple)): footprints = [footprints] * len(theories)
        if len(footprints) != len(theories) or any(footprint is None for footprint in footprints): raise ValueError('provide one footprint per theory')
        klim = init.get('klim', None)
        if klim is None: raise ValueError('provide klim: the integration weights are those of the grid it defines (fisher.py:44-46, 55)')
        k = np.linspace(*klim, num=500)
        data, covariance = init.get('data', None), init.get('covariance', None)
        self.mu, wmu = utils.weights_mu(mu=init.get('mu', 20))
        self.theories, self.footprints = list(theories), list(footprints)
        device = init.get('device', None)

        def observables(params):
            return [_SignalToNoiseObservable(theory=theory, mu=self.mu, k=k, data=dict(params or {})) for theory in self.theories]

        # the fiducial P(k, mu) that sets the covariance: one evaluation of the theories at ``covariance`` (or ``data``) parameters (fisher.py:50-56)
        fiducial = ObservablesGaussianLikelihood(observables=observables(covariance or data), precision=np.ones(len(self.theories) * k.size * self.mu.size), device=device)
        fiducial.initialize()
        prefactor = 4. * np.pi / (2. * (2. * np.pi)**3) * wmu
        precision, start = [], 0
        for footprint in self.footprints:
            pkmu = fiducial.flatdata[start:start + k.size * self.mu.size].reshape(k.size, self.mu.size)
            start += pkmu.size
            precision.append((prefactor * float(footprint.volume) * (k**2 * utils.weights_trapz(k))[:, None] * (pkmu + float(footprint.shotnoise))**(-2)).ravel())
        for context in getattr(fiducial, '_contexts', {}).values():
            close = getattr(context, 'close', None)
            if close is not None: close()
        init['observables'] = observables(data)
        init['precision'] = np.concatenate(precision)
        init['covariance'] = None          # (here the argument named the parameters of the fiducial covariance: consumed above)
        init['correct_covariance'] = init.get('correct_covariance', 'hartlap-percival2014')
        return super(SNWeightedPowerSpectrumLikelihood, self).initialize()
