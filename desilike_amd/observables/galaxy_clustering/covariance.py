"""Gaussian covariance matrix of power-spectrum / correlation-function multipoles (reference: desilike/observables/galaxy_clustering/covariance.py).

``ObservablesCovarianceMatrix(observables, footprints, resolution)(**params)`` -> covariance [n, n], as in the reference; ``evaluate_batch(theta)`` gives the
matrices of a whole batch of parameter points ``[B, n, n]`` (forecast loops): the theory multipoles come from the device theory kernels (``dl_eval_theory``) and stay
in HBM, the matrix elements are formed by ``dl_cov_apply`` (csrc/dl_cov.hip) from a plan -- cells of the matrix and their integration points -- that depends on
the binning only and is built here once.  Like the reference: one tracer, no cross-correlation of different tracers (covariance.py:277)."""
import numpy as np
from scipy import special

from .power_spectrum import TracerPowerSpectrumMultipolesObservable
from .correlation_function import TracerCorrelationFunctionMultipolesObservable


def integral_legendre_product(ells, range=(-1., 1.), norm=False):
    """Integral over mu of the product of the Legendre polynomials of orders ``ells`` (covariance.py:14-41)."""
    poly = np.poly1d([1.])
    for ell in np.atleast_1d(ells):
        poly = poly * special.legendre(int(ell))
    integ = poly.integ()
    toret = integ(range[-1]) - integ(range[0])
    return toret / (range[-1] - range[0]) if norm else toret


class BaseFootprint(object):
    """Density and volume of a survey (covariance.py:54-117): ``nbar`` [(h / Mpc)^3] or ``size`` (number of objects), ``volume`` [(Mpc / h)^3]."""

    def __init__(self, nbar=None, size=None, volume=None, attrs=None):
        if nbar is None and size is None:
            raise ValueError('provide either "size" (number of objects) or "nbar" (mean comoving density in (Mpc/h)^(-3))')
        if volume is None:
            raise ValueError('provide volume')
        self._nbar, self._size, self._volume = (None if value is None else np.asarray(value, dtype='f8') for value in (nbar, size, volume))
        if self._nbar is None: self._nbar = self._size / self._volume
        self.attrs = dict(attrs or {})

    @property
    def volume(self):
        return self._volume

    @property
    def size(self):
        return self._size if self._size is not None else self._nbar * self.volume

    @property
    def shotnoise(self):
        return self.volume / self.size

    def __and__(self, other):
        """Intersection (covariance.py:99-101)."""
        return self.__class__(nbar=self._nbar + other._nbar, volume=min(self.volume, other.volume))

    def copy(self):
        import copy
        return copy.deepcopy(self)


class BoxFootprint(BaseFootprint):
    """Box footprint."""


class CutskyFootprint(BaseFootprint):
    """Cut-sky footprint (covariance.py:123-271): surface ``area`` [deg^2], redshift range ``zrange`` (two values, or the redshifts where ``nbar`` [(h / Mpc)^3] is
    tabulated), ``nbar`` (scalar: surface density [deg^-2]) or ``size``.  ``cosmo``: anything with ``comoving_radial_distance(z)`` [Mpc / h] (the reference takes a
    cosmoprimo cosmology: out of scope here), or that function itself."""

    def __init__(self, nbar=None, size=None, area=None, zrange=None, cosmo=None, attrs=None):
        if nbar is None and size is None:
            raise ValueError('provide either "size" (number of objects) or "nbar" (angular density in (deg)^(-2))')
        if area is None or zrange is None:
            raise ValueError('provide area (in deg^2) and zrange (zmin, zmax)')
        for name, value in [('area', area), ('zrange', zrange), ('nbar', nbar)]:
            value = np.asarray(value if value is not None else np.nan, dtype='f8').flatten()
            if value.size <= 1: value = value.reshape(())
            setattr(self, '_' + name, value)
        self._size = size
        self.cosmo = cosmo
        self.attrs = dict(attrs or {})

    def _distance(self, z):
        if self.cosmo is None: raise ValueError('Provide cosmology')
        func = getattr(self.cosmo, 'comoving_radial_distance', self.cosmo)
        return np.asarray(func(z), dtype='f8')

    def _shells(self):
        return np.diff(self._distance(self._zrange)**3)

    def _nbar_shells(self):
        return self._nbar if self._nbar.size == self._zrange.size - 1 else (self._nbar[:-1] + self._nbar[1:]) / 2.

    @property
    def area(self):
        if self._area.ndim == 0: return self._area
        return np.mean(self._area) * (180. / np.pi)**2 * (4. * np.pi)

    @property
    def volume(self):
        return self.area / (180. / np.pi)**2 / 3. * self._shells().sum()

    @property
    def zavg(self):
        z = (self._zrange[:-1] + self._zrange[1:]) / 2.
        return np.average(z, weights=self._nbar_shells() * self._shells()) if self._nbar.ndim else np.mean(z)

    @property
    def zeff(self):
        z = (self._zrange[:-1] + self._zrange[1:]) / 2.
        return np.average(z, weights=self._nbar_shells()**2 * self._shells()) if self._nbar.ndim else np.mean(z)

    @property
    def size(self):
        if self._size is not None: return self._size
        if self._nbar.ndim: return self.area / (180. / np.pi)**2 / 3. * np.sum(self._nbar_shells() * self._shells())
        return self.area * self._nbar

    def __and__(self, other):
        raise NotImplementedError('intersection of two cut-sky footprints: give both observables the same footprint (one tracer: covariance.py:277)')


def _interp_coefficients(xq, xp):
    """np.interp(xq, xp, fp) = slope_j (xq - xp_j) + fp_j: interval j, (xq - xp_j), (xp_j+1 - xp_j); outside the grid the end value stands (a = 0 or the full interval)."""
    xq, xp = np.asarray(xq, dtype='f8'), np.asarray(xp, dtype='f8')
    j = np.clip(np.searchsorted(xp, xq, side='right') - 1, 0, xp.size - 2)
    h = xp[j + 1] - xp[j]
    a = np.clip(xq - xp[j], 0., h)
    return j.astype(np.int32), a, h


class CovariancePlanBuilder(object):
    """Cells and integration points of the blocks of covariance.py:355-456 for a list of observables (dicts: kind 'pk' | 'xi', ells, edges per multipole [n, 2], volume,
    shotnoise, theory k grid, theory ells): the host arrays ``dl_cov_create`` takes."""

    def __init__(self, observables, resolution=1):
        self.observables, self.resolution = observables, int(resolution)
        if self.resolution <= 0: raise ValueError('resolution must be a strictly positive integer')
        self.cell_i, self.cell_d, self.pt_i, self.pt_d, self.gtab, self.sym = [], [], [], [], [], []
        self._gkeys = {}
        sizes = [sum(len(e) for e in obs['edges']) for obs in observables]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)])
        self.n = int(self.offsets[-1])
        for io1 in range(len(observables)):
            for io2 in range(io1 + 1):
                self._block(io1, io2)

    def _gindex(self, obs1, obs2, ell1, ell2):
        key = (tuple(obs1['theory_ells']), tuple(obs2['theory_ells']), ell1, ell2)
        if key not in self._gkeys:
            table = np.zeros((5, 5))
            for ia, la in enumerate(obs1['theory_ells']):
                for ib, lb in enumerate(obs2['theory_ells']):
                    table[ia, ib] = integral_legendre_product((la, lb, ell1, ell2))
            self._gkeys[key] = len(self.gtab)
            self.gtab.append(table)
        return self._gkeys[key]

    def _add_cell(self, row, col, io1, io2, gindex, zero_lag, prefactor, front, den, const, j1, a1, h1, j2, a2, h2, w, w2):
        first = len(self.pt_i)
        for q in range(len(j1)):
            self.pt_i.append((j1[q], j2[q]))
            self.pt_d.append((a1[q], h1[q], a2[q], h2[q], w[q], w2[q]))
        self.cell_i.append((row, col, io1, io2, gindex, int(zero_lag), first, len(j1)))
        self.cell_d.append((prefactor, front, den, const))

    def _block(self, io1, io2):
        obs1, obs2 = self.observables[io1], self.observables[io2]
        start = len(self.cell_i)
        transpose = obs1['kind'] == 'pk' and obs2['kind'] == 'xi'          # covariance.py:420-421: computed as the (xi, pk) block, transposed
        if transpose: obs1, obs2, io1, io2 = obs2, obs1, io2, io1
        kinds = (obs1['kind'], obs2['kind'])
        volume = min(obs1['volume'], obs2['volume'])
        res = self.resolution

        def bin_volume(edges): return 4. / 3. * np.pi * (edges[1]**3 - edges[0]**3)

        def integ_points(edges): return np.linspace(edges[0], edges[1], res + 2)[1:-1]

        if kinds == ('xi', 'xi'):
            ks = [np.asarray(obs['theory_k'], dtype='f8') for obs in (obs1, obs2)]
            k = np.unique(np.concatenate(ks))
            k = k[(k >= max(kk.min() for kk in ks)) & (k <= min(kk.max() for kk in ks))]
            shell = 4. * np.pi * k**2 * np.concatenate([[k[1] - k[0]], k[2:] - k[:-2], [k[-1] - k[-2]]]) / 2.       # covariance.py:388-390 with utils.weights_trapz
            interp = [_interp_coefficients(k, kk) for kk in ks]
            sbar = {}
        row0, col0 = self.offsets[io1], self.offsets[io2]
        r0 = row0
        for ill1, ell1 in enumerate(obs1['ells']):
            c0 = col0
            for ill2, ell2 in enumerate(obs2['ells']):
                gindex = self._gindex(obs1, obs2, ell1, ell2)
                prefactor = (2 * ell1 + 1) * (2 * ell2 + 1) / volume
                for i1, bin1 in enumerate(obs1['edges'][ill1]):
                    for i2, bin2 in enumerate(obs2['edges'][ill2]):
                        row, col = r0 + i1, c0 + i2
                        if transpose: row, col = col, row
                        inter = (max(bin1[0], bin2[0]), min(bin1[1], bin2[1]))
                        if kinds == ('pk', 'pk'):                                             # covariance.py:397-406
                            if inter[0] >= inter[1]: continue
                            k = integ_points(inter)
                            j1, a1, h1 = _interp_coefficients(k, obs1['theory_k']); j2, a2, h2 = _interp_coefficients(k, obs2['theory_k'])
                            front = (2. * np.pi)**3 * bin_volume(inter) / np.prod([bin_volume(bin1), bin_volume(bin2)])
                            self._add_cell(row, col, io1, io2, gindex, False, prefactor, front, np.sum(k**2), 0., j1, a1, h1, j2, a2, h2, k**2, np.ones_like(k))
                        elif kinds == ('xi', 'pk'):                                           # covariance.py:408-416
                            s, k = integ_points(bin1), integ_points(bin2)
                            weights = np.sum(s[:, None]**2 * special.spherical_jn(ell1, s[:, None] * k), axis=0) / np.sum(s**2, axis=0)
                            j1, a1, h1 = _interp_coefficients(k, obs1['theory_k']); j2, a2, h2 = _interp_coefficients(k, obs2['theory_k'])
                            self._add_cell(row, col, io1, io2, gindex, False, prefactor, np.sign(1j**ell1).real, np.sum(k**2), 0., j1, a1, h1, j2, a2, h2, k**2, weights)
                        else:                                                                 # xi x xi: covariance.py:423-446
                            for key, s, ell in [((0, ill1, i1), integ_points(bin1), ell1), ((1, ill2, i2), integ_points(bin2), ell2)]:
                                if key not in sbar: sbar[key] = np.sum(s[:, None]**2 * special.spherical_jn(ell, s[:, None] * k), axis=0) / np.sum(s**2, axis=0)
                            weights = np.prod([sbar[(0, ill1, i1)], sbar[(1, ill2, i2)]], axis=0)
                            sign = np.sign(1j**(ell1 + ell2)).real
                            const = 0.
                            if inter[0] < inter[1]:
                                sn = integral_legendre_product((0, 0, ell1, ell2)) * obs1['shotnoise'] * obs2['shotnoise'] * (2 * ell1 + 1) * (2 * ell2 + 1) / volume
                                const = sign * bin_volume(inter) / np.prod([bin_volume(bin1), bin_volume(bin2)]) * sn
                            self._add_cell(row, col, io1, io2, gindex, True, prefactor, sign / (2. * np.pi)**3, 1., const, *interp[0], *interp[1], shell, weights)
                c0 += len(obs2['edges'][ill2])
            r0 += len(obs1['edges'][ill1])
        if io1 == io2:
            lo, hi = self.offsets[io1], self.offsets[io1 + 1]
            self.sym.extend((r, c) for r in range(lo, hi) for c in range(r + 1, hi))
        else:
            # the transposed block (covariance.py:352-353): the same cells with row and column exchanged (they share their points)
            for ci, cd in list(zip(self.cell_i, self.cell_d))[start:]:
                self.cell_i.append((ci[1], ci[0]) + tuple(ci[2:]))
                self.cell_d.append(cd)

    def arrays(self):
        def arr(rows, width, dtype):
            return np.array(rows, dtype=dtype).reshape(-1, width)
        return dict(n=self.n, cell_i=arr(self.cell_i, 8, np.int32), cell_d=arr(self.cell_d, 4, 'f8'), pt_i=arr(self.pt_i, 2, np.int32), pt_d=arr(self.pt_d, 6, 'f8'),
                    gtab=np.array(self.gtab, dtype='f8').reshape(-1, 5, 5), sym=arr(self.sym, 2, np.int32))


class ObservablesCovarianceMatrix(object):
    """Gaussian covariance matrix of the input observables (covariance.py:274-456): ``observables`` (one or a list of ``TracerPowerSpectrumMultipolesObservable`` /
    ``TracerCorrelationFunctionMultipolesObservable``), ``footprints`` (one for all, or one per observable), ``resolution``: integration points per bin.
    ``covariance = ObservablesCovarianceMatrix(...)(**params)``."""

    def __init__(self, observables, footprints=None, theories=None, resolution=1, device=0):
        from ...likelihoods import ObservablesGaussianLikelihood
        if theories is not None:
            raise NotImplementedError('theories are taken from the observables (their device theory kernels evaluate the multipoles)')
        self.observables = list(observables) if isinstance(observables, (list, tuple)) else [observables]
        footprints = list(footprints) if isinstance(footprints, (list, tuple)) else [footprints] * len(self.observables)
        if any(footprint is None for footprint in footprints): raise ValueError('provide footprints')
        self.footprints = [footprint.copy() for footprint in footprints]
        self.resolution = int(resolution)
        if self.resolution <= 0: raise ValueError('resolution must be a strictly positive integer')
        self.device = int(device)
        # the device pipeline of the observables: a likelihood with unit precision (only its theory kernels are used)
        for obs in self.observables: obs.initialize()
        size = sum(obs.wmatrix.size for obs in self.observables)
        self._likelihood = ObservablesGaussianLikelihood(observables=self.observables, precision=np.ones(size), device=self.device)
        self._plan = None

    @property
    def varied_params(self):
        return self._likelihood.varied_params

    @property
    def all_params(self):
        return self._likelihood.all_params

    def _descriptions(self):
        out = []
        for obs, footprint in zip(self.observables, self.footprints):
            wm = obs.wmatrix
            theory = wm.theory
            pk = isinstance(obs, TracerPowerSpectrumMultipolesObservable)
            out.append(dict(kind='pk' if pk else 'xi', ells=tuple(wm.ells), edges=[np.asarray(e, dtype='f8') for e in (wm.kedges if pk else wm.sedges)], volume=float(footprint.volume),
                            shotnoise=float(footprint.shotnoise), theory_k=np.asarray(theory.k, dtype='f8'), theory_ells=tuple(theory.ells)))
        return out

    def _get_plan(self):
        if self._plan is None:
            from ..._lib import CovariancePlan
            desc = self._descriptions()
            arrays = CovariancePlanBuilder(desc, resolution=self.resolution).arrays()
            n_ell = [len(d['theory_ells']) for d in desc]
            n_k = [len(d['theory_k']) for d in desc]
            ell0 = [list(d['theory_ells']).index(0) if 0 in d['theory_ells'] else -1 for d in desc]
            self._plan = CovariancePlan(arrays['n'], n_ell, n_k, ell0, [d['shotnoise'] for d in desc], arrays['cell_i'], arrays['cell_d'], arrays['pt_i'], arrays['pt_d'], arrays['gtab'],
                                        arrays['sym'], device=self.device)
        return self._plan

    def evaluate_batch(self, theta):
        """``theta [B, P]``: device tensor (columns = ``varied_params``) or host array -> covariance matrices ``[B, n, n]`` (same kind of container)."""
        import torch
        is_numpy = not isinstance(theta, torch.Tensor)
        th = torch.as_tensor(np.ascontiguousarray(theta, dtype='f8'), device=torch.device('cuda', self.device)) if is_numpy else theta
        ctx = self._likelihood._get_context()
        plan = self._get_plan()
        powers = []
        for iobs, shape in enumerate(plan.shapes):
            power = torch.empty((th.shape[0],) + shape, dtype=torch.float64, device=th.device)
            ctx.eval_theory(th, power, iobs=iobs)
            powers.append(power)
        out = plan.apply(powers)
        return out.cpu().numpy() if is_numpy else out

    def __call__(self, **params):
        values = [float(params.pop(param.name, params.pop(param.basename, param.value))) for param in self.varied_params]
        if params: raise ValueError('unknown parameters {}'.format(list(params)))
        self.covariance = self.evaluate_batch(np.array([values]))[0]
        return self.covariance
