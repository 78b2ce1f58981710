"""Window-matrix application (reference: desilike/observables/galaxy_clustering/window.py:151-533).

Init-time only: builds ``kin``, ``ellsin``, ``matrix_full`` (binning matrix ``window_matrix_bininteg`` 14-68, or a
user-provided dense matrix 307-321), ``kmask`` (297-305) and the shot-noise bookkeeping (445-457).  The product
``W . (power + sn_in) + offset - sn_out`` (459-473) for a batch is the fp64 MFMA GEMM ``dl_window_gemm``.
"""
import os

import numpy as np

from ...base import BaseCalculator
from ... import utils
from ...utils import window_matrix_bininteg  # noqa: F401
from . import _containers
from ._binning import MultipoleBins


def get_templates(templates, ells=(0, 2, 4), x=None):
    """name -> flat template [sum of the output sizes]; lists are named 'syst_{i}' (window.py:1253-1272); callables take (ell, x)."""
    from collections.abc import Mapping
    if templates is None: templates = {}
    if not isinstance(templates, Mapping):
        if callable(templates) or (isinstance(templates, np.ndarray) and templates.ndim == 1): templates = [templates]
        templates = {'syst_{:d}'.format(i): template for i, template in enumerate(templates)}
    toret = {}
    for name, template in templates.items():
        if x is not None:
            if callable(template):
                template = np.concatenate([template(ell, xx) for ell, xx in zip(ells, x)])
            template = np.ravel(np.asarray(template, dtype='f8'))
            size = sum(xx.size for xx in x)
            if template.size != size:
                raise ValueError('provided template is size {:d}, but expected {:d} = sum({})'.format(template.size, size, [xx.size for xx in x]))
        toret[name] = template
    return toret


class SystematicTemplatePowerSpectrumMultipoles(BaseCalculator):
    """Systematic templates added to the windowed multipoles, ``flatpower += sum_i syst_i template_i`` (window.py:1275-1309, 472-473): one linear parameter
    per template (analytically solvable), carried to the GPU as pass-through columns of the window matrix."""

    @classmethod
    def _default_params(cls, templates=None, **kwargs):
        return {name: dict(value=0., ref=dict(limits=[-1e-3, 1e-3]), delta=0.005, latex='s_{{{:d}}}'.format(i)) for i, name in enumerate(get_templates(templates))}

    def initialize(self):
        if self._initialized:
            return self
        self.ells = tuple(self.init.get('ells', (0, 2, 4)))
        x = self.init.get(self._xname, None)
        if x is None: x = self._default_x()
        if not isinstance(x, (tuple, list)): x = [x] * len(self.ells)
        setattr(self, self._xname, tuple(np.asarray(xx, dtype='f8') for xx in x))
        self.templates = get_templates(self.init.get('templates', ()), ells=self.ells, x=getattr(self, self._xname))
        self._initialized = True
        return self

    _xname = 'k'

    def _default_x(self):
        return np.linspace(0.01, 0.2, 101)


class TopHatFiberCollisionsPowerSpectrumMultipoles(BaseCalculator):
    r"""Fiber-collision kernels of Hahn et al. 2016 (arXiv:1609.01714, appendix) for a top-hat pair-loss function of amplitude ``fs`` below the
    transverse scale ``Dfc`` (reference: window.py:972-1049).  Init-time constants only:
    ``kernel_correlated [n_ell, n_ellin, n_k, n_kin]`` (the multipoles are multiplied by it: identity minus the collided pairs) and
    ``kernel_uncorrelated [n_ell, n_k]`` (an offset); the window folds both into its matrix / offset (window.py:428-438)."""

    def initialize(self):
        if self._initialized:
            return self
        from scipy import special
        init = self.init
        k = init.get('k', None)
        if k is None: k = np.linspace(0.01, 0.2, 101)
        self.k = np.array(k, dtype='f8')
        self.ells = tuple(init.get('ells', (0, 2, 4)))
        theory = init.get('theory', None)
        if theory is None:
            from ...theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles
            theory = KaiserTracerPowerSpectrumMultipoles()
        self.theory = theory
        theory.initialize()
        self.kin, self.ellsin = np.array(theory.k, dtype='f8'), tuple(theory.ells)
        self.with_uncorrelated = bool(init.get('with_uncorrelated', True))
        self.fs, self.Dfc = float(init.get('fs', 1.)), float(init.get('Dfc', 0.))

        def W2D(x):   # Fourier transform of the unit disc
            return 2. * special.j1(x) / x

        # polynomials H_{l l'} of the ratio of the smaller to the larger wavenumber (Hahn et al., appendix), l > l'
        def Hpoly(lmax, lmin, x):
            # (2, 0): x^2 - 1; (4, 0): 7/4 x^4 - 5/2 x^2 + 3/4; (4, 2): x^4 - x^2; (6, 0): 33/8 x^6 - 63/8 x^4 + 35/8 x^2 - 5/8; (6, 2): 11/4 x^6 - 9/2 x^4 + 7/4 x^2; (6, 4): x^6 - x^4
            table = {(2, 0): {2: 1., 0: -1.}, (4, 0): {4: 7. / 4., 2: -5. / 2., 0: 3. / 4.}, (4, 2): {4: 1., 2: -1.},
                     (6, 0): {6: 33. / 8., 4: -63. / 8., 2: 35. / 8., 0: -5. / 8.}, (6, 2): {6: 11. / 4., 4: -9. / 2., 2: 7. / 4.}, (6, 4): {6: 1., 4: -1.}}
            return sum(coeff * x**power for power, coeff in table[(lmax, lmin)].items())

        leg0 = np.array([(2. * ell + 1.) * special.eval_legendre(ell, 0.) for ell in self.ells])
        self.kernel_uncorrelated = -leg0[:, None] * self.fs * (np.pi * self.Dfc)**2 / self.k * W2D(self.k * self.Dfc)
        kk, qq = np.meshgrid(self.k, self.kin, indexing='ij')
        ratio = np.minimum(kk, qq) / np.maximum(kk, qq)
        base = np.minimum(qq / kk, 1.) * W2D(qq * self.Dfc)
        measure = self.kin * utils.weights_trapz(self.kin)
        diag = utils.matrix_lininterp(self.kin, self.k).T
        kernels = np.zeros((len(self.ells), len(self.ellsin), len(self.k), len(self.kin)), dtype='f8')
        for iout, ellout in enumerate(self.ells):
            for iin, ellin in enumerate(self.ellsin):
                if ellin == ellout:
                    fll = base * ratio**ellout
                    kernels[iout, iin] = diag
                else:
                    fll = np.where(((ellout >= ellin) & (kk >= qq)) | ((ellout <= ellin) & (kk <= qq)),
                                   base * (2. * ellout + 1.) / 2. * Hpoly(max(ellout, ellin), min(ellout, ellin), ratio), 0.)
                kernels[iout, iin] -= self.fs * self.Dfc**2 / 2. * fll * measure
        self.kernel_correlated = kernels
        self._initialized = True
        return self


class WindowedPowerSpectrumMultipoles(BaseCalculator):
    """
    Window effect on the power spectrum multipoles.

    Parameters (subset of the reference's, same meaning): ``klim``, ``k``, ``kedges``, ``ells``,
    ``wmatrix`` (None, ``{'resolution': n}``, or 2D array [n_out, n_ellin * n_kin] with ``kin``, ``ellsin``),
    ``kin``, ``kinrebin``, ``kinlim``, ``ellsin``, ``shotnoise``, ``wshotnoise``, ``theory``.
    File / lsstypes / pypower inputs are out of scope.
    """

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        wmatrix, kin, kinrebin, kinlim, ellsin = init.get('wmatrix', None), init.get('kin', None), init.get('kinrebin', 1), init.get('kinlim', None), init.get('ellsin', None)
        shotnoise, wshotnoise = init.get('shotnoise', None), init.get('wshotnoise', None)
        fiber_collisions, systematic_templates = init.get('fiber_collisions', None), init.get('systematic_templates', None)
        if isinstance(wmatrix, (str, os.PathLike)):
            # window matrix from a file (window.py:325-334): array-level containers, see desilike_amd/io.py (lsstypes / pypower objects themselves are not read)
            from ... import io
            loaded = io.load_window(wmatrix)
            wmatrix = loaded.pop('wmatrix')
            if kin is None: kin = loaded['kin']
            if ellsin is None: ellsin = loaded['ellsin']
            if wshotnoise is None: wshotnoise = loaded.get('wshotnoise', None)
            for key in ('k', 'ells'):
                if init.get(key, None) is None and init.get('kedges', None) is None and init.get('klim', None) is None: init[key] = loaded[key]
        # output binning (window.py:214-292): rules in _binning.MultipoleBins
        bins = MultipoleBins.resolve(x=init.get('k', None), edges=init.get('kedges', None), lim=init.get('klim', None), ells=init.get('ells', None),
                                     default_step=0.01, default_edges=np.arange(0.005, 0.21, 0.01), label='k')
        self.ells, self.k, self.kedges, self.kmasklim = bins.ells, bins.x, bins.edges, bins.masklim

        theory = init.get('theory', None)
        if theory is None:
            from ...theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles
            theory = self.init['theory'] = KaiserTracerPowerSpectrumMultipoles()
        self.theory = self._require(theory)

        self.matrix_full, self.kmask, self.offset = None, None, None
        if wmatrix is None:  # window.py:294-305
            self.ellsin = tuple(self.ells)
            self.kin, self.kmask = bins.input_grid()
        elif isinstance(wmatrix, dict):  # window.py:306-310
            self.ellsin = tuple(self.ells)
            self.kin, matrix_full = window_matrix_bininteg(self.kedges, **wmatrix)
            self.matrix_full = matrix_full.T
        elif isinstance(wmatrix, np.ndarray):  # window.py:311-324: dense matrix given on (ellsin, kin), optionally rebinned / cut along its input axis
            self.ellsin = tuple(ellsin or self.ells)
            matrix_full = np.array(wmatrix, dtype='f8')
            if matrix_full.shape[0] != bins.size:
                raise ValueError('output "wmatrix" size is {:d}, but got {:d} output "k"'.format(matrix_full.shape[0], bins.size))
            kin_given = np.ravel(np.asarray(kin, dtype='f8'))
            self.kin = kin_given[::kinrebin] if kinrebin is not None else kin_given.copy()
            if kinlim is not None: self.kin = self.kin[(self.kin >= kinlim[0]) & (self.kin <= kinlim[-1])]
            rebin = utils.matrix_lininterp(self.kin, kin_given)                     # [len(kin), len(kin_given)], the same for every input multipole
            blocks = matrix_full.reshape(matrix_full.shape[0], len(self.ellsin), kin_given.size)
            self.matrix_full = np.einsum('oli,ki->olk', blocks, rebin).reshape(matrix_full.shape[0], -1)
        elif _containers.is_matrix_container(wmatrix):   # window.py:337-352: an lsstypes-like WindowMatrix (duck-typed: .value(), .theory, .observable)
            # with ``kin`` every input multipole is rebinned from its own grid (window.py:347-349); without, they must share one (350-351)
            self.matrix_full, self.kin, self.ellsin = _containers.read_window(wmatrix, self.ells, self.k, ellsin=ellsin, kin=kin)
        else:
            raise NotImplementedError('window matrix of type {}: pass a 2D array with kin and ellsin, a file written by desilike_amd.io.save_window, or an object with '
                                      '.value(), .theory, .observable (desilike_amd/observables/galaxy_clustering/_containers.py)'.format(type(wmatrix).__name__))
        if fiber_collisions is not None:   # window.py:428-438: kernels folded into the matrix / offset
            self.theory.init.update(k=self.kin, ells=self.ellsin)
            fiber_collisions.init.update(k=self.kin, ells=self.ellsin, theory=self.theory)
            fiber_collisions.initialize()
            kc = fiber_collisions.kernel_correlated
            kernel = np.block([[kc[iout, iin] for iin in range(kc.shape[1])] for iout in range(kc.shape[0])])
            uncorrelated = fiber_collisions.kernel_uncorrelated.ravel() if fiber_collisions.with_uncorrelated else None
            if self.matrix_full is None:
                self.offset, self.matrix_full = uncorrelated, kernel
            else:
                if uncorrelated is not None: self.offset = self.matrix_full.dot(uncorrelated)
                self.matrix_full = self.matrix_full.dot(kernel)
            self.ellsin, self.kin = fiber_collisions.ellsin, fiber_collisions.kin
        if systematic_templates is not None:   # window.py:439-443
            if not isinstance(systematic_templates, SystematicTemplatePowerSpectrumMultipoles):
                systematic_templates = SystematicTemplatePowerSpectrumMultipoles(templates=systematic_templates)
            systematic_templates.init.update(k=self.k, ells=self.ells)
            systematic_templates.initialize()
        self.systematic_templates = systematic_templates
        self.theory.init.update(k=self.kin, ells=self.ellsin)
        if shotnoise is None:
            shotnoise = 0.
        else:
            shotnoise = float(shotnoise)
            if getattr(self.theory, '_kind', None) == 3 or getattr(self.theory, '_wants_shotnoise', False):   # theories using the shot noise take it from the observable (window.py:441-443)
                self.theory.init.setdefault('shotnoise', shotnoise)
        self.shotnoise = shotnoise
        # shot noise (window.py:445-457): added to the monopole of the theory before the window, removed from the monopole rows after it; a window
        # that carries its own response to a constant (``wshotnoise``) takes both roles
        monopole_rows = np.concatenate([np.full(len(kk), float(ell == 0)) for ell, kk in zip(self.ells, self.k)])
        if wshotnoise is None:
            self.shotnoisein = shotnoise * np.array([float(ell == 0) for ell in self.ellsin])
            self.shotnoiseout = shotnoise * monopole_rows
        else:
            self.shotnoisein = np.zeros(len(self.ellsin), dtype='f8')
            self.shotnoiseout = shotnoise * (monopole_rows - np.asarray(wshotnoise, dtype='f8'))
        self.wshotnoise = wshotnoise
        self.theory.initialize()
        self._initialized = True
        return self

    def _window_spec(self):
        self.initialize()
        wmatrix, offset, shotnoisein = self.matrix_full, self.offset, self.shotnoisein
        fold = getattr(self.theory, '_fold', None)
        if fold is not None:   # theories whose constant linear tail (broadband terms, emulator output layer, ...) is folded into the window matrix
            fold = fold()
            if np.any(shotnoisein != 0.):   # W . (sn_in (x) 1) (window.py:471) no longer maps onto device columns: fold it into the offset
                vector = np.repeat(shotnoisein, len(self.kin))
                extra = vector if wmatrix is None else wmatrix.dot(vector)
                offset = extra if offset is None else offset + extra
            shotnoisein = None
            wmatrix = fold if wmatrix is None else wmatrix.dot(fold)
        wmatrix = _append_systematic_templates(wmatrix, self.systematic_templates, self.kmask, len(self.ellsin) * len(self.kin))
        return dict(wmatrix=wmatrix, kmask=None if self.kmask is None else np.asarray(self.kmask, dtype='i4'), offset=offset,
                    shotnoise_in=shotnoisein, shotnoise_out=self.shotnoiseout)

    def _pass_params(self):
        """Names of the parameters the window adds as pass-through columns (systematic templates)."""
        self.initialize()
        return list(self.systematic_templates.templates) if self.systematic_templates is not None else []

    def _extra_params(self):
        self.initialize()
        return list(self.systematic_templates.params) if self.systematic_templates is not None else []

    @property
    def size(self):
        self.initialize()
        return sum(len(kk) for kk in self.k)


def _append_systematic_templates(wmatrix, systematic_templates, mask, n_in):
    """Window matrix with one more pass-through column per systematic template.  The templates live on the OUTPUT grid (after the row selection ``mask``):
    their values are scattered to the selected rows of the un-masked matrix."""
    if systematic_templates is None or not systematic_templates.templates:
        return wmatrix
    if wmatrix is None: wmatrix = np.eye(n_in)
    columns = np.zeros((wmatrix.shape[0], len(systematic_templates.templates)), dtype='f8')
    rows = np.arange(wmatrix.shape[0]) if mask is None else np.asarray(mask)
    for icol, template in enumerate(systematic_templates.templates.values()):
        columns[rows, icol] = template
    return np.hstack([wmatrix, columns])
