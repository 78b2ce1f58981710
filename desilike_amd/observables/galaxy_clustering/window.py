"""Window-matrix application (reference: desilike/observables/galaxy_clustering/window.py:151-533).

Init-time only: builds ``kin``, ``ellsin``, ``matrix_full`` (binning matrix ``window_matrix_bininteg`` 14-68, or a
user-provided dense matrix 307-321), ``kmask`` (297-305) and the shot-noise bookkeeping (445-457).  The product
``W . (power + sn_in) + offset - sn_out`` (459-473) for a batch is the fp64 MFMA GEMM ``dl_window_gemm``.
"""
import numpy as np

from ...base import BaseCalculator
from ... import utils
from ...utils import window_matrix_bininteg  # noqa: F401


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
        _default_step = 0.01
        klim, k, kedges, ells = init.get('klim', None), init.get('k', None), init.get('kedges', None), init.get('ells', None)
        wmatrix, kin, kinrebin, kinlim, ellsin = init.get('wmatrix', None), init.get('kin', None), init.get('kinrebin', 1), init.get('kinlim', None), init.get('ellsin', None)
        shotnoise, wshotnoise = init.get('shotnoise', None), init.get('wshotnoise', None)
        if ells is None:
            ells = list(klim) if klim is not None else (0, 2, 4)
        self.ells = tuple(ells)
        self.k = self.kmasklim = self.kedges = None
        if k is not None:
            if np.ndim(k[0]) == 0: k = [k] * len(self.ells)
            self.k = [np.array(kk, dtype='f8') for kk in k]
            if len(self.k) != len(self.ells): raise ValueError("provide as many k's as ells")
        input_klim = klim is not None
        if kedges is not None:
            if np.ndim(kedges[0]) == 0: kedges = [kedges] * len(self.ells)
            self.kedges = [np.array(kk, dtype='f8') for kk in kedges]
            self.kedges = [np.column_stack([edges[:-1], edges[1:]]) if edges.ndim <= 1 else edges for edges in self.kedges]
            if len(self.kedges) != len(self.ells): raise ValueError('provide as many kedges as ells')
            if klim is None:
                klim = {ell: (edges[0, 0], edges[-1, 1], np.mean(edges[..., 1] - edges[..., 0])) for ell, edges in zip(self.ells, self.kedges)}
        if input_klim:  # window.py:249-282
            klim = dict(klim)
            if self.k is not None:
                k, ells, self.kmasklim = [], [], {}
                for ill, ell in enumerate(self.ells):
                    kk = self.k[ill]
                    self.kmasklim[ell] = np.zeros(len(kk), dtype='?')
                    if ell not in klim: continue
                    self.kmasklim[ell][...] = True
                    if klim[ell] is not None:
                        (lo, hi, *step) = klim[ell]
                        kmask = (kk >= lo) & (kk <= hi)
                        kk = kk[kmask]
                        self.kmasklim[ell][...] = kmask
                    if kk.size:
                        k.append(kk)
                        ells.append(ell)
                self.k, self.ells = k, tuple(ells)
            elif list(self.ells) != list(klim):
                raise ValueError('incompatible ells = {} and klim = {}; just remove ells?'.format(self.ells, list(klim)))
            kedges = []
            for ill, ell in enumerate(self.ells):
                if klim[ell] is None:
                    kedges = None
                    break
                (lo, hi, *step) = klim[ell]
                if not step:
                    step = ((hi - lo) / self.k[ill].size,) if self.k is not None else (_default_step,)
                edges = np.arange(lo, hi + step[0] / 2., step=step[0])
                kedges.append(np.column_stack([edges[:-1], edges[1:]]))
            if self.kedges is None: self.kedges = kedges
        if self.kedges is None:
            if self.k is not None:
                self.kedges = []
                for xx in self.k:
                    tmp = (xx[:-1] + xx[1:]) / 2.
                    tmp = np.concatenate([[tmp[0] - (xx[1] - xx[0])], tmp, [tmp[-1] + (xx[-1] - xx[-2])]])
                    self.kedges.append(np.column_stack([tmp[:-1], tmp[1:]]))
            else:
                edges = np.arange(0.01 - _default_step / 2., 0.2 + _default_step, _default_step)
                self.kedges = [np.column_stack([edges[:-1], edges[1:]])] * len(self.ells)
        if self.k is None:
            self.k = [np.mean(edges, axis=-1) for edges in self.kedges]
        self.k = [np.array(kk) for kk in self.k]

        theory = init.get('theory', None)
        if theory is None:
            from ...theories.galaxy_clustering import KaiserTracerPowerSpectrumMultipoles
            theory = self.init['theory'] = KaiserTracerPowerSpectrumMultipoles()
        self.theory = self._require(theory)

        self.matrix_full, self.kmask, self.offset = None, None, None
        if wmatrix is None:  # window.py:294-305
            self.ellsin = tuple(self.ells)
            self.kin = np.unique(np.concatenate(self.k, axis=0))
            if not all(kk.shape == self.kin.shape and np.allclose(kk, self.kin) for kk in self.k):
                kmask = [np.searchsorted(self.kin, kk, side='left') for kk in self.k]
                assert all(np.allclose(self.kin[km], kk) for kk, km in zip(self.k, kmask)), self.k
                self.kmask = np.concatenate([self.kin.size * i + km for i, km in enumerate(kmask)], axis=0)
        elif isinstance(wmatrix, dict):  # window.py:306-310
            self.ellsin = tuple(self.ells)
            self.kin, matrix_full = window_matrix_bininteg(self.kedges, **wmatrix)
            self.matrix_full = matrix_full.T
        elif isinstance(wmatrix, np.ndarray):  # window.py:311-324
            from scipy import linalg
            self.ellsin = tuple(ellsin or self.ells)
            matrix_full = np.array(wmatrix, dtype='f8')
            ksize = sum(len(kk) for kk in self.k)
            if matrix_full.shape[0] != ksize:
                raise ValueError('output "wmatrix" size is {:d}, but got {:d} output "k"'.format(matrix_full.shape[0], ksize))
            kin = np.asarray(kin).flatten()
            self.kin = kin.copy()
            if kinrebin is not None: self.kin = self.kin[::kinrebin]
            if kinlim is not None: self.kin = self.kin[(self.kin >= kinlim[0]) & (self.kin <= kinlim[-1])]
            wmatrix_rebin = linalg.block_diag(*[utils.matrix_lininterp(self.kin, kin) for ell in self.ellsin])
            self.matrix_full = matrix_full.dot(wmatrix_rebin.T)
        else:
            raise NotImplementedError('window matrices from files / lsstypes / pypower objects are out of scope: pass a 2D array with kin and ellsin')
        self.theory.init.update(k=self.kin, ells=self.ellsin)
        if shotnoise is None:
            shotnoise = 0.
        else:
            shotnoise = float(shotnoise)
            if getattr(self.theory, '_kind', None) == 3:   # theories scaling their stochastic terms by the shot noise take it from the observable (window.py:441-443)
                self.theory.init.setdefault('shotnoise', shotnoise)
        self.shotnoise = shotnoise
        # window.py:445-457
        self.shotnoisein = np.array([shotnoise * (ell == 0) for ell in self.ellsin], dtype='f8')
        wshotnoisebase = np.concatenate([np.full_like(kk, (ell == 0), dtype='f8') for ell, kk in zip(self.ells, self.k)])
        self.shotnoiseout = shotnoise * wshotnoisebase
        if wshotnoise is not None:
            self.shotnoisein[...] = 0.
            self.shotnoiseout[...] = shotnoise * (wshotnoisebase - np.asarray(wshotnoise))
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
        return dict(wmatrix=wmatrix, kmask=None if self.kmask is None else np.asarray(self.kmask, dtype='i4'), offset=offset,
                    shotnoise_in=shotnoisein, shotnoise_out=self.shotnoiseout)

    @property
    def size(self):
        self.initialize()
        return sum(len(kk) for kk in self.k)
