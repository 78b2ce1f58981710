"""Duck-typed measurement / window / covariance containers.

The reference reads its data vectors, window matrices and covariances from objects of the third-party ``lsstypes`` package (``ObservableTree``, ``WindowMatrix``,
``CovarianceMatrix``; observables/galaxy_clustering/power_spectrum.py:123-233, window.py:325-352, likelihoods/base.py:594-603).  The package is absent from the image
and its classes cannot be instantiated here, but the reference only ever touches a narrow surface of them.  Any object exposing that surface is accepted
(``isinstance`` is never asked):

* a measurement (``data=`` / mocks): ``.ells`` and ``.get(ells=ell)`` -> a pole with ``.coords('k')`` [n], ``.edges('k')`` [n, 2], ``.value()`` [n] and, on the
  monopole, ``.values('shotnoise')`` (power_spectrum.py:165-179);
* a window matrix: ``.value()`` [n_out, n_in], ``.theory`` and ``.observable`` = measurements in the sense above (``.theory``: input multipoles and wavenumbers,
  ``.observable``: output rows), window.py:337-352;
* a covariance: ``.value()`` [n, n] and ``.observable`` with ``.get(observables=name)`` -> a measurement (rows of that observable), likelihoods/base.py:594-603.

Rows are cut to the requested ranges HERE, on the arrays (the reference calls ``.select`` / ``.match`` of the objects): bins whose centre lies in [lo, hi] of ``klim``
(``lsstypes``' own rule is third-party: unpinned), no rebinning (a ``klim`` step other than the bin width raises).  ``s`` instead of ``k`` for correlation functions."""
import numpy as np


def is_measurement(obj):
    return hasattr(obj, 'get') and hasattr(obj, 'ells') and not isinstance(obj, (dict, np.ndarray))


def is_matrix_container(obj):
    return callable(getattr(obj, 'value', None)) and hasattr(obj, 'observable') and not isinstance(obj, np.ndarray)


def _pole(tree, ell, coord):
    pole = tree.get(ells=ell)
    x = np.asarray(pole.coords(coord), dtype='f8')
    edges = np.asarray(pole.edges(coord), dtype='f8').reshape(-1, 2) if hasattr(pole, 'edges') else None
    return pole, x, edges


def read_measurement(tree, lim=None, coord='k'):
    """-> (ells, [x per ell], [edges [n + 1] per ell], [values per ell], shotnoise or None), rows cut to ``lim`` = {ell: (lo, hi[, step])} (power_spectrum.py:165-179)."""
    ells = [int(ell) for ell in (lim.keys() if lim else tree.ells)]
    list_x, list_edges, list_value = [], [], []
    for ell in ells:
        pole, x, edges = _pole(tree, ell, coord)
        mask = np.ones(x.size, dtype='?')
        if lim:
            lo, hi, *step = lim[ell]
            if step and edges is not None and not np.isclose(step[0], np.diff(edges, axis=-1).mean(), rtol=1e-3):
                raise NotImplementedError('rebinning a container ({} step {} vs bin width {:.4g}): rebin the measurement first'.format(coord, step[0], np.diff(edges, axis=-1).mean()))
            mask = (x >= lo) & (x <= hi)
        list_x.append(x[mask])
        list_value.append(np.asarray(pole.value(), dtype='f8')[mask])
        list_edges.append(None if edges is None else np.append(edges[mask][:, 0], edges[mask][-1, 1]))
    shotnoise = None
    if 0 in [int(ell) for ell in tree.ells]:
        pole = tree.get(ells=0)
        if hasattr(pole, 'values'):
            try: shotnoise = float(np.mean(pole.values('shotnoise')))
            except (KeyError, ValueError, AttributeError): shotnoise = None
    return tuple(ells), list_x, list_edges, list_value, shotnoise


def _rows_of(tree, ells, list_x, coord='k'):
    """Indices, in the flat vector of ``tree`` (all its multipoles, in its own order), of the bins whose centres are ``list_x`` for the multipoles ``ells``."""
    offsets, start = {}, 0
    for ell in tree.ells:
        x = np.asarray(tree.get(ells=ell).coords(coord), dtype='f8')
        offsets[int(ell)] = (start, x)
        start += x.size
    index = []
    for ell, xx in zip(ells, list_x):
        if int(ell) not in offsets: raise ValueError('ell = {:d} not found in the container (ells = {})'.format(int(ell), list(offsets)))
        off, x = offsets[int(ell)]
        nearest = np.abs(x[None, :] - np.asarray(xx)[:, None]).argmin(axis=1)
        if not np.allclose(x[nearest], xx, rtol=1e-4, atol=0.):
            raise ValueError('{}-coordinates {} for ell = {:d} could not be found in the container ({})'.format(coord, xx, int(ell), x))
        index.append(off + nearest)
    return np.concatenate(index)


def read_window(wmatrix, ells, list_x, ellsin=None, coord='k', kin=None):
    """-> (matrix [n_out, n_ellin * n_kin], kin, ellsin): rows matched to the output bins (``list_x`` per multipole of ``ells``), columns of the input multipoles
    ``ellsin`` (default: all of ``wmatrix.theory``).  Without ``kin`` the input multipoles must share one wavenumber grid, which is returned (window.py:350-351); with
    ``kin`` every multipole is rebinned from ITS OWN grid, ``matrix . blockdiag(matrix_lininterp(kin, k_pole))^T`` (window.py:347-349)."""
    value = np.asarray(wmatrix.value(), dtype='f8')
    theory, observable = wmatrix.theory, wmatrix.observable
    ellsin = [int(ell) for ell in (ellsin if ellsin is not None else theory.ells)]
    rows = _rows_of(observable, ells, list_x, coord=coord)
    grids = [np.asarray(theory.get(ells=ell).coords('k'), dtype='f8') for ell in ellsin]
    if kin is None:
        if not all(grid.shape == grids[0].shape and np.allclose(grid, grids[0]) for grid in grids):
            raise ValueError('input coordinates of "wmatrix" are not the same for all multipoles; pass a k-coordinate array to "kin"')
        cols = _rows_of(theory, ellsin, grids, coord='k')
        return value[np.ix_(rows, cols)], grids[0], tuple(ellsin)
    from ... import utils
    kin = np.ravel(np.asarray(kin, dtype='f8'))
    cols = _rows_of(theory, ellsin, grids, coord='k')
    matrix = value[np.ix_(rows, cols)]
    blocks, start = [], 0
    for grid in grids:
        blocks.append(matrix[:, start:start + grid.size].dot(utils.matrix_lininterp(kin, grid).T))
        start += grid.size
    return np.hstack(blocks), kin, tuple(ellsin)


def read_covariance(covariance, observables):
    """-> covariance [n, n] of the flat data vector of ``observables`` (initialised mirror observables; matched by ``observable.name``), likelihoods/base.py:594-603."""
    value = np.asarray(covariance.value(), dtype='f8')
    tree = covariance.observable
    names = list(getattr(tree, 'observables', [obs.name for obs in observables]))
    offsets, start = {}, 0
    for name in names:
        sub = tree.get(observables=name)
        coord = 's' if hasattr(next(obs for obs in observables if obs.name == name), 's') and not hasattr(next(obs for obs in observables if obs.name == name), 'k') else 'k'
        size = sum(np.asarray(sub.get(ells=ell).coords(coord)).size for ell in sub.ells)
        offsets[name] = (start, sub, coord)
        start += size
    index = []
    for obs in observables:
        if obs.name not in offsets: raise ValueError('observable {} not found in the covariance (observables = {})'.format(obs.name, names))
        off, sub, coord = offsets[obs.name]
        index.append(off + _rows_of(sub, obs.ells, getattr(obs, coord), coord=coord))
    index = np.concatenate(index)
    return value[np.ix_(index, index)]
