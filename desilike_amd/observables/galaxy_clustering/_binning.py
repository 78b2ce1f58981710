"""Output binning of the multipole observables: one resolver shared by P_ell(k) and xi_ell(s).

Behaviour reproduced (desilike/observables/galaxy_clustering/window.py:214-292 for k, 583-640 for s), stated as rules rather than as the reference's control flow:

* multipoles: ``ells`` if given, else the keys of ``lim``, else (0, 2, 4);
* coordinates ``x`` and bin ``edges`` may be one array (shared by all multipoles) or one per multipole; 1-D edges are consecutive bin boundaries;
* ``lim = {ell: (lo, hi[, step]) | None}`` selects multipoles (a multipole absent from ``lim`` is dropped) and coordinates (``lo <= x <= hi``); a multipole left
  without coordinates is dropped; without ``x`` the multipoles must be exactly the keys of ``lim``;
* missing edges come, in this order, from ``lim`` (regular bins of width ``step``, else ``(hi - lo) / len(x)``, else the default width -- only if every
  kept multipole has a range), from ``x`` (mid-points, end bins mirrored), or from the default grid;
* missing coordinates are the bin centres;
* xi_ell only (``lim_from_edges``): given bins without ``lim`` also cut given coordinates to the range the bins span.
"""
import numpy as np


def _per_multipole(value, n):
    """One array for all multipoles, or a sequence with one array per multipole -> list of n float arrays."""
    if np.ndim(value[0]) == 0:
        value = [value] * n
    return [np.array(v, dtype='f8') for v in value]


def _as_pairs(edges):
    """Bin boundaries [n + 1] -> rows (low, high) [n, 2]; arrays already in that shape pass through."""
    edges = np.asarray(edges, dtype='f8')
    return np.stack([edges[:-1], edges[1:]], axis=-1) if edges.ndim == 1 else edges


def _regular_pairs(lo, hi, step):
    return _as_pairs(np.arange(lo, hi + 0.5 * step, step))


def _pairs_around(x):
    """Bins whose boundaries are the mid-points of consecutive coordinates; the first / last bin is as wide on its outer side as on its inner side."""
    mid = 0.5 * (x[1:] + x[:-1])
    return _as_pairs(np.concatenate([[mid[0] - (x[1] - x[0])], mid, [mid[-1] + (x[-1] - x[-2])]]))


class MultipoleBins(object):
    """Resolved binning: ``ells`` (tuple), ``x`` / ``edges`` (one array per multipole), ``masklim`` ({ell: bool mask over the input x} when ``lim`` and ``x``
    were both given, else None)."""

    def __init__(self, ells, x, edges, masklim=None):
        self.ells, self.x, self.edges, self.masklim = tuple(ells), x, edges, masklim

    @property
    def size(self):
        return sum(len(xx) for xx in self.x)

    @classmethod
    def resolve(cls, x=None, edges=None, lim=None, ells=None, default_step=0.01, default_edges=None, label='k', lim_from_edges=False):
        if ells is None:
            ells = tuple(lim) if lim is not None else (0, 2, 4)
        ells = tuple(ells)
        n = len(ells)
        if x is not None:
            x = _per_multipole(x, n)
            if len(x) != n: raise ValueError("provide as many {}'s as ells".format(label))
        if edges is not None:
            edges = [_as_pairs(e) for e in _per_multipole(edges, n)]
            if len(edges) != n: raise ValueError('provide as many {}edges as ells'.format(label))
        if lim is None and edges is not None and lim_from_edges:
            # xi_ell only (window.py:593-594): given bins also act as a range cut on given coordinates
            lim = {ell: (e[0, 0], e[-1, 1], np.mean(e[..., 1] - e[..., 0])) for ell, e in zip(ells, edges)}
        masklim = None
        if lim is not None:
            lim = dict(lim)
            if x is not None:
                masklim, kept = {}, []
                for ell, xx in zip(ells, x):
                    inside = np.full(xx.shape, ell in lim)
                    if inside.any() and lim[ell] is not None:
                        lo, hi = lim[ell][:2]
                        inside = (xx >= lo) & (xx <= hi)
                    masklim[ell] = inside
                    if inside.any(): kept.append((ell, xx[inside]))
                ells, x = tuple(ell for ell, _ in kept), [xx for _, xx in kept]
            elif list(ells) != list(lim):
                raise ValueError('incompatible ells = {} and {}lim = {}; just remove ells?'.format(ells, label, list(lim)))
            if edges is None and all(lim[ell] is not None for ell in ells):
                edges = []
                for ill, ell in enumerate(ells):
                    lo, hi, *step = lim[ell]
                    step = step[0] if step else ((hi - lo) / x[ill].size if x is not None else default_step)
                    edges.append(_regular_pairs(lo, hi, step))
        if edges is None:
            edges = [_pairs_around(xx) for xx in x] if x is not None else [_as_pairs(default_edges)] * len(ells)
        if x is None:
            x = [e.mean(axis=-1) for e in edges]
        return cls(ells, [np.array(xx) for xx in x], edges, masklim)

    def input_grid(self):
        """Without a window matrix the theory is evaluated on the union of the output coordinates: returns (xin, mask) with ``mask`` the row selection
        into the [n_ell, len(xin)] theory grid, or None when every multipole lives on the full union (window.py:294-305)."""
        xin = np.unique(np.concatenate(self.x, axis=0))
        if all(xx.shape == xin.shape and np.allclose(xx, xin) for xx in self.x):
            return xin, None
        index = [np.searchsorted(xin, xx, side='left') for xx in self.x]
        if not all(np.allclose(xin[idx], xx) for idx, xx in zip(index, self.x)):
            raise ValueError('output coordinates do not lie on their union grid')
        return xin, np.concatenate([xin.size * ill + idx for ill, idx in enumerate(index)], axis=0)
