"""Init-time numerical helpers of the hot path (host side, run once per likelihood).

Each function states the reference routine whose behaviour it reproduces (paths relative to /root/reference/desilike).
"""
import numpy as np


def weights_leggauss(nx, sym=False):
    """Gauss-Legendre nodes / weights; ``sym=True``: positive half of ``2 nx`` nodes with symmetrised weights (utils.py:625-630)."""
    x, wx = np.polynomial.legendre.leggauss((1 + sym) * nx)
    if sym:
        x, wx = x[nx:], (wx[nx:] + wx[nx - 1::-1]) / 2.
    return x, wx


def weights_mu(mu, method='leggauss'):
    """mu-integration weights on [0, 1] (utils.py:633-643); only Gauss-Legendre is implemented on the device path."""
    if method != 'leggauss':
        raise NotImplementedError('only method="leggauss" is supported')
    if np.ndim(mu) != 0:
        raise ValueError('gauss integration does not take an array of mus')
    return weights_leggauss(int(mu), sym=True)


def legendre(ell, x):
    """Legendre polynomial L_ell(x) for the even multipoles used by the path (closed forms up to ell = 8)."""
    from numpy.polynomial import legendre as npleg
    coeffs = np.zeros(ell + 1)
    coeffs[ell] = 1.
    return npleg.legval(x, coeffs)


def multipole_weights(mu, wmu, ells):
    """``w_m (2 ell + 1) L_ell(mu_m)`` (theories/galaxy_clustering/base.py:201-204)."""
    return np.array([wmu * (2 * ell + 1) * legendre(ell, mu) for ell in ells])


def matrix_lininterp(xin, xout):
    """Linear-interpolation operator from the sorted grid ``xin`` to the points ``xout``: ``f(xout) = M^T f(xin)``, ``M`` of shape (len(xin), len(xout)).
    Points outside ``[xin[0], xin[-1]]`` get a zero column, except a point at (to rounding) the last node, which takes that node's value
    (behaviour of the reference's utils.py:646-657)."""
    xin, xout = np.asarray(xin, dtype='f8'), np.atleast_1d(np.asarray(xout, dtype='f8'))
    nin, columns = xin.size, np.arange(xout.size)
    matrix = np.zeros((nin, xout.size), dtype='f8')
    left = np.searchsorted(xin, xout, side='right') - 1          # node at or below each point
    inside = (left >= 0) & (left < nin - 1)
    lo = left[inside]
    t = (xout[inside] - xin[lo]) / (xin[lo + 1] - xin[lo])
    matrix[lo, columns[inside]] = 1. - t
    matrix[lo + 1, columns[inside]] = t
    at_end = (left == nin - 1) & np.isclose(xout, xin[-1])
    matrix[nin - 1, columns[at_end]] = 1.
    return matrix


def _shell_mean(lo, hi):
    """Volume-weighted mean radius of the spherical shell [lo, hi]: int x x^2 dx / int x^2 dx."""
    return 0.75 * (hi**4 - lo**4) / (hi**3 - lo**3)


def window_matrix_bininteg(list_edges, resolution=1):
    """Binning matrix in the continuous limit (behaviour of observables/galaxy_clustering/window.py:14-68): the average of the theory over a bin, weighted by the
    volume x^2 dx of spherical shells, is approximated by splitting every bin into ``resolution`` shells, each contributing the theory linearly interpolated
    at its volume-weighted mean radius from one common input grid (shells of width min(bin width) / resolution spanning all bins).

    ``list_edges``: per multipole an array [n_bins, 2] of (low, high) (or one such array).  Returns the input grid ``xin`` and the block-diagonal matrix
    [n_multipoles * len(xin), total number of bins]."""
    resolution = int(resolution)
    if resolution <= 0:
        raise ValueError('resolution must be a strictly positive integer')
    if np.ndim(list_edges[0]) == 0:
        list_edges = [list_edges]
    list_edges = [np.asarray(edges, dtype='f8') for edges in list_edges]
    width = min(np.min(edges[..., 1] - edges[..., 0]) for edges in list_edges) / resolution
    first, last = min(edges.min() for edges in list_edges), max(edges.max() for edges in list_edges)
    grid = np.arange(first, last + 0.5 * width, width)
    xin = _shell_mean(grid[:-1], grid[1:])
    blocks = []
    for edges in list_edges:
        shells = np.linspace(edges[:, 0], edges[:, 1], resolution + 1, axis=-1)            # [n_bins, resolution + 1]
        lo, hi = shells[:, :-1], shells[:, 1:]
        volume = hi**3 - lo**3
        weight = volume / volume.sum(axis=-1, keepdims=True)                               # shells of a bin, normalised
        interp = matrix_lininterp(xin, _shell_mean(lo, hi).ravel())                         # [len(xin), n_bins * resolution]
        blocks.append((interp * weight.ravel()).reshape(xin.size, len(edges), resolution).sum(axis=-1))
    nrows, ncols = xin.size * len(blocks), sum(block.shape[1] for block in blocks)
    full = np.zeros((nrows, ncols), dtype='f8')
    col = 0
    for ill, block in enumerate(blocks):
        full[ill * xin.size:(ill + 1) * xin.size, col:col + block.shape[1]] = block
        col += block.shape[1]
    return xin, full


def inv(mat, check_valid='raise'):
    """Matrix inverse, checked: ``mat . inverse`` must be the identity to 1e-3 (what the reference demands of an inverse covariance, utils.py:495-558);
    ``check_valid``: 'raise', 'warn' or 'ignore'."""
    mat = np.asarray(mat, dtype='f8')
    if mat.ndim == 0:
        return 1. / mat
    inverse = np.linalg.inv(mat)
    if check_valid == 'ignore':
        return inverse
    error = np.abs(mat.dot(inverse) - np.eye(mat.shape[0]))
    if not (error <= 1e-3 + 1e-3 * np.eye(mat.shape[0])).all():
        msg = 'Numerically inaccurate inverse matrix, max absolute diff {:.6f}.'.format(error.max())
        if check_valid == 'raise':
            raise np.linalg.LinAlgError(msg)
        import warnings
        warnings.warn(msg)
    return inverse


def blockinv(blocks, check_valid='raise'):
    """Inverse of a matrix given as a square list of lists of blocks, by block elimination -- the leading block against the (recursively inverted) remainder
    through its Schur complement, each dense inverse checked by :func:`inv`.  The reference inverts joint covariances this way (utils.py:561-599,
    likelihoods/base.py:617-619): numerically this is not the same as inverting the assembled matrix, so the order of elimination is kept."""
    nblocks = len(blocks)
    head = np.asarray(blocks[0][0], dtype='f8')
    if nblocks == 1:
        return inv(head, check_valid=check_valid)
    right = np.concatenate([np.atleast_2d(b) for b in blocks[0][1:]], axis=1)              # head row, remaining columns
    below = np.concatenate([np.atleast_2d(row[0]) for row in blocks[1:]], axis=0)          # head column, remaining rows
    rest_inv = blockinv([row[1:] for row in blocks[1:]], check_valid=check_valid)
    right_rest = right.dot(rest_inv)
    schur_inv = inv(head - right_rest.dot(below), check_valid=check_valid)
    rest_below = rest_inv.dot(below)
    n0, n = head.shape[0], head.shape[0] + rest_inv.shape[0]
    out = np.empty((n, n), dtype='f8')
    out[:n0, :n0] = schur_inv
    out[:n0, n0:] = -schur_inv.dot(right).dot(rest_inv)
    out[n0:, :n0] = -rest_below.dot(schur_inv)
    out[n0:, n0:] = rest_inv + rest_below.dot(schur_inv).dot(right).dot(rest_inv)
    return out


def weights_trapz(x):
    """Trapezoidal integration weights on the (non-uniform) grid ``x`` (reference: utils.py:614-622)."""
    x = np.asarray(x, dtype='f8')
    if x.size <= 1:
        return np.ones(max(x.size, 1), dtype='f8')[:x.size] if x.size else np.array(1.)
    w = np.empty_like(x)
    w[0], w[-1] = x[1] - x[0], x[-1] - x[-2]
    w[1:-1] = x[2:] - x[:-2]
    return w / 2.
