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
    """Matrix for linear interpolation from ``xin`` to ``xout`` (utils.py:646-657), shape (len(xin), len(xout))."""
    xin = np.asarray(xin)
    toret = np.zeros((len(xin), len(xout)), dtype='f8')
    for iout, xo in enumerate(xout):
        iin = np.searchsorted(xin, xo, side='right') - 1
        if 0 <= iin < len(xin) - 1:
            frac = (xo - xin[iin]) / (xin[iin + 1] - xin[iin])
            toret[iin, iout] = 1. - frac
            toret[iin + 1, iout] = frac
        elif np.isclose(xo, xin[-1]):
            toret[iin, iout] = 1.
    return toret


def window_matrix_bininteg(list_edges, resolution=1):
    """Binning window matrix in the continuous limit (observables/galaxy_clustering/window.py:14-68).

    Returns ``xin`` and the matrix of shape (n_in_total, n_out_total).
    """
    resolution = int(resolution)
    if resolution <= 0:
        raise ValueError('resolution must be a strictly positive integer')
    if np.ndim(list_edges[0]) == 0:
        list_edges = [list_edges]
    list_edges = [np.asarray(edges, dtype='f8') for edges in list_edges]
    step = min((edges[..., 1] - edges[..., 0]).min() for edges in list_edges) / resolution
    start, stop = min(np.min(edges) for edges in list_edges), max(np.max(edges) for edges in list_edges)
    edgesin = np.arange(start, stop + step / 2., step)
    xin = 3. / 4. * (edgesin[1:]**4 - edgesin[:-1]**4) / (edgesin[1:]**3 - edgesin[:-1]**3)
    matrices = []
    for edges in list_edges:
        x, w = [], []
        for ibin, edge in enumerate(edges):
            edge = np.linspace(*edge, resolution + 1)
            x.append(3. / 4. * (edge[1:]**4 - edge[:-1]**4) / (edge[1:]**3 - edge[:-1]**3))
            line = np.zeros(len(edges) * resolution, dtype='f8')
            tmp = edge[1:]**3 - edge[:-1]**3
            line[ibin * resolution:(ibin + 1) * resolution] = tmp / tmp.sum()
            w.append(line)
        matrices.append(matrix_lininterp(xin, np.concatenate(x)).dot(np.column_stack(w)))
    n = len(matrices)
    full = np.block([[matrices[i] if i == j else np.zeros((matrices[i].shape[0], matrices[j].shape[1])) for j in range(n)] for i in range(n)])
    return xin, full


def inv(mat, check_valid='raise'):
    """Matrix inverse with the reference's 1e-3 validity check (utils.py:495-558)."""
    mat = np.asarray(mat, dtype='f8')
    if mat.ndim == 0:
        return 1. / mat
    toret = np.linalg.inv(mat)
    if check_valid != 'ignore':
        tmp = mat.dot(toret)
        if not np.allclose(tmp, np.eye(tmp.shape[0]), rtol=1e-3, atol=1e-3):
            msg = 'Numerically inaccurate inverse matrix, max absolute diff {:.6f}.'.format(np.max(np.abs(tmp - np.eye(tmp.shape[0]))))
            if check_valid == 'raise':
                raise np.linalg.LinAlgError(msg)
            import warnings
            warnings.warn(msg)
    return toret


def blockinv(blocks, check_valid='raise'):
    """Block-wise (Schur complement) inverse, recursing over the first block like utils.py:561-599."""
    A = np.asarray(blocks[0][0])
    if (len(blocks), len(blocks[0])) == (1, 1):
        return inv(A, check_valid=check_valid)
    B = np.block([list(blocks[0][1:])])
    C = np.block([[b[0]] for b in blocks[1:]])
    invD = blockinv([b[1:] for b in blocks[1:]], check_valid=check_valid)
    invShur = inv(A - B.dot(invD).dot(C), check_valid=check_valid)
    toret = np.block([[invShur, -invShur.dot(B).dot(invD)], [-invD.dot(C).dot(invShur), invD + invD.dot(C).dot(invShur).dot(B).dot(invD)]])
    return toret


def weights_trapz(x):
    """Trapezoidal integration weights on the (non-uniform) grid ``x`` (reference: utils.py:614-622)."""
    x = np.asarray(x, dtype='f8')
    if x.size <= 1:
        return np.ones(max(x.size, 1), dtype='f8')[:x.size] if x.size else np.array(1.)
    w = np.empty_like(x)
    w[0], w[-1] = x[1] - x[0], x[-1] - x[-2]
    w[1:-1] = x[2:] - x[:-2]
    return w / 2.
