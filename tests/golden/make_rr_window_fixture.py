"""Golden matrix of the reference's RR-count window of correlation function multipoles (window.py:71-138: Legendre mixing from the mu-distribution of the random-random
pair counts in every output separation bin, times the bin integration), from the reference's own function on synthetic counts:

    python tests/golden/make_rr_window_fixture.py        (build container only; writes tests/golden/rr_window.npz)
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402, F401

from desilike.observables.galaxy_clustering.window import window_matrix_RR   # noqa: E402


def pairs(edges):
    return np.column_stack([edges[:-1], edges[1:]])


if __name__ == '__main__':
    rng = np.random.RandomState(31)
    sedges, muedges = np.linspace(0., 200., 101), np.linspace(-1., 1., 41)
    smid, mumid = (sedges[:-1] + sedges[1:]) / 2., (muedges[:-1] + muedges[1:]) / 2.
    wcounts = smid[:, None]**2 * (1. + 0.3 * mumid[None, :]**2) * (1. + 0.05 * rng.standard_normal((smid.size, mumid.size)))
    wcounts[(smid[:, None] < 60.) & (np.abs(mumid[None, :]) > 0.9)] = 0.          # a survey edge: no pairs along the line of sight at small separations
    sout = {0: np.linspace(20., 160., 15), 2: np.linspace(30., 150., 13)}
    sin, matrix = window_matrix_RR({ell: pairs(edges) for ell, edges in sout.items()}, pairs(sedges), pairs(muedges), wcounts, ellsin=(0, 2, 4), resolution=2)
    np.savez(os.path.join(here, 'rr_window.npz'), sedges=sedges, muedges=muedges, wcounts=wcounts, sout0=sout[0], sout2=sout[2], sin=sin, matrix=matrix)
    print(sin.shape, matrix.shape, np.abs(matrix).max())
