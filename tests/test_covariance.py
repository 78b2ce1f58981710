"""Gaussian covariance of power-spectrum / correlation-function multipoles (reference: observables/galaxy_clustering/covariance.py:274-456)
against the reference's own matrices (tests/golden/covariance.npz <- tests/golden/make_covariance_fixture.py): the oracle on CPU, the device kernel on the GPU."""
import os

import numpy as np
import pytest

from oracle import np_oracle as orc


def load(tag):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'covariance.npz'))
    observables, theories = [], []
    io = 0
    while '{}_obs{:d}_kind'.format(tag, io) in g.files:
        ells = [int(ell) for ell in g['{}_obs{:d}_ells'.format(tag, io)]]
        volume, shotnoise = g['{}_obs{:d}_footprint'.format(tag, io)]
        observables.append(dict(kind=str(g['{}_obs{:d}_kind'.format(tag, io)]), ells=ells, edges=[g['{}_obs{:d}_edges{:d}'.format(tag, io, ill)] for ill in range(len(ells))],
                                volume=float(volume), shotnoise=float(shotnoise)))
        theories.append(dict(k=g['{}_theory{:d}_k'.format(tag, io)], ells=[int(ell) for ell in g['{}_theory{:d}_ells'.format(tag, io)]], power=g['{}_theory{:d}_power'.format(tag, io)]))
        io += 1
    return g, observables, theories, int(g[tag + '_resolution'])


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_oracle_covariance_vs_reference(tag):
    g, observables, theories, resolution = load(tag)
    ref = g[tag + '_covariance']
    for ipoint in range(ref.shape[0]):
        got = orc.gaussian_covariance(observables, [dict(theory, power=theory['power'][ipoint]) for theory in theories], resolution=resolution)
        assert got.shape == ref[ipoint].shape
        assert np.allclose(got, ref[ipoint], rtol=1e-12, atol=1e-14 * np.abs(ref[ipoint]).max()), np.abs(got - ref[ipoint]).max()
