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


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_gpu_covariance_plan_vs_reference(tag):
    """dl_cov_* through ctypes on the theory spectra the reference consumed: every block kind (P x P, xi x P, xi x xi), every parameter point of the fixture."""
    import torch
    from desilike_amd._lib import CovariancePlan
    from desilike_amd.observables.galaxy_clustering.covariance import CovariancePlanBuilder
    g, observables, theories, resolution = load(tag)
    desc = [dict(obs, theory_k=theory['k'], theory_ells=theory['ells']) for obs, theory in zip(observables, theories)]
    arrays = CovariancePlanBuilder(desc, resolution=resolution).arrays()
    plan = CovariancePlan(arrays['n'], [len(t['ells']) for t in theories], [len(t['k']) for t in theories], [list(t['ells']).index(0) for t in theories], [obs['shotnoise'] for obs in observables],
                          arrays['cell_i'], arrays['cell_d'], arrays['pt_i'], arrays['pt_d'], arrays['gtab'], arrays['sym'], device=0)
    powers = [torch.as_tensor(theory['power'], dtype=torch.float64, device='cuda').contiguous() for theory in theories]
    got = plan.apply(powers).cpu().numpy()
    ref = g[tag + '_covariance']
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max()), np.abs(got - ref).max() / np.abs(ref).max()
    assert np.array_equal(got, np.swapaxes(got, 1, 2))                                   # symmetric, exactly
    # a batch: the same matrices whatever the position in the batch
    big = [power[:1].repeat(37, 1, 1).contiguous() for power in powers]
    batch = plan.apply(big).cpu().numpy()
    assert all(np.array_equal(batch[i], got[0]) for i in range(37))


@pytest.mark.gpu
def test_gpu_covariance_call_surface_vs_reference():
    """ObservablesCovarianceMatrix(observables, footprints, resolution)(**params) as in the reference (covariance.py:274-342), theory multipoles from the device kernels."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, KaiserTracerCorrelationFunctionMultipoles
    from desilike_amd.observables.galaxy_clustering import (TracerPowerSpectrumMultipolesObservable, TracerCorrelationFunctionMultipolesObservable, ObservablesCovarianceMatrix,
                                                            BoxFootprint)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'covariance.npz'))
    names = [str(name) for name in g['a_point_names']]
    # (a) P_ell, ell = (0, 2, 4)
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0.01, 0.2, 20), ells=(0, 2, 4), theory=theory, shotnoise=1e4)
    cov = ObservablesCovarianceMatrix(obs, footprints=BoxFootprint(volume=1e10, nbar=1e-4), resolution=3)
    for ipoint, point in enumerate(g['a_points']):
        got = cov(**dict(zip(names, point)))
        ref = g['a_covariance'][ipoint]
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max()), np.abs(got - ref).max() / np.abs(ref).max()
    theta = np.array([[dict(zip(names, point)).get(param.basename, param.value) for param in cov.varied_params] for point in g['a_points']])
    batch = cov.evaluate_batch(theta)
    assert np.allclose(batch, g['a_covariance'], rtol=1e-9, atol=1e-12 * np.abs(g['a_covariance']).max())
    # (c) P_ell and xi_ell together
    # (one template per theory: each then has the knots the reference gives it, full_shape.py:29 -- a shared template would get the union of both ranges)
    obs1 = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0.01, 0.2, 20), ells=(0, 2),
                                                   theory=KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5)), shotnoise=1e4)
    obs2 = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, sedges=np.linspace(20., 160., 15), ells=(0, 2),
                                                         theory=KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5)))
    cov = ObservablesCovarianceMatrix([obs1, obs2], footprints=BoxFootprint(volume=1e10, nbar=1e-4), resolution=3)
    got = cov(**dict(zip(names, g['c_points'][0])))
    ref = g['c_covariance'][0]
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max()), np.abs(got - ref).max() / np.abs(ref).max()


def test_footprints():
    """covariance.py:54-271: box and cut-sky footprints (volume, number of objects, shot noise, mean / effective redshift) with a tabulated distance."""
    from desilike_amd.observables.galaxy_clustering import BoxFootprint, CutskyFootprint
    box = BoxFootprint(volume=1e9, nbar=2e-4)
    assert np.isclose(box.size, 2e5) and np.isclose(box.shotnoise, 5e3)
    both = box & BoxFootprint(volume=5e8, size=1e5)
    assert np.isclose(both.volume, 5e8) and np.isclose(both.shotnoise, 1. / 4e-4)

    def distance(z): return 3000. * np.asarray(z)       # (any monotonic function: the reference calls cosmo.comoving_radial_distance)

    sky = CutskyFootprint(area=1000., zrange=(0.4, 0.6), nbar=500., cosmo=distance)
    volume = 1000. / (180. / np.pi)**2 / 3. * ((3000. * 0.6)**3 - (3000. * 0.4)**3)
    assert np.isclose(sky.volume, volume) and np.isclose(sky.size, 5e5) and np.isclose(sky.shotnoise, volume / 5e5) and np.isclose(sky.zavg, 0.5)
    z = np.linspace(0.4, 0.6, 5)
    tab = CutskyFootprint(area=1000., zrange=z, nbar=1e-4 * np.ones(5), cosmo=distance)
    assert np.isclose(tab.size, 1e-4 * volume) and 0.5 < tab.zavg < 0.6 and np.isclose(tab.zeff, tab.zavg)
