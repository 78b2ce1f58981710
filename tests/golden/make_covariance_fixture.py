"""Gaussian covariance matrices computed by the REFERENCE (desilike/observables/galaxy_clustering/covariance.py:274-456, ``ObservablesCovarianceMatrix``) for
power-spectrum and correlation-function multipoles, with the theory spectra it consumed.  Build container only:

    python tests/golden/make_covariance_fixture.py

The reference's ``utils.weights_trapz`` relies on ``jnp.insert`` clamping an out-of-range index (jax is absent here and numpy raises): replaced by the same weights
written for numpy -- the one reference line replaced, as in make_golden.cfg2_fc_syst."""
import os
import sys
import warnings

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')

import desilike.utils as ref_utils


def weights_trapz(x):
    x = np.asarray(x)
    return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.


ref_utils.weights_trapz = weights_trapz

from desilike.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, KaiserTracerCorrelationFunctionMultipoles
from desilike.observables.galaxy_clustering import (TracerPowerSpectrumMultipolesObservable, TracerCorrelationFunctionMultipolesObservable, ObservablesCovarianceMatrix,
                                                    BoxFootprint)


def capture(tag, cov, points, out):
    values, powers = [], None
    for params in points:
        c = cov(**params)
        values.append(np.asarray(c.value))
        theories = list(cov.theories)
        if powers is None: powers = [[] for _ in theories]
        for it, theory in enumerate(theories): powers[it].append(np.asarray(theory.power).copy())
    out[tag + '_covariance'] = np.array(values)
    for it, theory in enumerate(cov.theories):
        out['{}_theory{:d}_k'.format(tag, it)] = np.asarray(theory.k)
        out['{}_theory{:d}_ells'.format(tag, it)] = np.array(theory.ells)
        out['{}_theory{:d}_power'.format(tag, it)] = np.array(powers[it])
    for io, (obs, fp) in enumerate(zip(cov.observables, cov.footprints)):
        edges = obs.kedges if hasattr(obs, 'kedges') else obs.sedges
        out['{}_obs{:d}_kind'.format(tag, io)] = 'pk' if hasattr(obs, 'kedges') else 'xi'
        out['{}_obs{:d}_ells'.format(tag, io)] = np.array(obs.ells)
        for ill in range(len(obs.ells)):
            out['{}_obs{:d}_edges{:d}'.format(tag, io, ill)] = np.asarray(edges[ill])
        out['{}_obs{:d}_footprint'.format(tag, io)] = np.array([float(fp.volume), float(fp.shotnoise)])
    out[tag + '_resolution'] = np.array(cov.resolution)
    out[tag + '_points'] = np.array([[params[name] for name in sorted(points[0])] for params in points])
    out[tag + '_point_names'] = np.array(sorted(points[0]))


def main():
    out = {}
    points = [dict(b1=2., qpar=1., df=1.), dict(b1=1.5, qpar=1.02, df=0.9)]
    # (a) P_ell, ell = (0, 2, 4), theory evaluated at the bin centres, three integration points per bin
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0.01, 0.2, 20), ells=(0, 2, 4), theory=theory, shotnoise=1e4)
    capture('a', ObservablesCovarianceMatrix(obs, footprints=BoxFootprint(volume=1e10, nbar=1e-4), resolution=3), points, out)
    # (b) P_ell with a binning window (theory on the fine grid), different k-ranges per multipole, two integration points
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    kedges = np.linspace(0., 0.2, 41)
    kc = (kedges[:-1] + kedges[1:]) / 2.
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, klim={0: (0.02, 0.2, 0.005), 2: (0.02, 0.15, 0.005)}, ells=(0, 2), wmatrix={'resolution': 4}, theory=theory, shotnoise=5e3)
    capture('b', ObservablesCovarianceMatrix(obs, footprints=BoxFootprint(volume=3e9, nbar=2e-4), resolution=2), points, out)
    # (c) P_ell and xi_ell of the same tracer: auto blocks and the cross block
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    obs1 = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0.01, 0.2, 20), ells=(0, 2), theory=KaiserTracerPowerSpectrumMultipoles(template=template), shotnoise=1e4)
    obs2 = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, sedges=np.linspace(20., 160., 15), ells=(0, 2), theory=KaiserTracerCorrelationFunctionMultipoles(template=template))
    capture('c', ObservablesCovarianceMatrix([obs1, obs2], footprints=BoxFootprint(volume=1e10, nbar=1e-4), resolution=3), points[:1], out)
    fn = os.path.join(here, 'covariance.npz')
    np.savez_compressed(fn, **out)
    print('saved', fn, '{:.1f} kB'.format(os.path.getsize(fn) / 1e3), {k: np.shape(v) for k, v in out.items() if 'covariance' in k or 'power' in k or '_k' in k})


if __name__ == '__main__':
    main()
