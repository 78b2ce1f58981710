"""Golden vectors of the reference's tracer-velocity variant of the PNG theory (primordial_non_gaussianity.py:196-330: odd multipoles of the density-velocity cross
spectrum, 81 trapezoid nodes in mu on [-1, 1]), run with the reference's own code:

    python tests/golden/make_png_velocity_fixture.py        (build container only; writes tests/golden/png_velocity.npz)

Harness shim (as tests/golden/make_tns_fixture.py): ``utils.weights_trapz`` calls ``jnp.insert`` with an index one past the end (jax clamps it, numpy raises)."""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402

from desilike import utils   # noqa: E402


def _weights_trapz(x):
    x = np.asarray(x, dtype='f8')
    return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.


utils.weights_trapz = _weights_trapz

from desilike.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate   # noqa: E402
from desilike.theories.galaxy_clustering.primordial_non_gaussianity import PNGTracerVelocityPowerSpectrumMultipoles   # noqa: E402

def boundary():
    """The reference-side binding's key set for a likelihood on the odd multipoles: tests/golden/boundary_png_velocity.npz."""
    import make_boundary_fixture as mb
    from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike.likelihoods import ObservablesGaussianLikelihood
    theory = PNGTracerVelocityPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5), mode='b-p')
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'bv': 1.1, 'fnl_loc': 20., 'sigmau': 3.}, kedges=np.linspace(0.005, 0.105, 21), ells=(1, 3), wmatrix={'resolution': 2}, theory=theory)
    likelihood = ObservablesGaussianLikelihood(observables=[obs], covariance=mg.spd_covariance(40, seed=8, diag=1e10, amp=2e4))
    likelihood()
    mb.dump('png_velocity', likelihood, size=16, seed=10)


if __name__ == '__main__':
    if '--boundary' in sys.argv:
        boundary()
        sys.exit(0)
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = PNGTracerVelocityPowerSpectrumMultipoles(template=template, k=np.linspace(0.005, 0.15, 30), mode='b-p')
    theory()
    names = theory.varied_params.names()
    rng = np.random.RandomState(9)
    lo = dict(qpar=0.95, qper=0.95, dm=-0.05, df=0.9, b1=1.5, bv=0.7, sigmas=0., sigmau=0., fnl_loc=-50., p=0.8)
    hi = dict(qpar=1.05, qper=1.05, dm=0.05, df=1.1, b1=2.5, bv=1.3, sigmas=5., sigmau=8., fnl_loc=50., p=1.6)
    theta = np.column_stack([rng.uniform(lo[name], hi[name], 12) for name in names])
    power = []
    for row in theta:
        theory(**dict(zip(names, row)))
        power.append(np.asarray(theory.power).copy())
    kin = np.asarray(template.k)
    cosmo = template.cosmo
    pphi_prim = 9 / 25 * 2 * np.pi**2 / kin**3 * cosmo.get_primordial(mode='scalar').pk_interpolator()(kin) / cosmo.h**3
    np.savez(os.path.join(here, 'png_velocity.npz'), names=np.array(names), theta=theta, power=np.array(power), k=np.asarray(theory.k), ells=np.asarray(theory.ells), mu=np.asarray(theory.mu),
             wmu_ell=np.asarray(theory.wmu), k11=kin, pk_dd_fid=np.asarray(template.pk_dd_fid), alpha_fid=1. / (np.asarray(template.pk_dd_fid) / pphi_prim)**0.5,
             f_fid=float(template.f_fid), z=float(template.z), kp=template.kp, a=template.a)
    print(names, theory.ells, np.array(power).shape, len(theory.mu), np.abs(power[0]).max())
