"""A second, independent pin of the xi_ell path (build container only):

    python tests/golden/make_xi_pin_fixture.py

The reference's ``get_corr`` (theories/galaxy_clustering/base.py:127-136) calls the third-party ``cosmoprimo.PowerToCorrelation`` for its Hankel step; in the fixtures of
make_golden.py that class is the build's own oracle (tests/golden/refstub/cosmoprimo/__init__.py:40-50).  HERE the stand-in is written in this file and is nothing but
``scipy.fft.fht`` called directly (SciPy's FFTLog: an independent implementation of Hamilton's algorithm; same conventions as cosmoprimo's documented call
``PowerToCorrelation(k, ell, q=0, lowring=True)``: zero padding to the next power of two of 2 N, low-ringing offset, xi = (-1)^(ell/2) (2 pi)^(-3/2) s^(-3/2) H[k^(3/2) P]):
no class or function of oracle/ is imported.  The reference then runs its OWN prologue (interpolation to the FFTLog grid, high-k tail), epilogue (interpolation to s),
window and likelihood on top.  Stored as well: the reference's brute-force integral of the same P_ell (theories/galaxy_clustering/base.py:163-168, the formula of its
``plot`` method, on the reference's own theory grid; its ``utils.weights_trapz`` -- ``jnp.insert`` at index len(x) - 1 of an array of len(x) - 2 -- does not run on this
image's NumPy 2: the trapezoid weights it means are written out) -- FFTLog and quadrature agree to the truncation of the
k range (the test states the level).  cosmoprimo itself is absent: "1e-10 against cosmoprimo" stays unclaimed."""
import os
import sys
import warnings

import numpy as np
from scipy import fft, special

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')

import cosmoprimo


class ScipyFHTPowerToCorrelation(object):
    """``scipy.fft.fht`` and nothing else."""

    def __init__(self, k, ell=0, q=0, lowring=True, **kwargs):
        assert q == 0 and lowring
        self.k = np.asarray(k, dtype='f8')
        self.ells = np.atleast_1d(ell)
        n = self.k.size
        self.dln = np.log(self.k[-1] / self.k[0]) / (n - 1)
        self.npad = 1 << int(np.ceil(np.log2(2 * n)))
        self.lo = (self.npad - n) // 2

    def __call__(self, fun):
        fun = np.atleast_2d(np.asarray(fun, dtype='f8'))
        n = self.k.size
        kpad = self.k[0] * np.exp(self.dln * (np.arange(self.npad) - self.lo))
        s, xi = [], []
        for pk, ell in zip(fun, self.ells):
            offset = fft.fhtoffset(self.dln, mu=ell + 0.5, initial=0., bias=0.)
            a = np.zeros(self.npad)
            a[self.lo:self.lo + n] = pk * self.k**1.5
            A = fft.fht(a, self.dln, mu=ell + 0.5, offset=offset, bias=0.)
            sout = np.exp(offset) / kpad[::-1]
            s.append(sout[self.lo:self.lo + n])
            xi.append((-1.)**(ell // 2) / (2. * np.pi)**1.5 * A[self.lo:self.lo + n] * sout[self.lo:self.lo + n]**(-1.5))
        return np.array(s), np.array(xi)


cosmoprimo.PowerToCorrelation = ScipyFHTPowerToCorrelation     # what ``from cosmoprimo import PowerToCorrelation`` (tgc/base.py:76) finds

from desilike.theories.galaxy_clustering import (ShapeFitPowerSpectrumTemplate, KaiserTracerCorrelationFunctionMultipoles, BAOPowerSpectrumTemplate,
                                                 DampedBAOWigglesTracerCorrelationFunctionMultipoles)
from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
from desilike.likelihoods import ObservablesGaussianLikelihood
from desilike.base import vmap

sys.path.insert(0, here)
from make_golden import sample_theta


def bruteforce(theory):
    """tgc/base.py:163-168, verbatim arithmetic on the reference's own arrays."""
    kin, ells = np.asarray(theory.kin), theory.ells
    power = np.asarray(theory.power.power)
    x = np.log(kin)
    weights = np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.      # utils.weights_trapz (utils.py:620-622)
    corr = []
    for ill, ell in enumerate(ells):
        tmp = np.sum(kin**3 * power[ill] * weights * special.spherical_jn(ell, np.asarray(theory.s)[:, None] * kin), axis=-1)
        corr.append((-1) ** (ell // 2) / (2. * np.pi**2) * tmp)
    return np.array(corr)


def dump(name, like, theory, size, seed, brute=True):
    like()
    assert isinstance(theory.fftlog, ScipyFHTPowerToCorrelation)
    names = like.varied_params.names()
    theta = sample_theta(like, size, seed)
    (logpost, derived), errors = vmap(like, backend=None, errors='return', return_derived=True)({n: theta[:, i] for i, n in enumerate(names)})
    assert not errors
    corr, bf, power = [], [], []
    for row in theta:
        like(**dict(zip(names, row)))
        corr.append(np.asarray(theory.corr).copy()); power.append(np.asarray(theory.power.power).copy())
        if brute: bf.append(bruteforce(theory))
    fn = os.path.join(here, 'xi_pin_{}.npz'.format(name))
    np.savez_compressed(fn, names=np.array(names), theta=theta, corr=np.array(corr), bruteforce=np.array(bf), power=np.array(power), s=np.asarray(theory.s), kin=np.asarray(theory.kin),
                        ells=np.array(theory.ells), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
                        flatdata=np.asarray(like.flatdata))
    print('saved', fn, '{:.1f} kB'.format(os.path.getsize(fn) / 1e3), names)


def main():
    s = np.linspace(22.5, 167.5, 30)
    # Kaiser xi_ell on a ShapeFit template: the pipeline of fixture kaiser_xi (make_golden.py), same covariance, the transform swapped
    theory = KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, s=s, ells=(0, 2, 4), theory=theory)
    rng = np.random.RandomState(14)
    A = rng.standard_normal((90, 90)) * 3e-4
    dump('kaiser', ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (3e-3)**2 * np.eye(90)), theory, 24, 19)
    # BASELINE configs[3]: damped BAO xi_ell (the pipeline of fixture cfg4_bao_xi; no brute-force leg: the broadband terms are added in configuration space)
    theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=BAOPowerSpectrumTemplate(z=0.5), mode='reciso')
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=s, ells=(0, 2), theory=theory)
    for pname in ['sigmapar', 'sigmaper']:
        theory.init.params[pname].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
    rng = np.random.RandomState(4)
    A = rng.standard_normal((60, 60)) * 3e-4
    dump('bao', ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (3e-3)**2 * np.eye(60)), theory, 24, 9, brute=False)


if __name__ == '__main__':
    main()
