"""Prove the drop-in boundary from the REFERENCE side (build container only; the reference never travels to the GPU box):

    python tests/golden/make_boundary_fixture.py

Builds real desilike likelihoods with the reference's own classes (through tests/golden/refstub for the absent cosmoprimo / lsstypes), initialises them, runs
``integration/desilike_mi355x.py::extract_config`` -- the reference-side binding of INTEGRATION.md section 2 -- on them, and stores in ``boundary_<name>.npz``:
the flat ``dl_config`` key -> array set (``cfg/<key>``), a theta batch, and the reference's own ``vmap(likelihood, return_derived=True)`` outputs.
tests/test_gpu_boundary.py creates the device context from these keys alone (ctypes, no desilike_amd host mirror) and must reproduce the reference's numbers.
"""
import os
import sys
import warnings

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'integration'))
warnings.filterwarnings('ignore')

from desilike.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
from desilike.likelihoods import ObservablesGaussianLikelihood
from desilike.base import vmap

from desilike_mi355x import extract_config
from make_golden import dense_window, sample_theta


def covariance(n, seed):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal((n, n)) * 30.
    return A.dot(A.T) + 1e4 * np.eye(n)


def dump(name, likelihood, size=48, seed=42):
    likelihood()
    cfg = extract_config(likelihood)
    names = [str(n) for n in cfg['__varied__']]
    theta = sample_theta(likelihood, size, seed)
    theta[3, 0] = likelihood.varied_params[names[0]].prior.limits[1] + 0.01     # one row outside the prior
    (logpost, derived), errors = vmap(likelihood, backend=None, errors='return', return_derived=True)({n: theta[:, i] for i, n in enumerate(names)})
    out = {'cfg/' + key: value for key, value in cfg.items() if not key.startswith('__')}
    out.update(names=np.array(names), theta=theta, loglikelihood=np.asarray(derived[likelihood._param_loglikelihood]), logprior=np.asarray(derived[likelihood._param_logprior]),
               logposterior=np.asarray(logpost))
    fn = os.path.join(here, 'boundary_{}.npz'.format(name))
    np.savez_compressed(fn, **out)
    print('saved', fn, '{:.1f} kB'.format(os.path.getsize(fn) / 1e3), 'keys', len(cfg) - 1, 'errors', len(errors))


def main():
    kedges = np.linspace(0., 0.2, 41)
    # (1) BASELINE configs[1]: ShapeFit + Kaiser, dense survey-like window
    kin, wmat = dense_window(kedges, (0, 2, 4))
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix=wmat, kin=kin, ellsin=(0, 2, 4), theory=theory, shotnoise=1e4)
    dump('cfg2_dense', ObservablesGaussianLikelihood(observables=[obs], covariance=covariance(120, 1)))
    # (2) BASELINE configs[4] geometry: two tracers (namespaced b1 / sn0, shared ShapeFit parameters), joint covariance, binning windows
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    observables = []
    for tracer, kmax, b1, shotnoise in [('LRG', 0.2, 2., 1e4), ('ELG', 0.15, 1.3, 4e3)]:
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        nk = int(round(kmax / 0.005))
        observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1}, kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 4},
                                                                   theory=theory, shotnoise=shotnoise))
    dump('two_tracers', ObservablesGaussianLikelihood(observables=observables, covariance=covariance(210, 2)))
    # (3) EFT-like Kaiser (counter / stochastic terms), qisoqap AP mode
    template = ShapeFitPowerSpectrumTemplate(z=0.5, apmode='qisoqap')
    theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.8, 'ct0_2': 1.5}, kedges=kedges, ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    dump('eft_qisoqap', ObservablesGaussianLikelihood(observables=[obs], covariance=covariance(120, 3)))


if __name__ == '__main__':
    main()
