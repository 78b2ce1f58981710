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


def dump(name, likelihood, size=48, seed=42, unsolved=None):
    """``unsolved``: for likelihoods with analytically solved parameters (the reference's ``_solve`` needs jax: not runnable here) the same pipeline WITHOUT the
    '.marg' / '.best' flags -- the reference evaluates that one on a stencil of the solved parameters, which gives the exact quadratic form (value, gradient, Hessian)
    of its log-posterior in them; the closed forms of the marginalised / profiled posterior follow (the pin of tests/golden/make_golden.py::marg_multi)."""
    try:
        likelihood()
    except ModuleNotFoundError as exc:   # analytic solve (likelihoods/base.py:130: jax): the pipeline itself has been initialised and calculated by then
        assert 'jax' in str(exc) and unsolved is not None
    cfg = extract_config(likelihood)
    names = [str(n) for n in cfg['__varied__']]
    theta = sample_theta(likelihood, size, seed)
    theta[3, 0] = likelihood.varied_params[names[0]].prior.limits[1] + 0.01     # one row outside the prior
    out = {'cfg/' + key: value for key, value in cfg.items() if not key.startswith('__')}
    out.update(names=np.array(names), theta=theta)
    solved = [str(n) for n in cfg['__solved__']]
    if solved:
        out['solved'] = np.array(solved)
        unsolved()
        x0 = np.array([likelihood.all_params[n].value for n in solved])
        steps = np.array([max(abs(likelihood.all_params[n].proposal or 1.), 1e-2) * 5. for n in solved])
        ns = len(solved)
        rows = []
        for row in theta[:12]:
            base = dict(zip(names, row))

            def logpost(x): return unsolved(**{**base, **dict(zip(solved, x))})

            f0 = logpost(x0)
            g, H, fp = np.zeros(ns), np.zeros((ns, ns)), np.zeros(ns)
            for i in range(ns):
                e = np.zeros(ns); e[i] = steps[i]
                fp[i], fm = logpost(x0 + e), logpost(x0 - e)
                g[i], H[i, i] = (fp[i] - fm) / (2. * steps[i]), (fp[i] - 2. * f0 + fm) / steps[i]**2
            for i in range(ns):
                for j in range(i + 1, ns):
                    e = np.zeros(ns); e[i], e[j] = steps[i], steps[j]
                    H[i, j] = H[j, i] = (logpost(x0 + e) - fp[i] - fp[j] + f0) / (steps[i] * steps[j])
            rows.append((f0, g, H))
        out.update(marg_x0=x0, marg_c=np.array([r[0] for r in rows]), marg_g=np.array([r[1] for r in rows]), marg_H=np.array([r[2] for r in rows]))
        errors = {}
    else:
        (logpost, derived), errors = vmap(likelihood, backend=None, errors='return', return_derived=True)({n: theta[:, i] for i, n in enumerate(names)})
        out.update(loglikelihood=np.asarray(derived[likelihood._param_loglikelihood]), logprior=np.asarray(derived[likelihood._param_logprior]), logposterior=np.asarray(logpost))
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
    # (4) BASELINE configs[3]: damped-BAO xi_ell (ell = 0, 2; 30 s-bins), 'power' broadband; and the P_ell version with a binning window
    from desilike.theories.galaxy_clustering import (BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles, DampedBAOWigglesTracerPowerSpectrumMultipoles,
                                                     KaiserTracerCorrelationFunctionMultipoles)
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable

    def bao_likelihood(space, marg=False):
        template = BAOPowerSpectrumTemplate(z=0.5)
        if space == 'xi':
            theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='reciso')
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        else:
            theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template)
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
            n, scale = 112, 30.
        for name in ['sigmapar', 'sigmaper']:
            theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
        if marg:
            for param in theory.init.params.select(basename='al*'):
                param.update(derived='.marg')
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        return ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (10. * scale)**2 * np.eye(n))

    dump('cfg4_xi', bao_likelihood('xi'), size=24, seed=9)
    dump('cfg4_pk', bao_likelihood('pk'), size=24, seed=9)
    # (5) the DESI-style BAO fit: every broadband term solved analytically
    dump('cfg4_xi_marg', bao_likelihood('xi', marg=True), size=24, seed=9, unsolved=bao_likelihood('xi'))
    # (6) Kaiser xi_ell (ShapeFit template)
    theory = KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory)
    rng = np.random.RandomState(14)
    A = rng.standard_normal((90, 90)) * 3e-4
    dump('kaiser_xi', ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (3e-3)**2 * np.eye(90)), size=24, seed=19)
    # (7) two tracers with both shot-noise terms marginalised (Gaussian priors)
    template = ShapeFitPowerSpectrumTemplate(z=0.5)

    def two_tracers(marg):
        observables = []
        for tracer, kmax, b1, shotnoise in [('LRG', 0.2, 2., 1e4), ('ELG', 0.15, 1.3, 4e3)]:
            theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
            theory.init.params[tracer + '.sn0'].update(prior=dict(dist='norm', loc=0.1, scale=2.), **({'derived': '.marg'} if marg else {}))
            nk = int(round(kmax / 0.005))
            observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1, tracer + '.sn0': 0.3}, kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4),
                                                                       wmatrix={'resolution': 4}, theory=theory, shotnoise=shotnoise))
        return ObservablesGaussianLikelihood(observables=observables, covariance=covariance(210, 22))

    dump('two_tracers_marg', two_tracers(True), size=24, seed=24, unsolved=two_tracers(False))


if __name__ == '__main__':
    main()
