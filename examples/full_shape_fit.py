"""A full-shape fit from template to chains on one MI355X, with the reference's class names (cf. desilike's README example):

    python examples/full_shape_fit.py [output directory]

ShapeFit template -> Kaiser tracer multipoles -> windowed P_ell observable (mock data generated from the theory itself) -> Gaussian likelihood with the shot-noise
term marginalised analytically -> posterior maximum (batched Levenberg-Marquardt) -> Fisher matrix -> Metropolis-Hastings chains seeded by the maximiser's covariance
(device-resident, convergence by Gelman-Rubin) -> chain files in desilike's own format.  The synthetic analytic fiducial stands in for cosmoprimo: pass a
``TabulatedFiducial(k, pk_dd, f, pknow_dd)`` for real data."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles   # noqa: E402
from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable, ObservablesCovarianceMatrix, BoxFootprint   # noqa: E402
from desilike_amd.likelihoods import ObservablesGaussianLikelihood   # noqa: E402
from desilike_amd.profilers import GaussNewtonProfiler   # noqa: E402
from desilike_amd.fisher import Fisher   # noqa: E402
from desilike_amd.samplers import MCMCSampler   # noqa: E402


def main(outdir=None, quick=False):
    template = ShapeFitPowerSpectrumTemplate(z=0.8, fiducial='synthetic', apmode='qiso')      # (the synthetic spectrum is close to a power law: one dilation parameter)
    nbar = 2e-2
    theory = KaiserTracerPowerSpectrumMultipoles(template=template, shotnoise=1. / nbar)        # (as in the reference, the scale of sn0 is the theory's own argument)
    truth = {'b1': 1.8, 'qiso': 1.01, 'dm': 0.01, 'df': 0.98, 'sn0': 0.1}
    observable = TracerPowerSpectrumMultipolesObservable(data=truth, kedges=np.linspace(0.01, 0.2, 39), ells=(0, 2, 4), wmatrix={'resolution': 5}, theory=theory, shotnoise=1. / nbar)
    # Gaussian covariance of the multipoles for a 20 (Gpc / h)^3 box with nbar = 3e-3 (the theory multipoles at the fiducial parameters, on the device)
    covariance = ObservablesCovarianceMatrix(observable, footprints=BoxFootprint(volume=2e10, nbar=nbar), resolution=3)(**truth)
    likelihood = ObservablesGaussianLikelihood(observables=[observable], covariance=covariance)
    likelihood.all_params = {'sn0': {'derived': '.marg'}}          # analytic marginalisation of the shot-noise term
    print('varied:', likelihood.varied_params.names(), '| marginalised:', likelihood.solved_params.names())

    profiler = GaussNewtonProfiler(likelihood, seed=1)
    profiles = profiler.maximize(niterations=4)
    best = profiles.choice()
    print('posterior maximum:', {name: round(value, 5) for name, value in best.items()})
    intervals = profiler.interval(params=['qiso', 'b1'], cl=1., size=9)
    print('1-sigma intervals:', {name: tuple(round(float(v), 5) for v in limits) for name, limits in intervals.items()})

    fisher = Fisher(likelihood)(**{name: best[name] for name in profiler.fisher.varied_params.names()})
    print('Fisher errors    :', {name: round(float(value), 5) for name, value in zip(fisher.names(), fisher.std())})

    save_fn = None if outdir is None else os.path.join(outdir, 'chain_*.npy')
    sampler = MCMCSampler(likelihood, chains=8, covariance=profiles, seed=2, save_fn=save_fn)
    names = likelihood.varied_params.names()
    start = np.tile([best[name] for name in names], (8, 1))
    chains = sampler.run(start=start, check_every=200 if quick else 500, min_iterations=400 if quick else 1000, max_iterations=1200 if quick else 20000,
                         check={'max_eigen_gr': 0.05 if quick else 0.02})
    x = np.concatenate([np.column_stack([chain[name] for name in names])[len(chain['fweight']) // 2:] for chain in chains])
    w = np.concatenate([chain['fweight'][len(chain['fweight']) // 2:] for chain in chains])
    mean = np.average(x, weights=w, axis=0)
    std = np.sqrt(np.average((x - mean)**2, weights=w, axis=0))
    print('Gelman-Rubin - 1 :', round(float(sampler.diagnostics['eigen_gr'][-1]), 4), '| acceptance', round(float(sampler.acceptance_rate.mean()), 3))
    for name, m, s in zip(names, mean, std):
        print('  {:6s} = {:8.4f} +/- {:.4f}   (truth {:.4f})'.format(name, m, s, truth[name]))
    return {'best': best, 'mean': dict(zip(names, mean)), 'std': dict(zip(names, std)), 'truth': truth, 'eigen_gr': float(sampler.diagnostics['eigen_gr'][-1])}


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else None)
