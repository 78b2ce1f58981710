"""BASELINE configs[4] end to end on one GPU: 512 walkers x two config-2 tracers (n = 240), device-resident ensemble (dl_ensemble_*) against the host-driven
stretch move (NumPy proposals around one dl_eval_logposterior_host call per half-step); wall time per ensemble update."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_likelihood_config5
from desilike_amd.samplers import EmceeSampler

like = make_likelihood_config5(0)
for device_resident in (True, False):
    sampler = EmceeSampler(like, nwalkers=512, seed=42, use_emcee=False, device_resident=device_resident)
    sampler.run(niterations=300 if device_resident else 20)
    torch.cuda.synchronize()
    niter = 500 if device_resident else 200
    t0 = time.perf_counter()
    chain = sampler.run(niterations=niter)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / niter
    print('%-46s ensemble update (512 walkers): %7.1f us = %5.2f M evals/s; acceptance %.2f' % (
        'device-resident (dl_ensemble_run, chain drained once)' if device_resident else 'host-driven (NumPy stretch move + host calls)', 1e6 * dt, 512 / dt / 1e6, sampler.acceptance_fraction.mean()))
theta = np.column_stack([chain[p.name][-1][:256] for p in like.varied_params])
ctx = like._get_posterior_context()[0]
t0 = time.perf_counter()
for _ in range(200): ctx.eval_logposterior_host(theta)
print('dl_eval_logposterior_host(256 points): %.1f us' % (1e6 * (time.perf_counter() - t0) / 200))
