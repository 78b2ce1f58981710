"""Host-side profile of ``EmceeSampler.run`` (device-resident) on the config-5 likelihood: where the wall time beyond the device's goes."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_likelihood_config5
from desilike_amd.samplers import EmceeSampler

like = make_likelihood_config5(0)
sampler = EmceeSampler(like, nwalkers=512, seed=42, use_emcee=False, device_resident=True)
sampler.run(niterations=300)
torch.cuda.synchronize()
for niter in (100, 500, 2000):
    t0 = time.perf_counter(); sampler.run(niterations=niter); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('run(niterations=%d): %.1f ms = %.1f us per update' % (niter, 1e3 * dt, 1e6 * dt / niter))
pr = cProfile.Profile(); pr.enable()
sampler.run(niterations=500); torch.cuda.synchronize()
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(22); print(out.getvalue()[:5000])
