"""Config 5 end to end: EmceeSampler (built-in stretch move), 512 walkers x 2-tracer likelihood, 200 iterations on one GPU: wall time per ensemble update, split into
the GPU evaluation (two half-ensemble calls of 256 points) and the host side (proposals, conventions, copies)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_host_api import make_cfg5
from desilike_amd.samplers import EmceeSampler

g, like = make_cfg5()
sampler = EmceeSampler(like, nwalkers=512, seed=42, use_emcee=False)
sampler.run(niterations=5)
t0 = time.perf_counter()
chain = sampler.run(niterations=200)
dt = (time.perf_counter() - t0) / 200
theta = np.column_stack([chain[p.name][-1][:256] for p in like.varied_params])
t0 = time.perf_counter()
for _ in range(200): sampler.logposterior(theta)
dl = (time.perf_counter() - t0) / 200
ctx = like._get_context()
t0 = time.perf_counter()
for _ in range(200): ctx.eval_batch_host(theta)
dc = (time.perf_counter() - t0) / 200
print('ensemble update (512 walkers): %.1f us = %.2f M evals/s;  logposterior(256 points): %.1f us;  dl_eval_batch_host(256 points): %.1f us;  acceptance %.2f' % (
    1e6 * dt, 512 / dt / 1e6, 1e6 * dl, 1e6 * dc, sampler.acceptance_fraction.mean()))
