"""Host-side latency of the reference's call surfaces on the GPU path (BASELINE config 2 likelihood): likelihood(**params) (scalar), vmap(likelihood)(dict) at 256 points,
Context.eval_batch_host / eval_logposterior_host at 1 and 256 points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from desilike_amd import vmap

like = bench.make_likelihood(0)
ctx = like._get_context()
names = like.varied_params.names()
theta = bench.sample_theta(like, 256, seed=1)
point = dict(zip(names, theta[0]))
points = {name: theta[:, i] for i, name in enumerate(names)}
vlike = vmap(like, errors='return', return_derived=True)


def timeit(label, func, n=300):
    for _ in range(20): func()
    t0 = time.perf_counter()
    for _ in range(n): func()
    print('%-46s %8.1f us per call' % (label, 1e6 * (time.perf_counter() - t0) / n))


timeit('likelihood(**params)', lambda: like(**point))
timeit('vmap(likelihood)(256 points)', lambda: vlike(points))
timeit('Context.eval_batch_host(1 point)', lambda: ctx.eval_batch_host(theta[:1]))
timeit('Context.eval_batch_host(256 points)', lambda: ctx.eval_batch_host(theta))
timeit('Context.eval_logposterior_host(256 points)', lambda: ctx.eval_logposterior_host(theta))
