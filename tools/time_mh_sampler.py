"""Per-try time of the Metropolis-Hastings sampler THROUGH MCMCSampler (enqueue, device run, drain, bookkeeping) on the config-5 likelihood, over chains x proposals x stream groups."""
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bench
from desilike_amd.samplers import MCMCSampler
from desilike_amd.parallel import WalkerSharding
for C, V, S in [(64, 4, 1), (256, 1, 1), (256, 4, 1), (256, 4, 2), (1024, 1, 1), (1024, 2, 1)]:
    like = bench.make_likelihood_config5(0)
    s = MCMCSampler(like, chains=C, vectorize=V, streams=S, seed=42, sharding=WalkerSharding(group=False))
    s.run(check_every=300, max_iterations=900); s.learn = False
    torch.cuda.synchronize()
    t0 = time.perf_counter(); s.run(check_every=300, max_iterations=300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('chains {:d} x {:d}, {:d} stream(s): {:.1f} us per try, {:.2f} M evals/s, acceptance {:.2f}'.format(C, V, S, dt / 300 * 1e6, C * V * 300 / dt / 1e6, np.nanmean(s.acceptance_rate)))
    s._runner.close()
