"""Config 5: host enqueue time against device time of the ensemble run (is the loop launch-bound on the host or bound by the kernels?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import bench
from desilike_amd.samplers import EmceeSampler
from desilike_amd.parallel import WalkerSharding

likelihood = bench.make_likelihood_config5(0)
sampler = EmceeSampler(likelihood, nwalkers=512, seed=42, sharding=WalkerSharding(group=None, min_shard_rows=0), device_resident=True)
start, logposterior = sampler._get_start(512)
ens = sampler._get_ensemble()
ens.set_state(start, logposterior)
ens.run(300)
torch.cuda.synchronize()
for n in [50, 200, 1000, 3000]:
    t0 = time.perf_counter()
    ens.run(n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('iterations {:5d}: enqueue returned after {:8.1f} us/update, device done after {:8.1f} us/update'.format(n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n), flush=True)
