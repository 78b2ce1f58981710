import cProfile, pstats, sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bench
from desilike_amd.samplers import MCMCSampler
from desilike_amd.parallel import WalkerSharding
like = bench.make_likelihood_config5(0)
s = MCMCSampler(like, chains=256, vectorize=4, seed=42, sharding=WalkerSharding(group=False))
s.run(check_every=300, max_iterations=900); s.learn = False
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); s.run(check_every=300, max_iterations=300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print('us per try', dt / 300 * 1e6)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
