"""Where EmceeSampler(chains=K).run spends its wall time on one GPU (verdict r4 item 6: K = 2 ran 99 us per update per chain against 67 in tools/chains_probe.py).
    python tools/chains_run_probe.py K [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_likelihood_config5
from desilike_amd.samplers import EmceeSampler
from desilike_amd.parallel import WalkerSharding

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 300
like = make_likelihood_config5(0)
sampler = EmceeSampler(like, nwalkers=512, chains=K, seed=42, sharding=WalkerSharding(group=False), device_resident=True)
sampler.run(niterations=iterations)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    sampler.run(niterations=iterations)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('K = %d: sampler.run %.2f ms = %.1f us per update per chain, %.2f M evals/s' % (K, 1e3 * (t1 - t0), 1e6 * (t1 - t0) / iterations, K * 512 * iterations / (t1 - t0) / 1e6))
# the same batch by hand, timed piece by piece
local = sampler.local_chains()
runners = [sampler._get_runner(i) for i in local]
for rep in range(2):
    marks = [('start', time.perf_counter())]
    for runner in runners: runner.enqueue(iterations)
    marks.append(('enqueue', time.perf_counter()))
    for i, runner in enumerate(runners):
        out = runner.collect()
        marks.append(('collect %d' % i, time.perf_counter()))
        acc = np.asarray(runner.naccepted, dtype='f8')
        marks.append(('naccepted %d' % i, time.perf_counter()))
        it = runner.iteration
        marks.append(('iteration %d' % i, time.perf_counter()))
        sampler._blocks[local[i]].append(*out)
        marks.append(('append %d' % i, time.perf_counter()))
    print('   by hand: ' + ', '.join('%s +%.2f ms' % (name, 1e3 * (t - marks[j][1])) for j, (name, t) in enumerate(marks[1:])) + ' | total %.2f ms' % (1e3 * (marks[-1][1] - marks[0][1])))
