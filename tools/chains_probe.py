"""K chains (device-resident 512-walker ensembles on the two-tracer likelihood) on K streams of one GPU: wall time per update, and -- under
``rocprofv3 --kernel-trace`` -- the kernel timeline (tools/trace_overlap.py summarises how much the chains' kernels overlap).
    python tools/chains_probe.py K [iterations]"""
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
sampler.run(niterations=50)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    runners = [sampler._get_runner(i) for i in range(K)]
    for runner in runners: runner.enqueue(iterations)
    t1 = time.perf_counter()
    outs = [runner.collect(device=True) for runner in runners]
    t2 = time.perf_counter()
    print('K = %d: enqueue %.2f ms, device done after %.2f ms: %.1f us per update per chain, %.2f M evals/s' % (K, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e6 * (t2 - t0) / iterations, K * 512 * iterations / (t2 - t0) / 1e6))
