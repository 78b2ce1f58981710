"""Time the batched HMC sampler on BASELINE config 2 (6 parameters, analytic gradient): python tools/time_hmc.py [chains] [integration steps] [iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from bench_configs import make_cfg2   # noqa: E402
from desilike_amd.samplers import HMCSampler   # noqa: E402


def main():
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    niter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    g, like = make_cfg2()
    sampler = HMCSampler(like, chains=chains, seed=2, num_integration_steps=nsteps, adaptation={'niterations': 150})
    sampler.run(check_every=20, max_iterations=20)       # warm-up + first batch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler.run(check_every=niter, max_iterations=niter)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('chains = {:d}, {:d} leapfrog steps: {:.1f} us per leapfrog step, {:.2f} ms per transition, {:.2f} M gradient evaluations / s, acceptance {:.2f}, step size {:.3g}'.format(
        chains, nsteps, dt / niter / nsteps * 1e6, dt / niter * 1e3, chains * nsteps * niter / dt / 1e6, sampler.acceptance_rate.mean(), sampler.step_size))


if __name__ == '__main__':
    main()
