"""One 1024-point step split into C chunks on C HIP streams (one context replica each), (a) free-running -- the chunks are independent chains, (b) joined -- every step
forks from and joins the caller's stream through events, what a dependent sampler step would need.  Probe only: docs/EXPERIMENTS.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

like = bench.make_likelihood(0)
B = 1024
theta = torch.as_tensor(bench.sample_theta(like, B, seed=42), dtype=torch.float64, device='cuda').contiguous()
ref = torch.empty(B, dtype=torch.float64, device='cuda')
like._get_context().eval_logposterior(theta, ref)
torch.cuda.synchronize()
for C in (1, 2, 4):
    n = B // C
    ctxs = [like._get_context(replica=i) for i in range(C)]
    streams = [torch.cuda.Stream() for _ in range(C)]
    main = torch.cuda.current_stream()
    out = torch.empty(B, dtype=torch.float64, device='cuda')
    th = [theta[i * n:(i + 1) * n].contiguous() for i in range(C)]
    ou = [out[i * n:(i + 1) * n] for i in range(C)]
    fork = torch.cuda.Event(); joins = [torch.cuda.Event() for _ in range(C)]

    def free(steps):
        for _ in range(steps):
            for i in range(C): ctxs[i].eval_logposterior(th[i], ou[i], stream=streams[i].cuda_stream)

    def joined(steps):
        for _ in range(steps):
            fork.record(main)
            for i in range(C):
                streams[i].wait_event(fork)
                ctxs[i].eval_logposterior(th[i], ou[i], stream=streams[i].cuda_stream)
                joins[i].record(streams[i])
            for i in range(C): main.wait_event(joins[i])

    for name, fn in (('free-running', free), ('joined', joined)):
        fn(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(400); t1 = time.perf_counter(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
        print('%d chunk(s) of %4d, %-12s: %.1f us per 1024-point step (host enqueue %.1f us), same bits: %s' % (C, n, name, 1e6 * dt, 1e6 * (t1 - t0) / 400, bool(torch.equal(out, ref))))
