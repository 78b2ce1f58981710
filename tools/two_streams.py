"""Throughput with several independent walker ensembles in flight on separate HIP streams (one context each): the three kernels of a 1024-point step are
latency-bound and leave most of the chip idle, so independent steps overlap.  Not the bench metric (bench.py times sequential steps): an additional number."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from desilike_amd._lib import Context

like = bench.make_likelihood(0)
spec = like._spec({}, like._flatdata_list(), like.precision)
B = 1024
for nstreams in (1, 2, 3, 4):
    ctxs = [Context(spec, device=0) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    thetas = [torch.as_tensor(bench.sample_theta(like, B, seed=42 + i), dtype=torch.float64, device='cuda').contiguous() for i in range(nstreams)]
    outs = [torch.empty(B, dtype=torch.float64, device='cuda') for _ in range(nstreams)]
    def run(steps):
        for it in range(steps):
            i = it % nstreams
            ctxs[i].eval_logposterior(thetas[i], outs[i], stream=streams[i].cuda_stream)
    run(40); torch.cuda.synchronize()
    steps = 400
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('%d stream(s): %.1f us per 1024-point step, %.1f M evals/s' % (nstreams, 1e6 * dt, B / dt / 1e6))
    for c in ctxs: c.close()
