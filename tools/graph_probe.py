"""Probe: the three kernels of a 1024-point step replayed from a HIP graph (torch.cuda.CUDAGraph capture of dl_eval_batch) against plain stream launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_likelihood, sample_theta

like = make_likelihood(0)
ctx = like._get_context()
B = 1024
dev = torch.device('cuda', 0)
theta = torch.as_tensor(sample_theta(like, B, 42), dtype=torch.float64, device=dev).contiguous()
ll, lp = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)


def run_plain(steps):
    s = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): ctx.eval_batch(theta, loglike=ll, logprior=lp, status=st, stream=s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for _ in range(3): run_plain(20)
print('plain stream launches : %.2f us per step' % (1e6 * run_plain(400)))
ref = ll.clone()
for per_graph in (1, 8):
    with torch.cuda.stream(side):
        for _ in range(3): ctx.eval_batch(theta, loglike=ll, logprior=lp, status=st, stream=side.cuda_stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(per_graph): ctx.eval_batch(theta, loglike=ll, logprior=lp, status=st, stream=side.cuda_stream)
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    n = 400 // per_graph
    t0 = time.perf_counter()
    for _ in range(n): graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * per_graph)
    print('graph of %d step(s)     : %.2f us per step (results identical: %s)' % (per_graph, 1e6 * dt, bool(torch.equal(ll, ref))))
