"""How long after the last kernel does the host notice?  20-step runs closed by torch.cuda.synchronize alone, or by polling an event recorded after the last step
(then the same synchronize, which returns at once)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_likelihood, sample_theta

like = make_likelihood(0)
ctx = like._get_context()
B, K = 1024, 20
dev = torch.device('cuda', 0)
theta = torch.as_tensor(sample_theta(like, B, 42), dtype=torch.float64, device=dev).contiguous()
ll, lp = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream(dev)


def run(poll):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): ctx.eval_batch(theta, loglike=ll, logprior=lp, status=st, stream=s.cuda_stream)
    if poll:
        ev = torch.cuda.Event()
        ev.record(s)
        while not ev.query(): pass
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / K


for _ in range(50): run(False)
for mode in (False, True, False, True):
    t = np.array([run(mode) for _ in range(200)])
    print('poll' if mode else 'sync', 'us/step: median %.2f  p10 %.2f  p90 %.2f' % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)))
