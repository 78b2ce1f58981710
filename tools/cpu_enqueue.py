import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
import bench
lik = bench.make_likelihood(0)
ctx = lik._get_context()
B = 1024
theta = torch.as_tensor(bench.sample_theta(lik, B, seed=42), dtype=torch.float64, device='cuda').contiguous()
ll = torch.empty(B, dtype=torch.float64, device='cuda'); lp = torch.empty_like(ll); st = torch.empty(B, dtype=torch.int32, device='cuda')
stream = torch.cuda.current_stream()
for _ in range(50): ctx.eval_batch(theta, loglike=ll, logprior=lp, status=st, stream=stream.cuda_stream)
torch.cuda.synchronize()
for n in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(n): ctx.eval_batch(theta, loglike=ll, logprior=lp, status=st, stream=stream.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('n=%d enqueue %.2f us/step, total %.2f us/step' % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
