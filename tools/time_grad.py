"""Cost of the analytic gradient (dl_eval_logposterior_grad) against one evaluation and against the central-difference stencil (2 P + 1 evaluations), config 2, 1024 points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import make_likelihood, sample_theta

like = make_likelihood(0)
ctx = like._get_context()
B, P = 1024, len(like.varied_params)
theta = torch.as_tensor(sample_theta(like, B, 42), dtype=torch.float64, device='cuda').contiguous()
out = torch.empty(B, dtype=torch.float64, device='cuda')
grad = torch.empty((B, P), dtype=torch.float64, device='cuda')
stencil = theta.repeat(2 * P + 1, 1).contiguous()
out_s = torch.empty(B * (2 * P + 1), dtype=torch.float64, device='cuda')


def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


t_eval = timeit(lambda: ctx.eval_logposterior(theta, out))
t_grad = timeit(lambda: ctx.eval_logposterior_grad(theta, out, grad))
t_fd = timeit(lambda: ctx.eval_logposterior(stencil, out_s), n=50)
print('1024 points: evaluation %.1f us; value + analytic gradient %.1f us (%.2f x); central-difference stencil (%d evaluations) %.1f us (%.2f x)' % (t_eval, t_grad, t_grad / t_eval, 2 * P + 1, t_fd, t_fd / t_eval))
