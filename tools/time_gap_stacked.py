import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from bench_configs import make_cfg3_stacked
like, pt, theory, solved, networks = make_cfg3_stacked(marg=bool(int(sys.argv[1])))
ctx = like._get_context()
B = 4096
rng = np.random.RandomState(3)
theta = np.column_stack([np.clip(param.ref.sample(size=B, random_state=rng), *param.prior.limits) for param in like.varied_params])
th = torch.as_tensor(theta, dtype=torch.float64, device='cuda').contiguous()
out = torch.empty(B, dtype=torch.float64, device='cuda'); st = torch.empty(B, dtype=torch.int32, device='cuda')
for rep in range(4):
    for _ in range(20): ctx.eval_logposterior(th, out, status=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): ctx.eval_logposterior(th, out, status=st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('rep %d: %.1f us/step (launch loop %.1f us/call)' % (rep, 1e6 * (t2 - t0) / 200, 1e6 * (t1 - t0) / 200))
