"""Repeatability of the stacked emulator kernel (dl_emulated_stacked_kernel): the same batch evaluated N times must give bit-identical results (a missing barrier shows up as
an occasional difference), with and without solved parameters, at a full and at ragged batch sizes.
    python tools/stress_stacked.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from bench_configs import make_cfg3_stacked

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for marg in (True, False):
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=marg)
    ctx = like._get_context()
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=4096, random_state=rng), *param.prior.limits) for param in like.varied_params])
    for B in (4096, 1000, 17):
        th = torch.as_tensor(theta[:B], dtype=torch.float64, device='cuda').contiguous()
        ref = torch.empty(B, dtype=torch.float64, device='cuda')
        out = torch.empty(B, dtype=torch.float64, device='cuda')
        st = torch.empty(B, dtype=torch.int32, device='cuda')
        ctx.eval_logposterior(th, ref, status=st)
        torch.cuda.synchronize()
        bad = 0
        for it in range(N):
            ctx.eval_logposterior(th, out, status=st)
            if it % 10 == 9 or it == N - 1:
                torch.cuda.synchronize()
                bad += int((out != ref).sum().item())
        print('marg = %d, B = %4d: %d repetitions, %d differing values, all finite: %s' % (marg, B, N, bad, bool(torch.isfinite(ref).all())))
        assert bad == 0
print('stress ok')
