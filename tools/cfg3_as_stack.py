"""configs[2] (ONE network for the whole table) through the kernel of the stacked layout: the same description with ``emu0.type = 2``, one network, one group of 19
monomials (the host splits it into device groups of <= 5), amplitude 1.  Same folded operator, same columns: the two kernels are compared on the same numbers.
Probe: docs/EXPERIMENTS.md."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from bench_configs import make_cfg3_full
from desilike_amd._lib import Context

B = 4096
for marg in (True, False):
    g, like, pt, theory, solved = make_cfg3_full(marg=marg)
    like.initialize()
    spec = like._spec({}, like._flatdata_list(), like.precision)
    emu = spec['observables'][0]['emu0']
    nx = len(np.asarray(emu['xlimits']).reshape(-1, 2))
    spec2 = dict(spec); spec2['observables'] = [dict(spec['observables'][0])]
    spec2['observables'][0]['emu0'] = dict(emu, type=np.array([2], dtype='i4'), groups=np.array([0, 1, 0, 19], dtype='i4'), scale=np.zeros(nx + 1))
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=B, random_state=rng), *param.prior.limits) for param in like.varied_params])
    th = torch.as_tensor(theta, dtype=torch.float64, device='cuda').contiguous()
    outs = []
    for name, s in (('single-network kernel', spec), ('stacked kernel, 1 network', spec2)):
        ctx = Context(s, device=0)
        out = torch.empty(B, dtype=torch.float64, device='cuda'); st = torch.empty(B, dtype=torch.int32, device='cuda')
        for _ in range(300): ctx.eval_logposterior(th, out, status=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): ctx.eval_logposterior(th, out, status=st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        outs.append(out.cpu().numpy())
        print('marg = %d  %-28s %.2f us per %d points, %.2f M evals/s, status ok: %s' % (marg, name, 1e6 * dt, B, B / dt / 1e6, bool((st == 0).all())))
        ctx.close()
    err = np.abs(outs[1] - outs[0]) / np.maximum(1., np.abs(outs[0]))
    print('   max relative difference of the log-posteriors: %.2e' % err.max())
