"""Bit-repeatability stress of one batch size: ``python tools/stress_repeat.py B [repeats]`` evaluates a seeded batch of B points of config 2 (dense window) ``repeats``
times and reports every call whose results differ from the first (fixed summation orders everywhere: any difference is a race).  Run several copies at once to load the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from golden_utils import load_golden, spec_from_golden
from desilike_amd._lib import Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2537
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 500
g = load_golden('cfg2_shapefit_window_dense')
ctx = Context(spec_from_golden(g), device=0)
rng = np.random.RandomState(5)
theta = torch.as_tensor(rng.uniform([0.9, 0.9, -0.5, 0.5, 0.5, -3.], [1.1, 1.1, 0.5, 1.5, 3.5, 3.], size=(B, 6)), dtype=torch.float64, device='cuda').contiguous()
first, out = None, torch.empty(B, dtype=torch.float64, device='cuda')
bad = 0
for it in range(repeats):
    ctx.eval_batch(theta, loglike=out)
    torch.cuda.synchronize()
    if first is None: first = out.clone()
    elif not torch.equal(first, out):
        diff = (first - out).abs()
        rows = torch.nonzero(diff > 0).flatten().cpu().numpy()
        bad += 1
        if bad <= 5: print('call %d: %d rows differ, max |diff| %.3e, rows %s' % (it, rows.size, float(diff.max()), rows[:12]))
print('B = %d: %d of %d calls differ from the first' % (B, bad, repeats))
