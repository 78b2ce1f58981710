"""Time the PNG theory on its fixture's configuration (50 theory k x 20 mu, 1000 template knots, two splines per point): python tools/time_png.py [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from test_oracle_png import load   # noqa: E402
from test_gpu_png import spec_from_png_golden   # noqa: E402
from desilike_amd._lib import Context   # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    g = load('png_bphi_shapefit')
    ctx = Context(spec_from_png_golden(g), device=0)
    rng = np.random.RandomState(0)
    theta = np.tile(g['theta'][0], (B, 1)) * (1. + 0.01 * rng.standard_normal((B, g['theta'].shape[1])))
    th = torch.as_tensor(theta, device='cuda:0')
    out = torch.empty(B, dtype=torch.float64, device='cuda:0')
    for _ in range(3): ctx.eval_logposterior(th, out)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): ctx.eval_logposterior(th, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('PNG, B = {:d}: {:.1f} us per call, {:.0f} evals/s; finite: {}'.format(B, dt * 1e6, B / dt, bool(torch.isfinite(out).all())))


if __name__ == '__main__':
    main()
