"""Time the TNS one-loop path on the fixture's configuration (n_k = 120, n_k11 = 192, 500 template wavenumbers, 10 cosines): python tools/time_tns.py [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from bench_configs import load_tns as load   # noqa: E402
from bench_configs import spec_from_tns_golden   # noqa: E402
from desilike_amd._lib import Context   # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    name = sys.argv[2] if len(sys.argv) > 2 else 'tns'
    g = load(name)
    ctx = Context(spec_from_tns_golden(g), device=0)
    names = [str(n) for n in g['names']]
    rng = np.random.RandomState(0)
    base = g['theta'][0]
    theta = np.tile(base, (B, 1)) * (1. + 0.01 * rng.standard_normal((B, len(names))))
    dev = torch.device('cuda', 0)
    th = torch.as_tensor(theta, device=dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(3): ctx.eval_logposterior(th, out)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): ctx.eval_logposterior(th, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    n11, nq, nmu = len(g['k11_table']), len(g['c.k11']), 10
    flop = 2. * n11 * nq * nmu * 27 + 2. * n11 * nq * 12   # the 27 bilinear + 12 linear tables the reference integrates (algorithmic count; the kernel executes 32 + 16 columns)
    print('B = {:d}: {:.1f} us per call, {:.3f} us per point, {:.0f} evals/s; loop GEMM {:.1f} MFLOP per point (algorithmic) -> {:.1f} TFLOP/s end to end ({:.2f} of the fp64 matrix peak)'.format(
        B, dt * 1e6, dt * 1e6 / B, B / dt, flop / 1e6, flop * B / dt / 1e12, flop * B / dt / 78.6e12))
    print('finite:', bool(torch.isfinite(out).all()))


if __name__ == '__main__':
    main()
