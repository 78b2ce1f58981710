"""Time the device-resident blocked Metropolis-Hastings sampler on the two-tracer likelihood of BASELINE configs[4] (8 parameters, n = 240):
python tools/time_mh.py [chains] [vectorize] [tries]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from bench_configs import make_cfg5   # noqa: E402
from desilike_amd._lib import DeviceMH   # noqa: E402


def main():
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    vectorize = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ntries = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    g, like = make_cfg5()
    ctx, offset = like._get_posterior_context()
    P = ctx.n_params
    center = np.array([param.value for param in like.varied_params], dtype='f8')
    sigma = np.array([0.01 * max(abs(v), 0.5) for v in center])
    start = center + 0.3 * sigma * np.random.RandomState(0).standard_normal((chains, P))
    mh = DeviceMH(ctx, chains, vectorize=vectorize, seed=1, offset=offset)
    mh.set_covariance(np.diag(sigma))
    mh.set_state(start)
    mh.run(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = mh.run(ntries)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    count = out[3].cpu().numpy()
    print('chains = {:d}, vectorize = {:d}: {:.1f} us per try ({:d} rows), {:.2f} M evals/s, {:.0f} accepted moves / s, acceptance per try {:.2f}'.format(
        chains, vectorize, dt / ntries * 1e6, chains * vectorize, chains * vectorize * ntries / dt / 1e6, count.sum() / dt, count.mean() / ntries))


if __name__ == '__main__':
    main()
