"""Host-array entry point at large batches: staged copies (DL_HOST_MODE=0) against mapped buffers (3): where reading theta over PCIe from every kernel stops paying."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from bench import make_likelihood, sample_theta
from desilike_amd._lib import refresh_options as _refresh_options   # the library reads its DL_* switches once per process
like = make_likelihood(0)
ctx = like._get_posterior_context()[0]
for B in (1024, 2048, 4096, 8192, 32768):
    theta = np.ascontiguousarray(sample_theta(like, B, 42))
    ref = None
    for mode in ('0', '3', '0', '3'):
        os.environ['DL_HOST_MODE'] = mode; _refresh_options()
        for _ in range(20): out = ctx.eval_logposterior_host(theta)[0]
        t = []
        for _ in range(100):
            t0 = time.perf_counter_ns(); ctx.eval_logposterior_host(theta); t.append(1e-3 * (time.perf_counter_ns() - t0))
        if ref is None: ref = out
        print('B = %6d mode %s: median %8.1f us  (%.1f M evals/s)  identical %s' % (B, mode, np.median(t), B / np.median(t), np.array_equal(ref, out, equal_nan=True)), flush=True)
