"""The host-array entry point (``dl_eval_logposterior_host`` / ``dl_eval_batch_host``: what the reference-side binding and any unmodified desilike sampler call, samplers/base.py:144-200):
time per call at 1 / 16 / 256 / 1024 points, 1000 calls each, median / p99 / mean, for every DL_HOST_MODE (0 staged copies + stream synchronisation, 1 mapped buffers +
stream synchronisation, 2 mapped + event polling, 3 mapped + completion flag = the default), on BASELINE configs[1] (one tracer, 1024-point config) and the two-tracer
likelihood of configs[4].  Results of every mode are compared bit for bit with mode 0."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from bench import make_likelihood, make_likelihood_config5, sample_theta
from desilike_amd._lib import refresh_options as _refresh_options   # the library reads its DL_* switches once per process

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out = {}
for label, like in [('configs[1]', make_likelihood(0)), ('configs[4]', make_likelihood_config5(0))]:
    ctx = like._get_posterior_context()[0]
    theta_all = np.ascontiguousarray(sample_theta(like, 1024, 42))
    for B in (1, 16, 256, 1024):
        theta = np.ascontiguousarray(theta_all[:B])
        ref = None
        for mode in (0, 1, 2, 3):
            os.environ['DL_HOST_MODE'] = str(mode); _refresh_options()
            for _ in range(50): ctx.eval_logposterior_host(theta)
            t = np.empty(ncalls)
            for i in range(ncalls):
                t0 = time.perf_counter_ns()
                res = ctx.eval_logposterior_host(theta)
                t[i] = 1e-3 * (time.perf_counter_ns() - t0)
            res = np.asarray(res[0] if isinstance(res, tuple) else res)
            if ref is None: ref = res
            same = bool(np.array_equal(ref, res, equal_nan=True))
            out['{} B={} mode={}'.format(label, B, mode)] = dict(median_us=float(np.median(t)), p99_us=float(np.percentile(t, 99)), mean_us=float(t.mean()), min_us=float(t.min()), identical_to_mode0=same)
            print('%-11s B=%5d mode %d: median %7.1f us  p99 %7.1f  mean %7.1f  min %7.1f  identical %s' % (label, B, mode, np.median(t), np.percentile(t, 99), t.mean(), t.min(), same), flush=True)
# the loglikelihood / logprior / status variant (dl_eval_batch_host), default mode
os.environ.pop('DL_HOST_MODE', None); _refresh_options()
like = make_likelihood(0)
ctx = like._get_context()
theta = np.ascontiguousarray(sample_theta(like, 256, 42))
for _ in range(50): ctx.eval_batch_host(theta)
t = np.empty(ncalls)
for i in range(ncalls):
    t0 = time.perf_counter_ns(); ctx.eval_batch_host(theta); t[i] = 1e-3 * (time.perf_counter_ns() - t0)
out['configs[1] eval_batch_host B=256'] = dict(median_us=float(np.median(t)), p99_us=float(np.percentile(t, 99)), mean_us=float(t.mean()))
print('configs[1] dl_eval_batch_host B=256: median %.1f us  p99 %.1f  mean %.1f' % (np.median(t), np.percentile(t, 99), t.mean()))
print(json.dumps(out))
