"""Time series of the 256-point host call per DL_HOST_MODE: where the slow calls are (periodic? clustered?) -- diagnosis of the tail of tools/time_host_call.py."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from bench import make_likelihood_config5, sample_theta
from desilike_amd._lib import refresh_options as _refresh_options   # the library reads its DL_* switches once per process

like = make_likelihood_config5(0)
ctx = like._get_posterior_context()[0]
theta = np.ascontiguousarray(sample_theta(like, 256, 42))
n = 4000
gc.disable()
for mode in sys.argv[1:] or ['1', '3']:
    os.environ['DL_HOST_MODE'] = mode; _refresh_options()
    for _ in range(100): ctx.eval_logposterior_host(theta)
    t = np.empty(n)
    for i in range(n):
        t0 = time.perf_counter_ns(); ctx.eval_logposterior_host(theta); t[i] = 1e-3 * (time.perf_counter_ns() - t0)
    med = np.median(t)
    slow = np.flatnonzero(t > 1.3 * med)
    print('mode %s: median %.1f p99 %.1f p99.9 %.1f max %.1f; %d calls > 1.3 x median; gaps between them: %s' % (mode, med, np.percentile(t, 99), np.percentile(t, 99.9), t.max(), slow.size, np.diff(slow)[:40].tolist()))
    print('   slow values:', np.round(t[slow][:40], 1).tolist())
