"""Workgroup size of the BAO theory kernel (DL_BAO_THREADS) against the batch size: per-call time of the damped-BAO xi likelihood (config 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from time_configs import time_likelihood
from bench_configs import make_cfg4
for space in ('xi', 'pk'):
    g, like = make_cfg4(space)
    for B in (32, 256, 1024, 2048, 4096, 8192, 32768):
        time_likelihood('threads %s %s' % (os.environ.get('DL_BAO_THREADS', 'auto'), space), like, B)
