"""Config 3 alone (MLP tables + 5 marginalised parameters, SURVEY 8d size, 4096 points): the workload for profiling the fused emulator / feature-GEMM kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from time_configs import time_likelihood
from bench_configs import make_cfg3_full
g, like, pt, theory, solved = make_cfg3_full(marg=True)
time_likelihood('cfg3 (SURVEY 8d size): MLP tables + 5 marginalised parameters', like, 4096, steps=int(os.environ.get('STEPS', 40)))
