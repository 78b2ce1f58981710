"""Timing of the batched device FFTLog (dl_fftlog_apply) on the reference's grid (2048 points, npad = 4096), B points x 2 multipoles resident in HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from desilike_amd.fftlog import PowerToCorrelation

k = np.logspace(-4., 3., 2048)
dev = PowerToCorrelation(k, ell=(0, 2), engine='hip', device=0)
for B in (256, 1024, 8192):
    fun = torch.randn((B, 2, 2048), dtype=torch.float64, device='cuda:0')
    out = torch.empty_like(fun)
    for _ in range(3): dev.apply_device(fun, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 20
    e0.record()
    for _ in range(steps): dev.apply_device(fun, out=out)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / steps
    nt = B * 2
    flop = nt * (2 * 5. * 2048 * 11 + 40 * 2048)     # two complex FFTs of 2048 points (5 N log2 N) + spectrum step
    print('B=%5d: %8.1f us per call, %6.2f M transforms/s, %5.2f TFLOP/s (fp64), in+out %5.1f GB/s' % (B, us, nt / us, flop / us / 1e6, nt * 2 * 2048 * 8 / us / 1e3))
