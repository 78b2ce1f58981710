"""configs[1] likelihood at several batch sizes: microseconds per step and evaluations / s (device-resident inputs, back-to-back steps).
    [DL_LIB_PATH=...] python tools/time_batches.py [B ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

like = bench.make_likelihood(0)
ctx = like._get_context()
for B in [int(v) for v in sys.argv[1:]] or [256, 1024, 4096, 16384, 32768]:
    theta = torch.as_tensor(bench.sample_theta(like, B, seed=42), dtype=torch.float64, device='cuda').contiguous()
    out = torch.empty(B, dtype=torch.float64, device='cuda')
    n = max(20, min(400, (1 << 22) // B))
    for _ in range(max(20, n // 4)): ctx.eval_logposterior(theta, out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): ctx.eval_logposterior(theta, out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print('B = %6d: %8.2f us per step, %6.2f M evals/s' % (B, 1e6 * best, B / best / 1e6))
