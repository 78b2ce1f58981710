"""Timing of the other BASELINE configurations on the GPU (not bench lines: parity-test workloads, timed for DESIGN.md):
config 3 (MLP-emulated velocileptors tables + 5 analytically marginalised parameters, B = 4096), config 4 (damped-BAO xi_ell, B = 8192),
config 5 shape (two-tracer sum, B = 256).  HIP events of the library, one call in two."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch


def time_likelihood(label, like, B, steps=40, posterior=False):
    ctx = like._get_posterior_context()[0] if posterior else like._get_context()
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=B, random_state=rng), *param.prior.limits) for param in like.varied_params])
    th = torch.as_tensor(theta, dtype=torch.float64, device='cuda').contiguous()
    out = torch.empty(B, dtype=torch.float64, device='cuda')
    st = torch.empty(B, dtype=torch.int32, device='cuda')
    import gc
    gc.collect()   # contexts of the previous likelihood are destroyed NOW (hipFree / hipEventDestroy synchronise the device: 70 ms if the collector runs inside the timed loop)
    for _ in range(5): ctx.eval_logposterior(th, out, status=st)
    torch.cuda.synchronize()
    # steady state, like bench.py's legs: the first few hundred calls of a fresh context / the first tens of milliseconds after an idle period run 1.1 - 1.5 x slower
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < float(os.environ.get('PREWARM_MS', 300.)):
        for _ in range(16): ctx.eval_logposterior(th, out, status=st)
        torch.cuda.synchronize()
    ctx.profile_enable(2)
    t0 = time.perf_counter()
    for _ in range(steps): ctx.eval_logposterior(th, out, status=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms = ctx.profile_read(); ctx.profile_enable(0)
    ok = int((st == 0).sum().item())
    print('%-58s B=%5d  %8.1f us/step  %7.2f M evals/s  kernels (dispatch-event intervals, us): theory %.1f gemm %.1f finalize %.1f  [%d ok]' % (
        label, B, 1e6 * dt, B / dt / 1e6, *(1e3 * ms[k] for k in ['theory', 'window_gemm', 'finalize']), ok))


def main():
    from test_gpu_emulator import make_mlp_likelihood
    from bench_configs import make_cfg3_full
    from bench_configs import make_cfg4
    # BASELINE configs[2] at the size SURVEY 8d states: in = 6, 4 x 64 silu, 3 * 128 * 19 outputs, n_kin = 400, W 120 x 1200, n_s = 5
    g, like, pt, theory, solved = make_cfg3_full(marg=True)
    time_likelihood('cfg3 (SURVEY 8d size): MLP tables + 5 marginalised parameters', like, 4096)
    g, like, pt, theory, solved = make_cfg3_full(marg=False)
    time_likelihood('cfg3 (SURVEY 8d size) without marginalisation', like, 4096)
    g, like, pt, theory, solved = make_mlp_likelihood(marg=True)
    time_likelihood('reduced shape of round 1 (in = 3, 3 x 64, 69 k): + 5 marginalised', like, 4096)
    for space in ['xi', 'pk']:
        g, like = make_cfg4(space)
        time_likelihood('cfg4: damped BAO ' + space, like, 8192)
    # the DESI-style BAO fit: every broadband term solved analytically
    g, like = make_cfg4('xi')
    like.initialize()
    for param in like.observables[0].wmatrix.theory.init.params.select(basename='al*'):
        param.update(derived='.marg')
    like._invalidate()
    time_likelihood('cfg4: damped BAO xi, 10 broadband terms marginalised', like, 8192)
    time_likelihood('  same, marginalised once into the precision (samplers)', like, 8192, posterior=True)


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def small_batches():
    """config 5 shape: two tracers summed, half-ensembles of 256 walkers (32 per GPU at 8 GPUs): per-call latency matters, not throughput."""
    from bench_configs import make_cfg5
    g, like = make_cfg5()
    for B in (32, 256, 1024):
        time_likelihood('cfg5: two-tracer sum (240 data points)', like, B, steps=200)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'small':
    small_batches()
