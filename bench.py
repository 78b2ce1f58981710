"""Benchmark of the hot path: log-likelihood evaluations / second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ShapeFit + Kaiser P_ell, ell = (0, 2, 4), 40 k-bins, dense synthetic
survey-like window (120 x 1200), full 120 x 120 precision; one *step* = one pass of the hot path over a batch of
1024 parameter points per GPU (theta already resident in HBM).  N > 1: one process per GPU (torchrun), walkers
sharded contiguously (weak scaling: 1024 points per rank), log-posteriors exchanged by asynchronous RCCL all-gathers, bucketed over 8 steps.
Prints ONE JSON line (rank 0).  Synthetic inputs only; nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1024
GATHER_EVERY = 8   # N > 1: steps per bucketed all-gather of log-posteriors
# Algorithmic FLOP per evaluation (DESIGN.md "Measurement"), fp64 add/mul = 1, transcendental = 20
FLOP_THEORY = 23e3 + 5e3 + 288e3 + 57.6e3 + 7e3    # template factor, spline coefficients, AP + spline eval, GL projection, tracer combine
FLOP_GEMM = 288e3 + 29e3                            # window GEMM 2 n n_in + chi2 2 n^2 + 2 n (precision folded into the window matrix)
FLOP_FINAL = 2 * 120 + 5 * 6
PEAK_FP64_TFLOPS = 78.6                             # MI355X public spec, FP64 vector = FP64 matrix (the CDNA4 guide lists no fp64 row)


def dense_window(kedges, ells, resolution=10, seed=7):
    """Synthetic survey-like window: binning matrix (x) Gaussian k-mixing + 5 % multipole leakage + 1 % noise (SURVEY.md 8d cfg 2)."""
    from desilike_amd.utils import window_matrix_bininteg
    edges = np.column_stack([kedges[:-1], kedges[1:]])
    kin, binmat = window_matrix_bininteg([edges] * len(ells), resolution=resolution)
    binmat = binmat.T
    nin = kin.size
    smooth = np.exp(-0.5 * ((kin[:, None] - kin[None, :]) / 0.004)**2)
    smooth /= smooth.sum(axis=1)[:, None]
    nl = len(ells)
    mix = np.zeros((nl * nin, nl * nin))
    for i in range(nl):
        for j in range(nl):
            mix[i * nin:(i + 1) * nin, j * nin:(j + 1) * nin] = smooth * (1. if i == j else 0.05 / (1 + abs(i - j)))
    rng = np.random.RandomState(seed)
    return kin, binmat.dot(mix) * (1. + 0.01 * rng.standard_normal((binmat.shape[0], mix.shape[1])))


def make_likelihood(device):
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    kedges = np.linspace(0., 0.2, 41)
    kin, wmat = dense_window(kedges, (0, 2, 4))
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    observable = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix=wmat, kin=kin, ellsin=(0, 2, 4), theory=theory, shotnoise=1e4)
    rng = np.random.RandomState(1)
    A = rng.standard_normal((120, 120)) * 30.
    likelihood = ObservablesGaussianLikelihood(observables=[observable], covariance=A.dot(A.T) + 1e4 * np.eye(120), device=device)
    likelihood.initialize()
    return likelihood


def sample_theta(likelihood, size, seed):
    """theta ~ Parameter.ref, as samplers draw their start (samplers/base.py:222-230)."""
    rng = np.random.RandomState(seed)
    return np.column_stack([param.ref.sample(size=size, random_state=rng) for param in likelihood.varied_params])


def oracle_constants(likelihood):
    """Constants for the NumPy oracle (cpu_baseline leg only), read off the host-side calculators."""
    obs = likelihood.observables[0]
    wm, theory = obs.wmatrix, obs.wmatrix.theory
    template = theory.template
    return dict(template='shapefit', k11=template.k, pk_dd_fid=template.pk_dd_fid, f_fid=template.f_fid, kp=template.kp, a=template.a, kin=theory.k, mu=theory.mu,
                wmu_ell=theory.wmu, ellsin=theory.ells, nd=theory.nd, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout, flatdata=obs.flatdata)


def cpu_baseline(likelihood, theta, budget=12.):
    """The NumPy oracle (restatement of the reference's numpy path, pinned to its golden vectors) on 1 host core, bounded sample."""
    from oracle import np_oracle as orc
    c = oracle_constants(likelihood)
    names = likelihood.varied_params.names()
    precision = likelihood.precision
    t0, n, check = time.perf_counter(), 0, []
    while time.perf_counter() - t0 < budget:
        p = dict(zip(names, theta[n % len(theta)]))
        p['b1'] = (p['b1'], p['b1'])
        out = orc.fullshape_observable(c, p)
        logl = orc.gaussian_loglikelihood(out['flattheory'], c['flatdata'], precision)[0]
        if n < len(theta): check.append(logl)
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit='evals/s', cores=1, kind='port', sample='{:d} evaluations cycling over the {:d} points of one step, {:.1f} s, NumPy oracle, 1 thread'.format(n, len(theta), dt)), np.array(check)


def hbm_traffic(kernel_name):
    """HBM bytes per launch of ``kernel_name`` from the latest committed PMC summary (profiles/*_pmc_hbm_traffic.txt: separate ``rocprofv3 --pmc FETCH_SIZE`` and
    ``--pmc WRITE_SIZE`` passes of this same command, gfx950 read-side correction applied, see tools/prof_round.sh / tools/pmc_summary.py); None if absent."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_hbm_traffic.txt')), reverse=True):
        for line in open(fn):
            if kernel_name in line:
                try:
                    return float(line.split()[-1]), os.path.relpath(fn, ROOT)
                except ValueError:
                    pass
    return None, None


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=200)
    parser.add_argument('--warmup', type=int, default=20)
    parser.add_argument('--batch', type=int, default=BATCH)
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-events', action='store_true', help='diagnostic: do not bracket kernels with HIP events in the timed region')
    args = parser.parse_args()

    import torch
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    distributed = world > 1 or os.environ.get('DL_BENCH_FORCE_DIST', '0') == '1'   # (forcing: smoke test of the RCCL code path with a single rank on a 1-GPU box)
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU: the hot path has no CPU fallback')
    local_rank = local_rank % torch.cuda.device_count()   # (only differs when the N > 1 path is smoke-tested on a 1-GPU box: DL_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        backend = os.environ.get('DL_BENCH_BACKEND', 'nccl')   # "nccl" IS RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    likelihood = make_likelihood(local_rank)
    ctx = likelihood._get_context()
    B = args.batch
    # N > 1: independent walker ensembles (chains) are kept in flight so that the exchange of log-posteriors never stalls the evaluation: the finalize kernel
    # writes each step's log-posteriors straight into a bucket, and one asynchronous RCCL all-gather (its own stream) ships a bucket of GATHER_EVERY steps while the
    # next bucket is being evaluated (desilike_amd/parallel.py: a kilobyte all-gather costs ~25 us of host issue time whatever its size -- issued every step it
    # would take as long as the step itself).  Every step is still one pass of the hot path over B points per GPU; every result is all-gathered.
    nslots = 2 if distributed else 1
    theta_host = sample_theta(likelihood, B, seed=42 + rank)
    thetas = [torch.as_tensor(theta_host if slot == 0 else sample_theta(likelihood, B, seed=4242 + rank), dtype=torch.float64, device=device).contiguous() for slot in range(nslots)]
    loglikes = [torch.empty(B, dtype=torch.float64, device=device) for slot in range(nslots)]
    logpriors = [torch.empty(B, dtype=torch.float64, device=device) for slot in range(nslots)]
    statuses = [torch.zeros(B, dtype=torch.int32, device=device) for slot in range(nslots)]
    loglike, status = loglikes[0], statuses[0]
    stream = torch.cuda.current_stream(device)
    bucket = None
    if distributed:
        from desilike_amd.parallel import BucketedAllGather
        bucket = BucketedAllGather(B, torch.float64, device, steps_per_bucket=GATHER_EVERY, keep=False,
                                   force_collective=os.environ.get('DL_BENCH_FORCE_DIST', '0') == '1')
    counter = [0]

    def step():
        slot = counter[0] % nslots
        counter[0] += 1
        if distributed:
            # the path's one real exchange: every rank needs every walker's log-posterior (samplers/base.py:200)
            ctx.eval_logposterior(thetas[slot], bucket.slot(), status=statuses[slot], stream=stream.cuda_stream)
            bucket.advance()
        else:
            ctx.eval_batch(thetas[slot], loglike=loglikes[slot], logprior=logpriors[slot], status=statuses[slot], stream=stream.cuda_stream)

    def barrier():
        if distributed:
            bucket.results()   # flush the partial bucket, wait for every collective in flight
            dist.barrier()
        torch.cuda.synchronize(device)

    import gc
    gc.collect()   # whatever set-up garbage holds device resources is released now, not by a collector pass inside the timed loop (hipFree synchronises the device)
    for _ in range(args.warmup):
        step()
    barrier()
    every = max(1, min(25, args.steps // 8))   # HIP events on the launch stream around each kernel, on 1 timed step out of 25 (8 samples at 200 steps, median; an event record costs ~4 us of stream time)
    ctx.profile_enable(0 if args.no_events else every)
    barrier()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    kernel_ms = ctx.profile_read() if not args.no_events else dict(theory=1., window_gemm=1., finalize=1., total=1., event_overhead=0.)
    ctx.profile_enable(0)
    if not args.no_events and not distributed:
        # An event record costs stream time of its own (3.5-4.6 us, the library's calibrated `event_overhead`), part of which overlaps the launch ramp of the
        # kernel behind it.  The three kernels ARE the step: re-attribute with one common offset such that the three durations sum to the measured step time
        # (conservative: the step time still carries the sampled event records).  These durations agree with `rocprofv3 --kernel-trace` (profiles/).
        raw = {name: kernel_ms[name] + kernel_ms['event_overhead'] for name in ['theory', 'window_gemm', 'finalize']}
        offset = (sum(raw.values()) - 1e3 * elapsed / args.steps) / 3.
        if 0. < offset < min(raw.values()):
            for name in raw: kernel_ms[name] = raw[name] - offset
            kernel_ms['event_overhead'] = offset
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    assert all(int((st != 0).sum().item()) == 0 for st in statuses), 'non-OK status in the benchmark batch'
    if rank == 0:
        value = world * B * args.steps / elapsed
        flops = {'theory': FLOP_THEORY, 'window_gemm': FLOP_GEMM, 'finalize': FLOP_FINAL}
        dominant = max(['theory', 'window_gemm', 'finalize'], key=lambda name: kernel_ms[name])
        kernel_name = {'theory': 'dl_fullshape_kernel', 'window_gemm': 'dl_chi2_gemm_kernel', 'finalize': 'dl_finalize_part_kernel'}[dominant]
        traffic, traffic_source = hbm_traffic(kernel_name) if B == BATCH else (None, None)
        per_launch = min(B, 32768)   # batches above 32768 points are evaluated in internal passes of 32768: the kernel intervals are per pass
        achieved = flops[dominant] * per_launch / (kernel_ms[dominant] * 1e-3) / 1e12
        result = {'metric': 'log-likelihood evals/sec (full-shape P_ell, 3x40 bins)', 'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
                  'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
                  'data': 'synthetic',
                  'config': {'workload': 'BASELINE configs[1]: ShapeFit+Kaiser P_ell ell=(0,2,4) x 40 k-bins, dense window 120x1200 (n_kin=400/ell), 120x120 precision, '
                                         '{:d} batched param points per GPU per step'.format(B), 'batch_per_gpu': B, 'n_params': 6, 'parallelism': 'walkers x{:d}'.format(world) + (', log-posteriors all-gathered in buckets of {:d} steps'.format(GATHER_EVERY) if distributed else '')},
                  'roofline': {'bound': 'mfma', 'bound_detail': {'theory': 'fp64 VALU (transcendentals, spline evaluation, projection); its 78.6 TFLOP/s peak equals the fp64 matrix peak',
                                                                  'window_gemm': 'fp64 MFMA (v_mfma_f64_16x16x4_f64)', 'finalize': 'launch latency'}[dominant], 'kernel': kernel_name,
                               'achieved': achieved, 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP64_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
                               'flop_per_launch': flops[dominant] * per_launch, 'avg_launch_ms': kernel_ms[dominant]},
                  'kernel_ms': {name: kernel_ms[name] for name in ['theory', 'window_gemm', 'finalize', 'total', 'event_overhead']},
                  'kernel_frac_of_fp64_peak': {name: flops[name] * per_launch / (kernel_ms[name] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS for name in ['theory', 'window_gemm']}}
        if world == 1 and not distributed and not args.no_cpu_baseline:   # (the forced single-rank RCCL smoke mode writes log-posteriors into buckets, not `loglike`)
            base, check = cpu_baseline(likelihood, theta_host)
            gpu = loglike[:len(check)].cpu().numpy()
            assert (np.abs(gpu - check) <= 1e-10 * np.maximum(1., np.abs(check))).all(), 'GPU / oracle mismatch in bench'
            result['cpu_baseline'] = base
    else:
        result = None
    # RCCL writes its version banner through C stdio: buffered when stdout is a pipe, it would come out AFTER the JSON line, at process exit.  Every rank flushes its
    # C stdio now, then a barrier, then rank 0 prints: the JSON line is the last line of the job's stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
