"""Benchmark of the hot path: log-likelihood evaluations / second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ShapeFit + Kaiser P_ell, ell = (0, 2, 4), 40 k-bins, dense synthetic
survey-like window (120 x 1200), full 120 x 120 precision; one *step* = one pass of the hot path over a batch of
1024 parameter points per GPU (theta already resident in HBM).

N > 1: one process per GPU.  Launched under ``torchrun`` (RANK / WORLD_SIZE in the environment) each process is one rank; launched plainly
(``python bench.py --gpus N``) the parent -- which never touches the GPU -- starts the N ranks itself as fresh child processes and relays rank 0's JSON line.
Walkers are sharded contiguously (weak scaling: 1024 points per rank and step); the log-posteriors are exchanged by RCCL all-gathers issued through the library's
own C ABI (``dl_comm_*``; asynchronous, bucketed over 8 steps).  The line also carries BASELINE configs[4] as written -- ONE 512-walker ensemble on two config-2
tracers, sharded over the ranks with a synchronous all-gather per half-step (strong scaling) -- under ``config5_strong``.
Prints ONE JSON line (rank 0).  Synthetic inputs only; nothing here reads /root/reference.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Kernel arguments in device memory (the HIP runtime's default on this image; stated explicitly, before anything initialises HIP, because the step is three
# latency-bound launches whose first instructions wait for their arguments: with HIP_FORCE_DEV_KERNARG=0 the same step takes 36.3 instead of 26.2 us)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

BATCH = 1024
PREWARM_LEG_MS = 300.   # untimed fixed-duration run before the timed calls of the other_configs legs (the headline's own: --prewarm-ms)
# N > 1: steps per bucketed all-gather of log-posteriors.  One collective costs the evaluation stream 10 - 20 us whatever it carries (the event pair that orders the side stream
# against it; tools/gather_cost_probe.py, tools/gather_every_sweep.sh: 8 / 16 / 32 / 64 steps per bucket = +2.9 / +1.6 / +0.9 / +0.5 us per 24.3 us step on a single-rank communicator):
# fewer, larger collectives -- 32 steps = 256 KB per rank, 0.8 ms of evaluation between two of them
GATHER_EVERY = int(os.environ.get('DL_BENCH_GATHER_EVERY', '32'))
# Algorithmic FLOP per evaluation (DESIGN.md "Measurement"), fp64 add/mul = 1, transcendental = 20
FLOP_THEORY = 23e3 + 5e3 + 288e3 + 57.6e3 + 7e3    # template factor, spline coefficients, AP + spline eval, GL projection, tracer combine
FLOP_GEMM = 288e3 + 29e3                            # window GEMM 2 n n_in + chi2 2 n^2 + 2 n (precision folded into the window matrix)
FLOP_FINAL = 2 * 120 + 5 * 6
PEAK_FP64_TFLOPS = 78.6                             # MI355X public spec, FP64 vector = FP64 matrix (the CDNA4 guide lists no fp64 row)
# The reference itself (desilike through tests/golden/refstub), timed in the BUILD container by tests/golden/make_golden.py on this workload's shape
# (vmap(likelihood) python loop, 1 process, Xeon 2.1 GHz): it cannot run on the GPU box, so the number is quoted, not measured here
REFERENCE_IN_BUILD_CONTAINER = {'value': 430., 'unit': 'evals/s', 'cores': 1, 'range': [359., 572.], 'where': 'build container (Xeon 2.1 GHz), tests/golden/make_golden.py / SURVEY.md section 6'}


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


def synthetic_covariance(n, seed=1):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal((n, n)) * 30.
    return A.dot(A.T) + 1e4 * np.eye(n)


def make_observable(tracer=None, b1=2., shotnoise=1e4, template=None, seed=7):
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    kedges = np.linspace(0., 0.2, 41)
    kin, wmat = dense_window(kedges, (0, 2, 4), seed=seed)
    if template is None: template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    kwargs = {} if tracer is None else dict(tracers=tracer)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template, **kwargs)
    data = {'b1': b1} if tracer is None else {'{}.b1'.format(tracer): b1}
    return TracerPowerSpectrumMultipolesObservable(data=data, kedges=kedges, ells=(0, 2, 4), wmatrix=wmat, kin=kin, ellsin=(0, 2, 4), theory=theory, shotnoise=shotnoise)


def make_likelihood(device):
    """BASELINE configs[1]."""
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    likelihood = ObservablesGaussianLikelihood(observables=[make_observable()], covariance=synthetic_covariance(120), device=device)
    likelihood.initialize()
    return likelihood


def make_likelihood_config5(device):
    """BASELINE configs[4]: two config-2 tracers (separate b1 / sn0 namespaces, shared ShapeFit parameters), n = 240, block-diagonal precision."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    from scipy import linalg
    template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    observables = [make_observable('LRG', 2., 1e4, template, seed=7), make_observable('ELG', 1.3, 4e3, template, seed=8)]
    covariance = linalg.block_diag(synthetic_covariance(120, seed=1), synthetic_covariance(120, seed=2))
    likelihood = ObservablesGaussianLikelihood(observables=observables, covariance=covariance, device=device)
    likelihood.initialize()
    return likelihood


def sample_theta(likelihood, size, seed):
    """theta ~ Parameter.ref, as samplers draw their start (samplers/base.py:222-230)."""
    rng = np.random.RandomState(seed)
    return np.column_stack([param.ref.sample(size=size, random_state=rng) for param in likelihood.varied_params])


def oracle_constants(likelihood, iobs=0):
    """Constants for the NumPy oracle (cpu_baseline leg and the post-hoc checks only), read off the host-side calculators."""
    obs = likelihood.observables[iobs]
    wm, theory = obs.wmatrix, obs.wmatrix.theory
    template = theory.template
    return dict(template='shapefit', k11=template.k, pk_dd_fid=template.pk_dd_fid, f_fid=template.f_fid, kp=template.kp, a=template.a, kin=theory.k, mu=theory.mu,
                wmu_ell=theory.wmu, ellsin=theory.ells, nd=theory.nd, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout, flatdata=obs.flatdata)


def oracle_logposterior(likelihood, theta):
    """Post-hoc checker (never timed, never on the product path): log-posterior of the rows of ``theta`` by the NumPy oracle -- every observable through
    ``fullshape_observable`` (per-tracer b1 / sn0 namespaces, shared template parameters), joint Gaussian chi2, priors of the varied parameters."""
    from oracle import np_oracle as orc
    names = likelihood.varied_params.names()
    consts = [oracle_constants(likelihood, iobs) for iobs in range(len(likelihood.observables))]
    bias_names = [obs.wmatrix.theory._bias_names() for obs in likelihood.observables]
    flatdata = np.concatenate(likelihood._flatdata_list())
    priors = []
    for param in likelihood.varied_params:
        prior = param.prior
        priors.append(dict(dist=prior.dist, limits=tuple(prior.limits), loc=getattr(prior, 'loc', 0.), scale=getattr(prior, 'scale', 1.)))
    out = np.empty(len(theta))
    for i, row in enumerate(theta):
        p = dict(zip(names, row))
        theory = []
        for c, bias in zip(consts, bias_names):
            q = {name: p[name] for name in ['qpar', 'qper', 'dm', 'df'] if name in p}
            q['b1'] = (p[bias['b1X']], p[bias['b1Y']])
            q['sn0'] = p[bias['sn0']]
            theory.append(orc.fullshape_observable(c, q)['flattheory'])
        out[i] = orc.gaussian_loglikelihood(np.concatenate(theory), flatdata, likelihood.precision)[0]
    return out + orc.logprior(np.asarray(theta), priors)


def _oracle_loop(payload):
    """One CPU worker of the baseline: the NumPy oracle cycling over the points of one step for ``budget`` seconds; returns (evaluations, seconds, first log-likelihoods)."""
    c, names, precision, theta, budget, ncheck = payload
    sys.path.insert(0, ROOT)
    from oracle import np_oracle as orc
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)   # one BLAS thread per worker: the reported core count is the number of workers
    except ImportError:
        limiter = contextlib.nullcontext()
    with limiter:
        t0, n, check = time.perf_counter(), 0, []
        while time.perf_counter() - t0 < budget:
            p = dict(zip(names, theta[n % len(theta)]))
            p['b1'] = (p['b1'], p['b1'])
            out = orc.fullshape_observable(c, p)
            logl = orc.gaussian_loglikelihood(out['flattheory'], c['flatdata'], precision)[0]
            if n < ncheck: check.append(logl)
            n += 1
        dt = time.perf_counter() - t0
    return n, dt, np.array(check)


def cpu_baseline(likelihood, theta, budget=10.):
    """The NumPy oracle (restatement of the reference's numpy path, pinned to its golden vectors) on the host cores, bounded samples: 1 process / 1 thread, then
    N = nproc independent processes each cycling over the step's points (the reference's MPI data parallelism, desilike/base.py:310-316)."""
    c = oracle_constants(likelihood)
    names = likelihood.varied_params.names()
    n, dt, check = _oracle_loop((c, names, likelihood.precision, theta, budget, len(theta)))
    base = dict(value=n / dt, unit='evals/s', cores=1, kind='port',
                sample='{:d} evaluations cycling over the {:d} points of one step, {:.1f} s, NumPy oracle, 1 process, 1 thread'.format(n, len(theta), dt),
                reference_in_build_container=REFERENCE_IN_BUILD_CONTAINER)
    ncores = os.cpu_count() or 1
    try:
        import multiprocessing as mp
        saved = {key: os.environ.get(key) for key in ['OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS']}
        for key in saved: os.environ[key] = '1'
        try:
            with mp.get_context('spawn').Pool(ncores) as pool:   # fresh interpreters: nothing of this process's GPU state is inherited
                t0 = time.perf_counter()
                results = pool.map(_oracle_loop, [(c, names, likelihood.precision, theta, budget, 0)] * ncores)
                wall = time.perf_counter() - t0
        finally:
            for key, value in saved.items():
                if value is None: os.environ.pop(key, None)
                else: os.environ[key] = value
        total = sum(r[0] for r in results)
        span = max(r[1] for r in results)
        base['multi'] = dict(value=total / span, unit='evals/s', cores=ncores, kind='port',
                             sample='{:d} independent processes x {:.1f} s, {:d} evaluations in total ({:.1f} s wall incl. start-up)'.format(ncores, span, total, wall))
    except Exception as exc:   # the 1-core number stands on its own
        base['multi'] = dict(error=repr(exc))
    return base, check


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources of the tree (desilike_amd/csrc/*.h *.hpp *.hip, sorted by name): ``tools/prof_round.sh`` writes it next to the
    summaries it takes (profiles/<tag>_source_hash.txt), and figures copied from a committed profile enter the bench line only when that hash is the running tree's."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, 'desilike_amd', 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'desilike_amd', 'csrc', '*.hpp')) + glob.glob(os.path.join(ROOT, 'desilike_amd', 'csrc', '*.hip'))):
        h.update(os.path.basename(fn).encode()); h.update(open(fn, 'rb').read())
    return h.hexdigest()[:16]


def profile_is_current(fn):
    """True when the committed profile ``fn`` (profiles/<tag>_...) was taken on the kernel sources of this tree (profiles/<tag>_source_hash.txt)."""
    tag = os.path.basename(fn).split('_')[0]
    hfn = os.path.join(ROOT, 'profiles', tag + '_source_hash.txt')
    return os.path.isfile(hfn) and open(hfn).read().split()[0] == source_hash()


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


def trace_average(kernel_name):
    """Average duration (ms) of ``kernel_name`` in the latest committed ``rocprofv3 --kernel-trace --stats`` summary of this same command (profiles/*_kernel_stats.csv, written by
    tools/prof_round.sh WITHOUT the host-array leg: every launch in it is a 1024-point launch of the timed step), with the file it came from; (None, None) if absent.  The variant
    with the most calls is taken when a kernel template has several instantiations in the file."""
    import csv
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9][a-z]_kernel_stats.csv')), reverse=True):
        best = None
        with open(fn) as f:
            for row in csv.DictReader(f):
                if kernel_name in row.get('Name', ''):
                    calls = int(row['Calls'])
                    if best is None or calls > best[0]: best = (calls, float(row['AverageNs']) * 1e-6)
        if best is not None: return best[1], os.path.relpath(fn, ROOT)
    return None, None


def compact_line(result):
    """The bench line reduced to what a reader checks: the contract's headline fields, `roofline` and `cpu_baseline` of the headline, and per leg
    {id, value, ms_per_step, frac of the fp64 peak, kernel microseconds, error against the oracle / the reference}."""
    def num(x, digits=4):
        return None if x is None else float('{:.{}g}'.format(float(x), digits))

    out = {key: result[key] for key in ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'] if key in result}
    out['value'], out['ms_per_step'] = num(out['value'], 6), num(out['ms_per_step'], 6)
    cfg = result.get('config', {})
    out['config'] = {'workload': 'BASELINE configs[1]: ShapeFit+Kaiser P_ell (0,2,4) x 40 k-bins, dense window 120x1200, {:d} batched points per GPU per step'.format(cfg.get('batch_per_gpu', 0)),
                     'batch_per_gpu': cfg.get('batch_per_gpu'), 'parallelism': cfg.get('parallelism'), 'collective': cfg.get('collective'), 'ranks': cfg.get('ranks')}
    r = result.get('roofline', {})
    out['roofline'] = {'bound': r.get('bound'), 'kernel': r.get('kernel'), 'achieved': num(r.get('achieved')), 'peak': r.get('peak'), 'unit': r.get('unit'), 'frac': num(r.get('frac')),
                       'traffic': r.get('traffic'), 'traffic_from_committed_profile': r.get('traffic_source'), 'avg_launch_us': num(1e3 * r['avg_launch_ms']) if r.get('avg_launch_ms') else None,
                       'event_samples': r.get('event_samples'), 'trace_avg_launch_us': num(1e3 * r['trace_avg_launch_ms']) if r.get('trace_avg_launch_ms') else None, 'trace_frac': num(r.get('trace_frac')),
                       'traffic_step_over_algorithmic': num(r.get('traffic_step_over_algorithmic'))}
    out['kernel_us'] = {name: num(1e3 * v) for name, v in (result.get('kernel_ms') or {}).items() if v}
    if result.get('cpu_baseline'):
        b = result['cpu_baseline']
        out['cpu_baseline'] = {key: b.get(key) for key in ['value', 'unit', 'cores', 'kind', 'sample']}
        out['cpu_baseline']['value'] = num(b.get('value'))
        if isinstance(b.get('multi'), dict) and 'value' in b['multi']: out['cpu_baseline']['all_threads'] = {'value': num(b['multi']['value']), 'cores': b['multi'].get('cores')}
    if result.get('sustained'): out['sustained'] = {'value': num(result['sustained'].get('value'), 5), 'seconds': num(result['sustained'].get('seconds', result['sustained'].get('elapsed_s')), 3)}
    if isinstance(result.get('streams'), dict): out['streams'] = {key: (num(v.get('value'), 5) if isinstance(v, dict) else num(v, 5) if isinstance(v, (int, float)) else v) for key, v in result['streams'].items() if key != 'workload'}
    if isinstance(result.get('streams'), list): out['streams'] = [{'streams': leg.get('streams'), 'value': num(leg.get('value'), 5)} for leg in result['streams'] if isinstance(leg, dict)]
    hc = result.get('host_call')
    if isinstance(hc, dict):
        if 'error' in hc: out['host_call'] = {'error': hc['error']}
        else:
            out['host_call'] = {'256': {k: num(v) for k, v in (hc.get('per_batch_size', {}).get('256') or {}).items()}}
            hd = hc.get('host_driven_ensemble')
            if isinstance(hd, dict): out['host_call']['host_driven_ensemble'] = {k: num(v) for k, v in hd.items() if isinstance(v, (int, float)) and k in ('us_per_update', 'value')}
    legs = []
    for i, leg in enumerate(result.get('other_configs') or []):
        if not isinstance(leg, dict): continue
        rr = leg.get('roofline', {}) if isinstance(leg.get('roofline'), dict) else {}
        w = str(leg.get('workload', ''))
        ident = ('configs[2]-stacked-layout' if 'layout the reference ships' in w else 'configs[2]-single-network' if 'configs[2]' in w else 'configs[3]-bao-xi' if 'configs[3]' in w else
                 'tns-one-loop' if 'TNS' in w else 'other_configs[{:d}]'.format(i))
        item = {'id': ident, 'value': num(leg.get('value'), 5), 'unit': leg.get('unit'), 'ms_per_step': num(leg.get('ms_per_step'), 5), 'batch': leg.get('batch'),
                'frac': num(rr.get('frac')), 'kernel_us': num(1e3 * rr['avg_launch_ms']) if rr.get('avg_launch_ms') else None,
                'oracle_err': num((leg.get('oracle_check') or {}).get('max_rel_err_vs_oracle'), 3)}
        if leg.get('reference_check'): item['reference_err'] = num(leg['reference_check'].get('max_rel_err_vs_reference'), 3)
        if 'error' in leg: item['error'] = leg['error']
        legs.append(item)
    if legs: out['other_configs'] = legs
    cw = result.get('chains_weak')
    if isinstance(cw, dict): out['chains_weak'] = {'error': cw['error']} if 'error' in cw else {'value': num(cw.get('value'), 5), 'unit': cw.get('unit'), 'n_gpus': cw.get('n_gpus'), 'scaling': cw.get('scaling'), 'per_k': [{'chains_per_gpu': c.get('chains_per_gpu'), 'chains': c.get('chains'), 'value': num(c.get('value'), 5), 'us_per_update_per_chain': num(c.get('us_per_update_per_chain'))} for c in cw.get('per_k', []) if isinstance(c, dict)]}
    mh = result.get('mh_chains')
    if isinstance(mh, dict): out['mh_chains'] = {'value': num(mh.get('value'), 5), 'unit': mh.get('unit')} if 'error' not in mh else {'error': mh['error']}
    st = result.get('config5_strong')
    if isinstance(st, dict): out['config5_strong'] = {key: (num(st[key], 5) if isinstance(st.get(key), float) else st.get(key)) for key in ['value', 'unit', 'us_per_update', 'n_gpus', 'sharded'] if key in st}
    if result.get('gathered_check') is not None: out['gathered_check'] = result['gathered_check']
    out['record'] = 'compact (the complete record is the line before this one)'
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(ngpus):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as fresh child processes (this parent has made no GPU call and makes none), relay rank 0's
    stdout (the JSON line), send the other ranks' output to stderr; the first failure ends the job."""
    port = free_port()
    procs = []
    for rank in range(ngpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if rank == 0 else sys.stderr))
    code = 0
    pending = list(procs)
    while pending:
        for proc in list(pending):
            rc = proc.poll()
            if rc is None: continue
            pending.remove(proc)
            if rc != 0 and code == 0:
                code = rc
                for other in pending: other.terminate()   # exactly the children started above
        time.sleep(0.05)
    return code


def timed_steps(step, barrier, steps):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def init_rccl_group(parallel, local_rank, rank, world, timeout=180.):
    """``parallel.RcclGroup`` with a deadline and a self-check.  ``ncclCommInitRank`` blocks until every rank has arrived: a rank that died (or a wrong WORLD_SIZE) would
    hang the job silently -- the communicator is created on a helper thread and a rank that waits longer than ``timeout`` seconds raises.  Then every rank all-gathers its
    own rank number and checks the result: the first collective of the run either works, visibly, or fails here."""
    import threading
    import torch
    box = {}

    def create():
        try:
            box['group'] = parallel.RcclGroup(local_rank, rank=rank, world=world)
        except BaseException as exc:   # noqa: BLE001 (reported by the caller)
            box['error'] = exc

    thread = threading.Thread(target=create, daemon=True)
    thread.start()
    thread.join(timeout)
    if thread.is_alive():
        raise TimeoutError('communicator not created after {:.0f} s (DL_COMM_TIMEOUT): is every one of the {:d} ranks alive and on its own GPU?'.format(timeout, world))
    if 'error' in box:
        raise box['error']
    group = box['group']
    device = torch.device('cuda', local_rank)
    send = torch.full((4,), float(rank), dtype=torch.float64, device=device)
    recv = torch.full((4 * world,), -1., dtype=torch.float64, device=device)
    group.allgather_into(recv, send)
    torch.cuda.synchronize(device)
    expected = torch.arange(world, dtype=torch.float64, device=device).repeat_interleave(4)
    if not torch.equal(recv, expected):
        raise RuntimeError('RCCL all-gather self-check failed on rank {:d}: got {}'.format(rank, recv.cpu().tolist()))
    return group


def config5_strong(group, device, local_rank, rank, world, iterations, warmup=300):
    """BASELINE configs[4] as written: ONE ensemble of 512 walkers on the two-tracer likelihood; every half-step's 256 proposals are split over the ranks
    (min_shard_rows = 0) and the log-posteriors all-gathered synchronously (in place, on the evaluation stream) before the accept step: strong scaling."""
    import torch
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding, RcclGroup
    likelihood = make_likelihood_config5(local_rank)
    sharding = WalkerSharding(group=group, min_shard_rows=0)
    sampler = EmceeSampler(likelihood, nwalkers=512, seed=42, sharding=sharding, device_resident=True)
    start, logposterior = sampler._get_start(512)
    ens = sampler._get_ensemble()
    ens.set_state(start, logposterior)
    nparams = ens.n_params
    chain = torch.empty((iterations, 512, nparams), dtype=torch.float64, device=device)
    chain_logp = torch.empty((iterations, 512), dtype=torch.float64, device=device)

    def barrier():
        torch.cuda.synchronize(device)
        if group is not None: group.barrier()
        torch.cuda.synchronize(device)

    ens.run(warmup)
    barrier()
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < PREWARM_LEG_MS:   # (steady state, like the other legs; every rank runs the same number of collectives: time-based only for one rank)
        ens.run(warmup)
        barrier()
        if world > 1: break
    t0 = time.perf_counter()
    ens.run(iterations, chain=chain, chain_logp=chain_logp)
    barrier()
    elapsed = time.perf_counter() - t0
    if group is not None: elapsed = group.max(elapsed)
    coords, logp, nacc = ens.get_state()
    sharded = isinstance(group, RcclGroup) and (world > 1 or os.environ.get('DL_ENS_FORCE_COMM', None) is not None)
    assert np.isfinite(logp).all() and np.isfinite(chain_logp.cpu().numpy()).all()
    checked = None
    if rank == 0:
        # the final log-posteriors of ALL 512 walkers against the NumPy oracle (pinned on the reference's outputs at this very shape: tests/golden/cfg5_bench.npz)
        ref = oracle_logposterior(likelihood, coords)
        err = np.abs(logp - ref) / np.maximum(1., np.abs(ref))
        assert (err <= 1e-10).all(), 'GPU / oracle mismatch on the config-5 ensemble: {:.3e}'.format(err.max())
        checked = {'points': int(len(ref)), 'max_rel_err_vs_oracle': float(err.max()), 'tolerance': 1e-10}
    return {'workload': 'BASELINE configs[4]: EnsembleSampler (stretch move), 512 walkers x two config-2 tracers (n = 240), {:d} ensemble updates, device-resident'.format(iterations),
            'value': 512 * iterations / elapsed, 'unit': 'evals/s', 'scaling': 'strong', 'n_gpus': world, 'us_per_update': 1e6 * elapsed / iterations,
            'rows_per_gpu_per_half_step': ens.info('rows_per_rank'), 'sharded': sharded,
            'exchange': 'one in-place ncclAllGather of 256 log-posteriors per half-step on the evaluation stream' if sharded else 'none (single rank, or a host-side group: every rank evaluates all walkers)',
            'acceptance_fraction': float(nacc.sum()) / (512. * ens.info('iteration')), 'n_params': nparams, 'oracle_check': checked}


def _timed_context(ctx, theta, steps, warmup, posterior_out, status, device):
    """Seconds per call of ``ctx.eval_logposterior`` on the resident batch ``theta`` + the library's dispatch-attached kernel intervals (median, ms) of sampled calls."""
    import gc
    import torch
    # whatever the earlier legs left behind (contexts of other streams, replicas of the host-array leg) is destroyed NOW: a collector pass inside the timed loop that frees
    # device resources synchronises the device (hipFree / hipEventDestroy: ~70 ms, seen as 2 M instead of 80 M evals/s on whichever leg it hit)
    gc.collect()
    # warm-up like the headline's: a fixed duration (PREWARM_LEG_MS) on top of `warmup` calls -- the first few hundred calls of a fresh context and the first tens of
    # milliseconds after an idle period (the oracle checks between the legs run on the CPU) are 1.1 - 1.5 x slower than the steady state the figures are meant to describe
    for _ in range(warmup): ctx.eval_logposterior(theta, posterior_out, status=status)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < PREWARM_LEG_MS:
        for _ in range(16): ctx.eval_logposterior(theta, posterior_out, status=status)
        torch.cuda.synchronize(device)
    every = max(1, steps // 8)
    ctx.profile_enable(every)
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(steps): ctx.eval_logposterior(theta, posterior_out, status=status)
        torch.cuda.synchronize(device)
        elapsed = (time.perf_counter() - t0) / steps
    finally:
        gc.enable()
    kernel_ms = ctx.profile_read()
    ctx.profile_enable(0)
    return elapsed, kernel_ms


def cfg3_flop_counts(pt, n, n_solved, n_mono=19):
    """The BUILD's own algorithmic count for BASELINE configs[2] (emulated theories): the last MLP layer x k-interpolation x window x L^T are
    folded into ONE operator at context creation, so a point costs the hidden layers of the three engines, the product of that operator with the basis
    (n_mono x n columns x n_basis), the monomial contraction of the 1 + n_solved rows, and the Gram matrix of those rows (SURVEY 8d's 1.8 MFLOP counts the
    unfolded last layer 64 x 7296 and the 120 x 1200 window separately; against it the fused kernel would run above the peak)."""
    hidden = pt.engines['pktable'].layers[:-1]
    forward = sum(2 * k.shape[0] * k.shape[1] + 25 * k.shape[1] for k, b in hidden)                   # MACs + one silu (exp, division ~ 25) per unit
    forward += sum(2 * k.shape[0] * k.shape[1] + 25 * k.shape[1] for name in ['sigma8', 'fsigma8'] for k, b in pt.engines[name].layers)
    n_basis = hidden[-1][0].shape[1] + 1
    # monomial rows: the residual row contracts all n_mono monomials; the derivative row of a solved alpha* / sn* has at most two non-zero monomials (round 4: the
    # kernel multiplies only those, so the dense count 2 n_mono n (1 + n_solved) of round 3 would credit work that is no longer done)
    return {'forward': forward, 'folded_operator': 2 * n_mono * n * n_basis, 'monomial_rows': 2 * n * (n_mono + 2 * n_solved), 'gram': (1 + n_solved) * (2 + n_solved) * n,
            'solve': 2 * n_solved**3 // 3 + 4 * n_solved**2}


def host_call_leg(likelihood, sizes=(1, 16, 256, 1024), ncalls=500):
    """The host-array entry point (``dl_eval_logposterior_host``: what the reference-side binding and an unmodified desilike sampler call, samplers/base.py:144-200,
    samplers/emcee.py:69): NumPy arrays in, NumPy arrays out, one call per batch -- PCIe-inclusive by construction.  Median / p99 / mean time per call (host clock around
    the ctypes call) of ``ncalls`` calls at each batch size, after 50 untimed ones; the results are checked against the device-resident path bit for bit."""
    import torch
    ctx = likelihood._get_posterior_context()[0]
    theta_all = np.ascontiguousarray(sample_theta(likelihood, max(sizes), seed=77))
    device = torch.device('cuda', ctx.device)
    ref = torch.empty(max(sizes), dtype=torch.float64, device=device)
    ctx.eval_logposterior(torch.as_tensor(theta_all, dtype=torch.float64, device=device).contiguous(), ref)
    torch.cuda.synchronize(device)
    ref = ref.cpu().numpy()
    out = {}
    for B in sizes:
        theta = np.ascontiguousarray(theta_all[:B])
        for _ in range(50): got = ctx.eval_logposterior_host(theta)[0]
        assert np.array_equal(got, ref[:B], equal_nan=True), 'host-array entry point differs from the device-resident path'
        t = np.empty(ncalls)
        for i in range(ncalls):
            t0 = time.perf_counter_ns()
            ctx.eval_logposterior_host(theta)
            t[i] = 1e-3 * (time.perf_counter_ns() - t0)
        out[str(B)] = {'median_us': float(np.median(t)), 'p99_us': float(np.percentile(t, 99)), 'mean_us': float(t.mean()), 'evals_per_s': float(B / (1e-6 * t.mean()))}
    return {'entry_point': 'dl_eval_logposterior_host (host pointers in / out; pinned, device-mapped staging read and written by the kernels themselves; completion flag)',
            'calls_per_size': ncalls, 'includes': 'ctypes call, staging copy, PCIe reads / writes of the kernels, completion wait', 'per_batch_size': out,
            'host_driven_ensemble': host_driven_ensemble()}


def host_driven_ensemble(iterations=200):
    """What an UNMODIFIED host-driven ensemble sampler pays end to end (emcee's loop in the reference, samplers/emcee.py:69: proposals in NumPy on the host, the log-posteriors of
    each half of the walkers through ONE host-array call): BASELINE configs[4]'s 512 walkers x two tracers with ``EmceeSampler(device_resident=False)`` -- two
    ``dl_eval_logposterior_host`` calls of 256 points plus the NumPy stretch move per update.  The device-resident ensemble of the same likelihood is the `config5_strong` leg."""
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    like = make_likelihood_config5(0)
    sampler = EmceeSampler(like, nwalkers=512, seed=42, sharding=WalkerSharding(group=False), device_resident=False, use_emcee=False)
    sampler.run(niterations=20)
    t0 = time.perf_counter()
    chain = sampler.run(niterations=iterations)
    elapsed = time.perf_counter() - t0
    assert np.isfinite(chain['logposterior'][-1]).all()
    return {'workload': 'EmceeSampler(device_resident=False): host-side stretch move, two 256-point host-array calls per update, 512 walkers x two config-2 tracers', 'iterations': iterations,
            'us_per_update': 1e6 * elapsed / iterations, 'value': 512 * iterations / elapsed, 'unit': 'evals/s'}


def other_configs(device, steps=40, warmup=5, ncheck=8):
    """BASELINE configs[2] (MLP-emulated tables + 5 analytically marginalised parameters, 4096 points) and configs[3] (damped-BAO xi_ell through the Hankel operator,
    8192 points) on this GPU: evaluations / s over ``steps`` calls on a resident batch, the dominant kernel's roofline fraction with the build's own FLOP count, and a
    post-hoc check of ``ncheck`` points against the NumPy oracle (the parity tests proper: tests/test_gpu_emulator.py, tests/test_gpu_bao.py)."""
    import gc
    import torch
    from bench_configs import make_cfg3_full, cfg3_oracle_solution, make_cfg4, bao_point
    from oracle import np_oracle as orc
    out = []

    def sample(like, B, seed):
        rng = np.random.RandomState(seed)
        return np.column_stack([np.clip(param.ref.sample(size=B, random_state=rng), *param.prior.limits) for param in like.varied_params])

    # ---- configs[2] ----
    g, like, pt, theory, solved = make_cfg3_full(marg=True)
    B = 4096
    ctx = like._get_context()
    theta_host = sample(like, B, 3)
    theta = torch.as_tensor(theta_host, dtype=torch.float64, device=device).contiguous()
    post, status = torch.empty(B, dtype=torch.float64, device=device), torch.zeros(B, dtype=torch.int32, device=device)
    gc.collect()
    elapsed, kernel_ms = _timed_context(ctx, theta, steps, warmup, post, status, device)
    assert int((status != 0).sum().item()) == 0
    loglike = ctx.eval_batch_host(theta_host[:ncheck])[0]
    err = max(abs(loglike[i] - cfg3_oracle_solution(like, pt, theory, solved, theta_host[i])['loglikelihood']) / max(1., abs(loglike[i])) for i in range(ncheck))
    assert err <= 1e-10, 'GPU / oracle mismatch on configs[2]: {:.3e}'.format(err)
    flops = cfg3_flop_counts(pt, n=like.flatdata.size, n_solved=len(solved))
    fused = flops['forward'] + flops['folded_operator'] + flops['monomial_rows'] + flops['gram']
    slot = max(['theory', 'window_gemm'], key=lambda name: kernel_ms[name])
    achieved = fused * B / (kernel_ms[slot] * 1e-3) / 1e12
    out.append({'workload': 'BASELINE configs[2]: MLP-emulated velocileptors-style tables (in 6 -> 4 x 64 silu -> 3 x 128 x 19) + 19-monomial combination + cubic interpolation to n_kin = 400 + window 120 x 1200 '
                            '+ 5 analytically marginalised parameters, {:d} batched points'.format(B),
                'value': B / elapsed, 'unit': 'evals/s', 'ms_per_step': 1e3 * elapsed, 'steps': steps, 'dtype': 'f64', 'batch': B,
                'roofline': {'bound': 'mfma', 'kernel': 'dl_emulated_feature_gram_kernel (MLP forward + folded-operator product + Gram epilogue, one launch)', 'flop_per_eval': flops,
                             'flop_per_launch': fused * B, 'avg_launch_ms': kernel_ms[slot], 'achieved': achieved, 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP64_TFLOPS,
                             'flop_count': "the build's own algorithm (last layer x interpolation x window x L^T folded at create): DESIGN.md section 4; SURVEY 8d's unfolded 1.8 MFLOP / eval would exceed the peak"},
                'kernel_ms': {name: kernel_ms[name] for name in ['theory', 'window_gemm', 'finalize']},
                'oracle_check': {'points': ncheck, 'max_rel_err_vs_oracle': float(err), 'tolerance': 1e-10}})
    del ctx, like, pt, theory
    gc.collect()

    # ---- configs[2] on the emulator layout the reference ships (emulators/conversion.py:44-98): stacked engines, amplitude rescale, redshift blend ----
    try:
        out.append(_stacked_config(device, steps, warmup, ncheck, sample))
    except Exception as exc:   # (reported, not fatal)
        import traceback
        traceback.print_exc(file=sys.stderr)
        out.append({'workload': 'BASELINE configs[2], jaxeffort emulator layout', 'error': repr(exc)})
    gc.collect()

    # ---- configs[3] ----
    g, like = make_cfg4('xi')
    B = 8192
    ctx = like._get_context()
    names = like.varied_params.names()
    theta_host = sample(like, B, 77)
    theta = torch.as_tensor(theta_host, dtype=torch.float64, device=device).contiguous()
    post, status = torch.empty(B, dtype=torch.float64, device=device), torch.zeros(B, dtype=torch.int32, device=device)
    gc.collect()
    elapsed, kernel_ms = _timed_context(ctx, theta, steps, warmup, post, status, device)
    assert int((status != 0).sum().item()) == 0
    loglike = ctx.eval_batch_host(theta_host[:ncheck])[0]
    c = g['obs0']
    gfix = dict(g); gfix['names'] = np.array(names)
    err = 0.
    for i in range(ncheck):
        power, broadband = bao_point(gfix, theta_host[i])
        ref = orc.gaussian_loglikelihood(np.ravel(orc.get_corr(power, c['kin'], c['s'], (0, 2)) + broadband), c['flatdata'], like.precision)[0]
        err = max(err, abs(loglike[i] - ref) / max(1., abs(ref)))
    assert err <= 1e-10, 'GPU / oracle mismatch on configs[3]: {:.3e}'.format(err)
    nkin, nmu, n = len(c['kin']), len(c['mu']), like.flatdata.size
    nbb = len(c['broadband_params'])
    flops = {'bao_theory': 150 * nkin * nmu, 'hankel_window_gemm': 2 * n * (2 * nkin + nbb), 'chi2': 2 * n * n + 2 * n}
    achieved = flops['bao_theory'] * B / (kernel_ms['theory'] * 1e-3) / 1e12
    out.append({'workload': 'BASELINE configs[3]: damped-BAO xi_ell (ell = 0, 2; 30 s-bins; 300 log-k x 10 mu; FFTLog Hankel transform folded into a constant operator) + Gaussian likelihood, '
                            '{:d} batched points'.format(B),
                'value': B / elapsed, 'unit': 'evals/s', 'ms_per_step': 1e3 * elapsed, 'steps': steps, 'dtype': 'f64', 'batch': B,
                'roofline': {'bound': 'valu', 'kernel': 'dl_bao_kernel (P(k, mu) on the 300-point log grid, no MFMA)', 'flop_per_eval': flops, 'flop_per_launch': flops['bao_theory'] * B,
                             'avg_launch_ms': kernel_ms['theory'], 'achieved': achieved, 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP64_TFLOPS,
                             'flop_count': "SURVEY 8d's 150 FLOP per (k, mu) evaluation for the theory kernel; the FFTLog of SURVEY's count (0.49 MFLOP / eval) does not run per step: get_corr is linear in "
                                           "P_ell, the build applies it as a 60 x 610 operator inside the window GEMM"},
                'kernel_ms': {name: kernel_ms[name] for name in ['theory', 'window_gemm', 'finalize']},
                'oracle_check': {'points': ncheck, 'max_rel_err_vs_oracle': float(err), 'tolerance': 1e-10}})
    del ctx, like
    gc.collect()

    # ---- TNS one-loop theory: the reference's own perturbation-theory producer (full_shape.py:688-971), the step left of the path ----
    try:
        out.append(_tns_config(device, steps, ncheck, orc))
    except Exception as exc:   # (reported, not fatal: the lines of configs 2 and 3 above stay)
        import traceback
        traceback.print_exc(file=sys.stderr)
        out.append({'workload': 'TNS one-loop theory', 'error': repr(exc)})
    return out


def _stacked_config(device, steps, warmup, ncheck, sample):
    import torch
    from bench_configs import make_cfg3_stacked, cfg3_stacked_oracle_solution
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=True)
    B = 4096
    ctx = like._get_context()
    theta_host = sample(like, B, 3)
    theta = torch.as_tensor(theta_host, dtype=torch.float64, device=device).contiguous()
    post, status = torch.empty(B, dtype=torch.float64, device=device), torch.zeros(B, dtype=torch.int32, device=device)
    elapsed, kernel_ms = _timed_context(ctx, theta, steps, max(warmup, 200), post, status, device)   # (the first ~200 calls of a fresh context run 1.5x slower: tools/time_gap_stacked.py)
    assert int((status != 0).sum().item()) == 0
    loglike = ctx.eval_batch_host(theta_host[:ncheck])[0]
    err = max(abs(loglike[i] - cfg3_stacked_oracle_solution(like, pt, theory, solved, theta_host[i])['loglikelihood']) / max(1., abs(loglike[i])) for i in range(ncheck))
    assert err <= 1e-10, 'GPU / oracle mismatch on configs[2] (stacked layout): {:.3e}'.format(err)
    # ... and against the REFERENCE ITSELF at this shape (tests/golden/boundary_cfg3_stacked_bench.npz: the exact quadratic of the reference's own log-posterior in the solved
    # parameters at 12 points, through its Emulator.from_state + REPT tracer on the same synthetic weights; tests/golden/make_boundary_fixture.py): the context the timed loop ran on
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'boundary_cfg3_stacked_bench.npz'))
    names = like.varied_params.names()
    assert sorted(names) == sorted(str(n) for n in g['names'])
    rows = g['theta'][:len(g['marg_c'])][:, [[str(n) for n in g['names']].index(name) for name in names]]
    ll, lp, st = ctx.eval_batch_host(rows)[:3]
    marg = np.asarray(g['cfg/marg.kind']).astype(bool)
    ref_err = 0.
    for i in np.flatnonzero(st == 0):
        c, grad, H = g['marg_c'][i], g['marg_g'][i], g['marg_H'][i]
        ref = c - 0.5 * grad.dot(np.linalg.solve(H, grad)) - 0.5 * np.linalg.slogdet(-H[np.ix_(marg, marg)])[1]
        ref_err = max(ref_err, abs(ll[i] + lp[i] - ref) / max(1., abs(ref)))
    assert (st == 0).sum() >= len(rows) - 1 and ref_err <= 1e-10, 'GPU / reference mismatch on configs[2] (stacked layout): {:.3e}'.format(ref_err)
    spec = like._spec({}, like._flatdata_list(), like.precision)['observables'][0]
    groups, widths = np.asarray(spec['emu0']['groups']), [int(w) for w in np.ravel(spec['emu0']['widths'])]
    n, H = like.flatdata.size, widths[-1]
    per_network = sum(2 * a * b + 25 * b for a, b in zip(widths[:-1], widths[1:]))                       # MACs + one activation (~25) per unit
    flops = {'networks': int(groups[:, 1].max()) * per_network, 'folded_operator': int(sum(2 * n * ((te - tb) * H + 1) * (m1 - m0) for tb, te, m0, m1 in groups)),
             'monomial_rows': 2 * n * (19 + 2 * len(solved)), 'gram': (1 + len(solved)) * (2 + len(solved)) * n}
    fused = sum(flops.values())
    # two launches per step (csrc/dl_emu_stacked_split.h): the network chains carry the events of the theory phase, the feature GEMMs (+ Gram matrices and the solve in the tail) those of
    # the GEMM phase; the fraction is priced on the SUM of the two intervals (DL_NO_STK_SPLIT=1: one launch, in the GEMM phase alone)
    launch_ms = kernel_ms['theory'] + kernel_ms['window_gemm']
    achieved = fused * B / (launch_ms * 1e-3) / 1e12
    return {'workload': "BASELINE configs[2] on the emulator layout the reference ships (emulators/conversion.py:44-98): engines '11' / 'loop' / 'ct' / 'st' x 7 redshifts x 3 multipoles = 84 networks "
                        '(5 -> 5 x 64 tanh -> n_m x 60), amplitude rescale by logA, REPT tracer between two emulated redshifts ({:d} networks reach the device), 19-monomial combination + cubic interpolation '
                        'to n_kin = 400 + window 120 x 1200 + 5 analytically marginalised parameters, {:d} batched points'.format(int(groups[:, 1].max()), B),
            'value': B / elapsed, 'unit': 'evals/s', 'ms_per_step': 1e3 * elapsed, 'steps': steps, 'dtype': 'f64', 'batch': B,
            'roofline': {'bound': 'mfma', 'kernel': 'dl_stk_chain_kernel (one wave per (network, 16-point tile): every layer by MFMA, no barrier) + dl_emulated_stacked_gemm_kernel (folded-operator product per monomial group; '
                                                      'Gram matrices and the marginalised solve in its tail): two launches per step, priced on the sum of their intervals', 'flop_per_eval': flops,
                         'flop_per_launch': fused * B, 'avg_launch_ms': launch_ms, 'launches_ms': {'network_chains': kernel_ms['theory'], 'feature_gemms': kernel_ms['window_gemm']},
                         'achieved': achieved, 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP64_TFLOPS,
                         'flop_count': "the build's own algorithm: final layers x y-scalers x assembly x redshift blend x interpolation x window x L^T folded at create"},
            'kernel_ms': {name: kernel_ms[name] for name in ['theory', 'window_gemm', 'finalize']},
            'oracle_check': {'points': ncheck, 'max_rel_err_vs_oracle': float(err), 'tolerance': 1e-10},
            'reference_check': {'fixture': 'tests/golden/boundary_cfg3_stacked_bench.npz', 'points': int((st == 0).sum()), 'max_rel_err_vs_reference': float(ref_err), 'tolerance': 1e-10}}


def _tns_config(device, steps, ncheck, orc):
    import gc
    import torch
    from bench_configs import load_tns, tns_oracle_point, spec_from_tns_golden
    from desilike_amd._lib import Context
    g = load_tns('tns')
    ctx = Context(spec_from_tns_golden(g), device=device.index or 0)
    B, tsteps = 4096, max(4, steps // 8)
    rng = np.random.RandomState(11)
    names = [str(name) for name in g['names']]
    lo = dict(qpar=0.97, qper=0.97, dm=-0.03, df=0.9, sigmav=0., b1=1.2, b2=-1., bs=-1., b3=-1., sn0=-0.5)
    hi = dict(qpar=1.03, qper=1.03, dm=0.03, df=1.1, sigmav=6., b1=2.8, b2=1., bs=1., b3=1., sn0=0.5)
    theta_host = np.column_stack([rng.uniform(lo[name], hi[name], B) for name in names])
    theta = torch.as_tensor(theta_host, dtype=torch.float64, device=device).contiguous()
    post, status = torch.empty(B, dtype=torch.float64, device=device), torch.zeros(B, dtype=torch.int32, device=device)
    elapsed, kernel_ms = _timed_context(ctx, theta, tsteps, 2, post, status, device)
    assert int((status != 0).sum().item()) == 0
    ntns = min(ncheck, 4)
    loglike = ctx.eval_batch_host(theta_host[:ntns])[0]
    q = g['c.k11']
    kernels = orc.tns_kernels(g['k11_table'], q, orc.weights_trapz(q))
    err = 0.
    t0 = time.perf_counter()
    for i in range(ntns):
        ref = tns_oracle_point(g, theta_host[i], kernels=kernels)
        err = max(err, abs(loglike[i] - ref) / max(1., abs(ref)))
    oracle_seconds = (time.perf_counter() - t0) / ntns
    assert err <= 1e-10, 'GPU / oracle mismatch on the TNS theory: {:.3e}'.format(err)
    n11, nq, nmu, nkin = len(g['k11_table']), len(q), 10, len(g['c.kin'])
    flops = {'loop_gemm_algorithmic': 2 * n11 * (nq * nmu * 27 + nq * 12), 'loop_gemm_executed': 2 * n11 * (-(-nq * nmu // 16) * 16 * 32 + -(-nq // 4) * 4 * 16),
             'assembly': 2 * 29 * 5 * n11 + 2 * 5 * n11 * n11 + 60 * nkin * len(g['c.mu']), 'window_gemm': 2 * len(g['c.flatdata']) * 3 * nkin}
    achieved = flops['loop_gemm_algorithmic'] * B / (kernel_ms['theory'] * 1e-3) / 1e12
    result = ({'workload': 'TNS one-loop theory (reference: tns_pt, full_shape.py:749-833): 29 loop tables on {:d} wavenumbers from {:d} template wavenumbers x {:d} cosines per point, spline / AP / FoG / '
                            'projection to 3 x {:d} multipoles, window 120 x {:d}, Gaussian likelihood, {:d} batched points'.format(n11, nq, nmu, nkin, 3 * nkin, B),
                'value': B / elapsed, 'unit': 'evals/s', 'ms_per_step': 1e3 * elapsed, 'steps': tsteps, 'dtype': 'f64', 'batch': B,
                'roofline': {'bound': 'mfma', 'kernel': 'dl_tns_loop_kernel (per table wavenumber: [points x 5000 pairs (mu, q)] . [5000 x 27 + 500 x 12] fp64 MFMA GEMM, left operand formed in registers from '
                                                        'LDS-resident templates)', 'flop_per_eval': flops, 'flop_per_launch': flops['loop_gemm_algorithmic'] * B, 'avg_launch_ms': kernel_ms['theory'],
                             'achieved': achieved, 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP64_TFLOPS,
                             'flop_count': 'the 27 bilinear + 12 linear tables the reference integrates (the kernel pads them to 32 + 16 columns and the pair list to a multiple of 16: executed count beside it)'},
                'kernel_ms': {name: kernel_ms[name] for name in ['theory', 'window_gemm', 'finalize']},
                'oracle_check': {'points': ntns, 'max_rel_err_vs_oracle': float(err), 'tolerance': 1e-10},
                'cpu_baseline': {'value': 1. / oracle_seconds, 'unit': 'evals/s', 'cores': 1, 'kind': 'port', 'sample': '{:d} evaluations of the NumPy oracle (kernels precomputed), 1 process'.format(ntns),
                                 'reference_in_build_container': {'value': 2.44, 'unit': 'evals/s', 'cores': 1, 'where': "the reference's own TNS likelihood (numpy fallback of its jax code, "
                                                                  "tests/golden/make_tns_fixture.py harness), build container (Xeon 2.1 GHz)"}}})
    ctx.close()
    del ctx
    gc.collect()
    return result


def sustained_leg(step, barrier, seconds, B, world, seconds_per_step, group=None):
    """The same step for >= ``seconds`` of wall time (the driver's GPU-busy sampling sees nothing of a 0.5 ms timed region): evaluations / s over the whole stretch.
    The number of steps is fixed beforehand from the timed region's rate (the same on every rank: the exchange is issued per step)."""
    nsteps = 256 * max(1, int(np.ceil(1.15 * seconds / seconds_per_step / 256.)))
    barrier()
    t0 = time.perf_counter()
    for block in range(nsteps // 256):
        for _ in range(256): step()
        if block % 64 == 63: barrier()      # keep the launch queue bounded
    barrier()
    elapsed = time.perf_counter() - t0
    if group is not None: elapsed = group.max(elapsed)
    return {'seconds': elapsed, 'steps': nsteps, 'value': world * B * nsteps / elapsed, 'unit': 'evals/s', 'ms_per_step': 1e3 * elapsed / nsteps}


def streams_leg(likelihood, device, B, ks=(1, 2, 4), steps=400, seed=777):
    """K independent batches in flight on K HIP streams (one context replica each: same constants, own workspaces): each kernel of a 1024-point step fills the chip
    for one round and pays its ramp and drain alone; with several steps in flight those tails overlap.  Evaluations / s at K = 1, 2, 4."""
    import torch
    out = []
    for K in ks:
        ctxs = [likelihood._get_context(replica=i) for i in range(K)]
        streams = [torch.cuda.Stream(device=device) for _ in range(K)]
        thetas = [torch.as_tensor(sample_theta(likelihood, B, seed + i), dtype=torch.float64, device=device).contiguous() for i in range(K)]
        posts = [torch.empty(B, dtype=torch.float64, device=device) for _ in range(K)]
        stats = [torch.zeros(B, dtype=torch.int32, device=device) for _ in range(K)]
        torch.cuda.synchronize(device)

        def run(n):
            for _ in range(n):
                for ctx, stream, theta, post, stat in zip(ctxs, streams, thetas, posts, stats):
                    ctx.eval_logposterior(theta, post, status=stat, stream=stream.cuda_stream)
            torch.cuda.synchronize(device)

        run(20)
        t0 = time.perf_counter()
        run(steps)
        elapsed = time.perf_counter() - t0
        assert all(int((stat != 0).sum().item()) == 0 for stat in stats)
        if K > 1:   # same points through replica 0 alone: same bits
            check = torch.empty(B, dtype=torch.float64, device=device)
            ctxs[0].eval_logposterior(thetas[-1], check)
            torch.cuda.synchronize(device)
            assert torch.equal(check, posts[-1]), 'context replicas disagree'
        out.append({'streams': K, 'value': K * B * steps / elapsed, 'unit': 'evals/s', 'us_per_step': 1e6 * elapsed / (K * steps)})
    return out


def chains_weak(group, local_rank, rank, world, chains_per_gpu=(1, 2, 4), iterations=300, warmup=100):
    """Chain-parallel sampling THROUGH THE SAMPLER (the reference's own scaling mode: one chain per group of ranks, desilike/utils.py:1040-1148): K chains per GPU,
    each a device-resident 512-walker ensemble on the two-tracer likelihood with its own Philox key and HIP stream; nothing is exchanged inside the run, the new samples
    of all chains are all-gathered once at its end (what `check_every` does in production).  Weak scaling: chains = K x n_gpus."""
    import torch
    from desilike_amd.samplers import EmceeSampler
    from desilike_amd.parallel import WalkerSharding
    likelihood = make_likelihood_config5(local_rank)
    device = torch.device('cuda', local_rank)
    out = []
    for K in chains_per_gpu:
        sampler = EmceeSampler(likelihood, nwalkers=512, chains=K * world, seed=42, sharding=WalkerSharding(group=group if group is not None else False), device_resident=True)
        # three untimed batches: the staging buffers exist, and the host-side sample store of every chain (amortised doubling, desilike_amd/samplers.py::_ChainStore) has
        # grown twice -- a growth is a fresh allocation whose pages are touched for the first time (6 ms per 12 MB chain, tools/chains_run_probe.py), and round 4 timed exactly
        # the one batch that pays it for every chain (K = 2 looked slower in aggregate than K = 1).  Timed: `nbatch` batches, the growth of the fifth among them (amortised).
        for _ in range(3): sampler.run(niterations=iterations)
        torch.cuda.synchronize(device)
        if group is not None: group.barrier()
        nbatch = 4
        t0 = time.perf_counter()
        for _ in range(nbatch): sampler.run(niterations=iterations)      # each: enqueue K ensembles on K streams, drain, all-gather the chains
        torch.cuda.synchronize(device)
        if group is not None: group.barrier()
        elapsed = (time.perf_counter() - t0) / nbatch
        if group is not None: elapsed = group.max(elapsed)
        logp = np.array([chain['logposterior'][-1] for chain in sampler.chains])
        assert np.isfinite(logp).all() and len(sampler.chains) == K * world
        entry = {'chains_per_gpu': K, 'chains': K * world, 'value': K * world * 512 * iterations / elapsed, 'unit': 'evals/s', 'us_per_update_per_chain': 1e6 * elapsed / iterations,
                 'batches_timed': nbatch, 'includes': 'enqueue, device run, drain of the chains to the host, append to the host-side sample store, all-gather of the chains across ranks'}
        if rank == 0 and K == chains_per_gpu[-1]:
            # every chain's final ensemble against the oracle (16 walkers each)
            names = likelihood.varied_params.names()
            worst = 0.
            for chain in sampler.chains:
                coords = np.column_stack([chain[name][-1] for name in names])[::32]
                ref = oracle_logposterior(likelihood, coords)
                worst = max(worst, float((np.abs(chain['logposterior'][-1][::32] - ref) / np.maximum(1., np.abs(ref))).max()))
            assert worst <= 1e-10, 'GPU / oracle mismatch on the chains: {:.3e}'.format(worst)
            entry['oracle_check'] = {'points': 16 * len(sampler.chains), 'max_rel_err_vs_oracle': worst, 'tolerance': 1e-10}
            entry['gelman_rubin_eigen_minus_1'] = None
            if K * world > 1:
                sampler.check(max_eigen_gr=0.03)
                entry['gelman_rubin_eigen_minus_1'] = float(sampler.diagnostics['eigen_gr'][-1])
        out.append(entry)
        for runner in sampler._runners.values(): runner.ens.close()
    best = max(out, key=lambda entry: entry['value'])
    return {'workload': 'EmceeSampler(chains = K x n_gpus, nwalkers = 512) on two config-2 tracers (n = 240): chain-parallel, device-resident, {:d} updates per chain'.format(iterations),
            'scaling': 'weak', 'n_gpus': world, 'value': best['value'], 'unit': 'evals/s', 'best_chains_per_gpu': best['chains_per_gpu'], 'per_k': out,
            'exchange': 'none inside the run; one all-gather of the chains at the end' if world > 1 else 'none (single rank)'}


def mh_chains(local_rank, configs=((256, 1), (256, 4)), tries=300):
    """The reference's own Metropolis-Hastings sampler (desilike/samplers/mcmc.py) THROUGH THE SAMPLER on the two-tracer likelihood: C chains x V speculative proposals per
    try are one batch, device-resident (dl_mh_*).  Single rank, no collective.  Every chain's current state is checked against the oracle."""
    import torch
    from desilike_amd.samplers import MCMCSampler
    from desilike_amd.parallel import WalkerSharding
    likelihood = make_likelihood_config5(local_rank)
    device = torch.device('cuda', local_rank)
    names = likelihood.varied_params.names()
    out = []
    for C, V in configs:
        sampler = MCMCSampler(likelihood, chains=C, vectorize=V, seed=42, sharding=WalkerSharding(group=False))
        sampler.run(check_every=tries, max_iterations=3 * tries)  # warm-up: burn-in from the reference distributions, the proposal covariance learnt from the chains twice
        sampler.learn = False                                      # (timed: the sampling itself)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        sampler.run(check_every=tries, max_iterations=tries)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        coords = np.array([state[0] for state in sampler._state])[::max(1, C // 32)]
        logp = np.array([state[1] for state in sampler._state])[::max(1, C // 32)]
        ref = oracle_logposterior(likelihood, coords)
        worst = float((np.abs(logp - ref) / np.maximum(1., np.abs(ref))).max())
        assert worst <= 1e-10, 'GPU / oracle mismatch on the Metropolis-Hastings chains: {:.3e}'.format(worst)
        naccepted = int(sum(state[3] for state in sampler._state))
        out.append({'chains': C, 'vectorize': V, 'rows_per_try': C * V, 'value': C * V * tries / elapsed, 'unit': 'evals/s', 'us_per_try': 1e6 * elapsed / tries,
                    'accepted_moves_per_s': sum(len(chain['fweight']) for chain in sampler.chains) / elapsed if all(chain is not None for chain in sampler.chains) else None,
                    'mean_acceptance_rate': float(np.nanmean(sampler.acceptance_rate)), 'accepted_moves_total': naccepted,
                    'includes': 'enqueue, device run, drain of the recorded states to the host', 'oracle_check': {'points': int(len(ref)), 'max_rel_err_vs_oracle': worst, 'tolerance': 1e-10}})
        sampler._runner.close()
    best = max(out, key=lambda entry: entry['value'])
    return {'workload': 'MCMCSampler (blocked Metropolis-Hastings, desilike/samplers/mcmc.py) on two config-2 tracers (n = 240, 8 parameters): chains x speculative proposals in one batch, '
                        'device-resident, {:d} tries'.format(tries), 'n_gpus': 1, 'value': best['value'], 'unit': 'evals/s', 'per_config': out}


def dry_run(rank, world):
    """Everything of the N-rank launch path that does not need a GPU: ranks started, host-side group formed, one exchange, rank 0 prints the line skeleton."""
    from desilike_amd import parallel
    group = None
    gathered = np.array([0.])
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        group = parallel.TorchGroup()
        gathered = parallel.WalkerSharding(group=group, min_shard_rows=0).map(lambda rows: rows[:, 0] * 2., np.arange(10.)[:, None])
        assert np.array_equal(gathered, 2. * np.arange(10.))
        elapsed = group.max(float(rank))
        assert elapsed == world - 1
        group.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'log-likelihood evals/sec (full-shape P_ell, 3x40 bins)', 'value': None, 'n_gpus': world, 'dry_run': True,
                          'config': {'ranks': group.world if group is not None else 1, 'collective': 'torch.distributed gloo' if group is not None else 'none'}}), flush=True)
    if group is not None:
        group.barrier()
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=200)
    parser.add_argument('--warmup', type=int, default=20)
    parser.add_argument('--batch', type=int, default=BATCH)
    parser.add_argument('--prewarm-ms', type=float, default=400., help='untimed fixed-duration run of the step before the W warm-up steps (clocks ramp up; reported as prewarm_ms)')
    parser.add_argument('--config5-iterations', type=int, default=300, help='ensemble updates of the strong-scaling configs[4] measurement (0: skip)')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--sustained-seconds', type=float, default=3., help='fixed-duration run of the same step after the timed region, reported as `sustained` (0: skip)')
    parser.add_argument('--no-other-configs', action='store_true', help='skip BASELINE configs[2] / configs[3] (`other_configs`)')
    parser.add_argument('--no-streams', action='store_true', help='skip the K-batches-in-flight measurement (`streams`)')
    parser.add_argument('--no-host-call', action='store_true', help='skip the host-array entry point measurement (`host_call`)')
    parser.add_argument('--chains-iterations', type=int, default=300, help='ensemble updates per chain of the chain-parallel sampler measurement `chains_weak` (0: skip)')
    parser.add_argument('--dry-run', action='store_true', help='launcher check without a GPU: start the ranks, form the (gloo) group, exchange, print the line skeleton')
    parser.add_argument('--no-events', action='store_true', help='diagnostic: no HIP events attached to the kernels in the timed region')
    parser.add_argument('--allow-torch-fallback', action='store_true', help="diagnostic: if RCCL through the C ABI (dl_comm_*) fails, go on with torch.distributed's nccl binding instead of failing")
    args = parser.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))    # (before anything that could touch HIP is imported)

    import torch
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.dry_run:
        return dry_run(rank, world)
    backend = os.environ.get('DL_BENCH_BACKEND', 'rccl')   # 'gloo': smoke test of the N > 1 code path on a 1-GPU box (ranks share the GPU, host-side exchange)
    force_dist = os.environ.get('DL_BENCH_FORCE_DIST', '0') == '1'   # single-rank RCCL communicator: exercises the collective calls on a 1-GPU box
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU: the hot path has no CPU fallback')
    ndev = torch.cuda.device_count()
    if world > ndev and backend == 'rccl':
        raise RuntimeError('{:d} ranks but {:d} GPU(s) visible: RCCL needs one GPU per rank (DL_BENCH_BACKEND=gloo runs the N > 1 code path with the ranks sharing a GPU)'.format(world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    group, collective = None, 'none'
    if world > 1 or force_dist:
        from desilike_amd import parallel
        if backend == 'rccl':
            # The exchange of the N > 1 line is RCCL through the library's own C ABI (dl_comm_*).  If that fails the run FAILS (per-rank error line, non-zero exit):
            # a line measured on torch.distributed's binding instead would hide a broken communicator behind a green result (--allow-torch-fallback: diagnostics only).
            try:
                group = init_rccl_group(parallel, local_rank, rank, world, timeout=float(os.environ.get('DL_COMM_TIMEOUT', 180.)))
                collective = 'RCCL {} through the C ABI (dl_comm_allgather_f64)'.format(group.rccl_version)
            except Exception as exc:
                print('bench.py rank {:d} / {:d} (GPU {:d}): RCCL through the C ABI failed: {}'.format(rank, world, local_rank, exc), file=sys.stderr, flush=True)
                if not args.allow_torch_fallback:
                    sys.stderr.flush()
                    os._exit(3)   # (not sys.exit: a rank stuck in ncclCommInitRank on another thread must not keep the process alive)
                import torch.distributed as dist
                dist.init_process_group(backend='nccl', rank=rank, world_size=world)
                group = parallel.TorchGroup(device=local_rank)
                collective = 'torch.distributed nccl (RCCL) -- FALLBACK, the C-ABI communicator failed'
        else:
            import torch.distributed as dist
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            group = parallel.TorchGroup()
            collective = 'torch.distributed ' + backend
        parallel.set_default_group(group)
    distributed = group is not None

    likelihood = make_likelihood(local_rank)
    ctx = likelihood._get_context()
    B = args.batch
    # N > 1: independent walker ensembles (chains) are kept in flight so that the exchange of log-posteriors never stalls the evaluation: the finalize kernel
    # writes each step's log-posteriors straight into a bucket, and one asynchronous RCCL all-gather (its own stream) ships a bucket of GATHER_EVERY steps while the
    # next bucket is being evaluated (desilike_amd/parallel.py).  Every step is still one pass of the hot path over B points per GPU; every result is all-gathered.
    nslots = 2 if distributed else 1
    theta_host = sample_theta(likelihood, B, seed=42 + rank)
    thetas = [torch.as_tensor(theta_host if slot == 0 else sample_theta(likelihood, B, seed=4242 + rank), dtype=torch.float64, device=device).contiguous() for slot in range(nslots)]
    loglikes = [torch.empty(B, dtype=torch.float64, device=device) for slot in range(nslots)]
    logpriors = [torch.empty(B, dtype=torch.float64, device=device) for slot in range(nslots)]
    statuses = [torch.zeros(B, dtype=torch.int32, device=device) for slot in range(nslots)]
    loglike = loglikes[0]
    stream = torch.cuda.current_stream(device)
    bucket = None
    if distributed:
        from desilike_amd.parallel import BucketedAllGather
        bucket = BucketedAllGather(B, torch.float64, device, steps_per_bucket=GATHER_EVERY, keep=False, group=group, force_collective=force_dist)
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
            torch.cuda.synchronize(device)
            group.barrier()
        torch.cuda.synchronize(device)

    import gc
    gc.collect()   # whatever set-up garbage holds device resources is released now, not by a collector pass inside the timed loop (hipFree synchronises the device)
    # fixed-duration pre-warm (its own key in the line): a 20-step run is 0.6 ms of GPU work -- without it the timed region would start on idle clocks
    # (the pre-warm evaluates without the exchange: its length is time-based, hence different on every rank -- collectives must be issued in the same number everywhere)
    scratch = torch.empty(B, dtype=torch.float64, device=device)
    t0 = time.perf_counter()
    prewarm_steps = 0
    while 1e3 * (time.perf_counter() - t0) < args.prewarm_ms:
        for _ in range(16): ctx.eval_logposterior(thetas[0], scratch, status=statuses[0], stream=stream.cuda_stream)
        torch.cuda.synchronize(device)
        prewarm_steps += 16
    prewarm_ms = 1e3 * (time.perf_counter() - t0)
    # Long runs: all three kernels on 8+ sampled steps (0.4 us per step of overhead at 200 steps).  Short runs (the driver's --steps 20): a kernel launched with
    # events is followed by a ~3 us gap, so only the kernel that dominates the step (the theory kernel: established by the long runs and by rocprofv3,
    # profiles/) carries events, on every sixth step: 3 samples at --steps 20 (5 until round 3: the `sustained` leg below runs without events and the two figures are meant
    # to be compared).  A sampled launch costs the step ~5 us (measured at --steps 20: 26.9 - 27.2 us per step with 10 samples, 24.4 - 24.8 us without events;
    # tools/sync_probe.py: the closing synchronisation is not the difference); the other kernels are reported as null.
    short = args.steps < 100
    only = 'theory' if short else None
    every = max(1, min(25, args.steps // (3 if short else 8)))
    if not args.no_events: ctx.profile_enable(1, only=only)   # the warm-up steps go through the event path too (its first use allocates: not a sample)
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events attached to the kernels' own dispatch packets on the launch stream (hipExtLaunchKernelGGL: no event records between the kernels), on 8 or more
    # steps of a long timed region, 5 of the driver's 20
    ctx.profile_enable(0 if args.no_events else every, only=only)
    gc.disable()
    elapsed = timed_steps(step, barrier, args.steps)
    gc.enable()
    kernel_ms = ctx.profile_read() if not args.no_events else dict(theory=1., window_gemm=1., finalize=1., total=1., event_overhead=0., samples=0, samples_per_kernel={})
    ctx.profile_enable(0)
    if distributed:
        elapsed = group.max(elapsed)

    assert all(int((st != 0).sum().item()) == 0 for st in statuses), 'non-OK status in the benchmark batch'
    gathered_check = None
    if distributed:
        # the exchange, checked end to end (outside the timed region): every rank's log-posteriors of its first batch are all-gathered through the group; every rank must hold
        # the SAME gathered array (a checksum of it is all-gathered and compared), and rank 0 checks a few points of EVERY rank's slice against the oracle (it can draw any
        # rank's batch: the seeds are 42 + rank)
        post = torch.empty(B, dtype=torch.float64, device=device)
        ctx.eval_logposterior(thetas[0], post, status=statuses[0], stream=stream.cuda_stream)
        torch.cuda.synchronize(device)
        everyone = np.asarray(group.allgather(post.cpu().numpy())).reshape(world, B)
        assert np.array_equal(everyone[rank], post.cpu().numpy()), 'the gathered array does not hold this rank\'s own log-posteriors in its slice'
        digest = float(np.frombuffer(everyone.tobytes(), dtype=np.uint32).astype(np.uint64).sum() % (1 << 52))
        digests = np.asarray(group.allgather(np.array([digest]))).reshape(world)
        assert (digests == digest).all(), 'ranks hold different gathered log-posteriors: {}'.format(digests)
        if rank == 0:
            worst, npts = 0., 4
            for r in range(world):
                rows = sample_theta(likelihood, B, seed=42 + r)[:npts]
                ref = oracle_logposterior(likelihood, rows)
                worst = max(worst, float((np.abs(everyone[r, :npts] - ref) / np.maximum(1., np.abs(ref))).max()))
            assert worst <= 1e-10, 'gathered log-posteriors / oracle mismatch: {:.3e}'.format(worst)
            gathered_check = {'ranks': world, 'identical_on_every_rank': True, 'oracle_points_per_rank': npts, 'max_rel_err_vs_oracle': worst, 'tolerance': 1e-10}
    sustained = sustained_leg(step, barrier, args.sustained_seconds, B, world, elapsed / args.steps, group=group) if args.sustained_seconds > 0. else None
    def guarded(name, leg):
        # secondary, single-rank legs (no collective inside): a failure there is REPORTED in the line ({'error': ...}), it does not take the headline measurement with it
        try:
            return leg()
        except Exception as exc:
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {'leg': name, 'error': repr(exc)}

    streams = guarded('streams', lambda: streams_leg(likelihood, device, B)) if (not args.no_streams and not distributed and B == BATCH) else None
    others = guarded('other_configs', lambda: other_configs(device)) if (not args.no_other_configs and rank == 0 and B == BATCH) else None
    host_call = guarded('host_call', lambda: host_call_leg(likelihood)) if (rank == 0 and B == BATCH and not args.no_host_call) else None
    chains = chains_weak(group if (distributed and world > 1) else None, local_rank, rank, world, iterations=args.chains_iterations) if (args.chains_iterations > 0 and B == BATCH) else None
    mh = guarded('mh_chains', lambda: mh_chains(local_rank)) if (args.chains_iterations > 0 and rank == 0 and B == BATCH) else None
    strong = None
    if args.config5_iterations > 0 and B == BATCH:
        strong = config5_strong(group, device, local_rank, rank, world, args.config5_iterations)
    if rank == 0:
        value = world * B * args.steps / elapsed
        flops = {'theory': FLOP_THEORY, 'window_gemm': FLOP_GEMM, 'finalize': FLOP_FINAL}
        dominant = 'theory' if short else max(['theory', 'window_gemm', 'finalize'], key=lambda name: kernel_ms[name])
        kernel_name = {'theory': 'dl_fullshape_kernel', 'window_gemm': 'dl_chi2_gemm_kernel', 'finalize': 'dl_finalize_part_kernel'}[dominant]
        traffic, traffic_source = hbm_traffic(kernel_name) if B == BATCH else (None, None)
        if traffic_source is not None and not profile_is_current(traffic_source): traffic, traffic_source = None, None      # (a profile of another source tree says nothing about this one: ADVICE r5)
        # the whole step against its algorithmic bytes (fused: theta in, the whitened operand once, three outputs per point): PMC bytes of the step's three kernels
        step_kernels = ['dl_fullshape_kernel', 'dl_chi2_gemm_kernel<true, true, 32', 'dl_finalize_part_kernel']
        step_traffic = [hbm_traffic(name)[0] for name in step_kernels] if (B == BATCH and traffic_source is not None) else [None]
        traffic_step = float(sum(step_traffic)) if all(t is not None for t in step_traffic) else None
        algorithmic_bytes = float(B * 6 * 8 + 128 * 1280 * 8 + B * (8 + 8 + 4))
        trace_ms, trace_source = trace_average(kernel_name) if B == BATCH else (None, None)
        if trace_source is not None and not profile_is_current(trace_source): trace_ms, trace_source = None, None
        per_launch = min(B, 32768)   # batches above 32768 points are evaluated in internal passes of 32768: the kernel intervals are per pass
        achieved = flops[dominant] * per_launch / (kernel_ms[dominant] * 1e-3) / 1e12
        # which unit bounds the dominant kernel: the theory kernel issues no MFMA (fp64 VALU: transcendentals, spline evaluation, projection -- its 78.6 TFLOP/s peak
        # equals the fp64 matrix peak); the chi2 GEMM is fp64 MFMA (v_mfma_f64_16x16x4_f64)
        bound = {'theory': 'valu', 'window_gemm': 'mfma', 'finalize': 'latency'}[dominant]
        result = {'metric': 'log-likelihood evals/sec (full-shape P_ell, 3x40 bins)', 'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
                  'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
                  'data': 'synthetic', 'prewarm_ms': prewarm_ms, 'prewarm_steps': prewarm_steps,
                  'config': {'workload': 'BASELINE configs[1]: ShapeFit+Kaiser P_ell ell=(0,2,4) x 40 k-bins, dense window 120x1200 (n_kin=400/ell), 120x120 precision, '
                                         '{:d} batched param points per GPU per step'.format(B), 'batch_per_gpu': B, 'n_params': 6,
                             'parallelism': 'walkers x{:d}'.format(world) + (', log-posteriors all-gathered in buckets of {:d} steps'.format(GATHER_EVERY) if distributed else ''),
                             'collective': collective, 'ranks': group.world if distributed else 1, 'rccl_version': getattr(group, 'rccl_version', None) if distributed else None},
                  'roofline': {'bound': bound, 'bound_detail': {'theory': 'fp64 VALU (no MFMA in this kernel: transcendentals, spline evaluation, projection); peak = 78.6 TFLOP/s fp64 vector',
                                                                 'window_gemm': 'fp64 MFMA (v_mfma_f64_16x16x4_f64), peak = 78.6 TFLOP/s fp64 matrix', 'finalize': 'launch latency'}[dominant],
                               'kernel': kernel_name, 'achieved': achieved, 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP64_TFLOPS, 'traffic': traffic,
                               'traffic_source': traffic_source, 'from_committed_profile': traffic_source is not None, 'source_hash': source_hash(), 'traffic_step': traffic_step, 'algorithmic_bytes': algorithmic_bytes,
                               'traffic_step_over_algorithmic': (traffic_step / algorithmic_bytes) if traffic_step is not None else None,
                               'traffic_step_kernels': dict(zip(['theory', 'window_gemm', 'finalize'], step_traffic)) if traffic_step is not None else None,
                               'flop_per_launch': flops[dominant] * per_launch, 'avg_launch_ms': kernel_ms[dominant],
                               'trace_avg_launch_ms': trace_ms, 'trace_source': trace_source,
                               'trace_frac': (flops[dominant] * per_launch / (trace_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS) if trace_ms else None,
                               'event_samples': int(kernel_ms.get('samples_per_kernel', {}).get(dominant, kernel_ms.get('samples', 0))),
                               'event_mode': 'the dominant kernel only, every {:d} steps (short run)'.format(every) if short else 'all kernels of a sampled step, every {:d} steps'.format(every)},
                  'kernel_ms': {name: (kernel_ms[name] if kernel_ms[name] > 0. else None) for name in ['theory', 'window_gemm', 'finalize']},
                  'kernel_frac_of_fp64_peak': {name: flops[name] * per_launch / (kernel_ms[name] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS for name in ['theory', 'window_gemm'] if kernel_ms[name] > 0.}}
        if gathered_check is not None: result['gathered_check'] = gathered_check
        if sustained is not None:
            sustained['agrees_with_value_within'] = abs(sustained['value'] / value - 1.)
            result['sustained'] = sustained
        if streams is not None: result['streams'] = streams
        if host_call is not None: result['host_call'] = host_call
        if others is not None: result['other_configs'] = others
        if chains is not None: result['chains_weak'] = chains
        if mh is not None: result['mh_chains'] = mh
        if strong is not None:
            result['config5_strong'] = strong
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
        group.barrier()
    if result is not None:
        # two lines: the complete record (every leg with its workload description, FLOP accounts, latencies), then -- LAST, the line the driver parses and keeps -- the
        # same headline with every leg reduced to its numbers (< 4 KB: the driver stores a bounded tail of the output)
        print(json.dumps(dict(result, record='complete')), flush=True)
        print(json.dumps(compact_line(result)), flush=True)
    if distributed:
        group.barrier()
        group.close()
        if backend != 'rccl':
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
