"""GPU (-m gpu): a seeded sweep over SHAPES of the ShapeFit + Kaiser likelihood (a1 - a9): multipoles in and out, number of bins, theory resolution of the binning window,
one or two tracers, dense covariances, fixed / varied parameters -- n = 10 .. 615 data points (N_pad 128 .. 640), K = 36 .. 1320 theory columns, ragged batches on either
side of the 2048-row switch between the chi2 GEMM and the split-K path.  Every configuration against the NumPy oracle (pinned on the reference's outputs at the shapes of
tests/golden) at 1e-10, and the same rows whatever batch they sit in."""
import numpy as np
import pytest

from bench import oracle_logposterior   # the post-hoc checker of bench.py (oracle/np_oracle.py through the host-side constants)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def build(seed):
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    rng = np.random.RandomState(seed)
    ells = [(0,), (0, 2), (0, 2, 4), (2, 4), (0, 4)][rng.randint(5)]
    nk = int(rng.randint(5, 42))
    kmin, kmax = [(0., 0.2), (0.02, 0.3), (0.01, 0.12)][rng.randint(3)]
    resolution = int(rng.randint(1, 11))
    tracers = [None, ('LRG', 'ELG')][int(rng.rand() < 0.35)]
    if seed >= 24: ells, nk = (0, 2, 4), 41
    if seed >= 20: tracers = [('LRG', 'ELG', 'QSO'), ('BGS', 'LRG', 'ELG', 'QSO', 'LAE')][seed % 2]   # n up to 615: beyond the 32 column blocks of the chi2 GEMM's panel table
    template = ShapeFitPowerSpectrumTemplate(z=float(rng.uniform(0.3, 1.4)), fiducial='synthetic')
    observables = []
    dense = bool(rng.rand() < 0.4)    # a dense survey-like window on its own theory grid, more theory multipoles than data multipoles (window.py:428-438 / 459-473)
    for tracer in ([None] if tracers is None else tracers):
        kwargs = {} if tracer is None else dict(tracers=tracer)
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, **kwargs)
        prefix = '' if tracer is None else tracer + '.'
        data = {prefix + 'b1': float(rng.uniform(1.2, 2.4)), 'dm': 0.01}
        kedges = np.linspace(kmin, kmax, nk + 1)
        if dense:
            import bench
            ellsin = (0, 2, 4)
            kin, full = bench.dense_window(kedges, ellsin, resolution=resolution, seed=int(rng.randint(1000)))    # [3 nk, 3 n_kin]: rows of the multipoles that are observed
            rows = np.concatenate([np.arange(nk) + nk * ellsin.index(ell) for ell in ells])
            wmatrix = dict(wmatrix=full[rows], kin=kin, ellsin=ellsin)
        else:
            wmatrix = dict(wmatrix={'resolution': resolution})
        observables.append(TracerPowerSpectrumMultipolesObservable(data=data, kedges=kedges, ells=ells, theory=theory, shotnoise=float(rng.uniform(2e3, 2e4)), **wmatrix))
    n = len(ells) * nk * len(observables)
    A = rng.standard_normal((n, n)) * 30.
    like = ObservablesGaussianLikelihood(observables=observables, covariance=A.dot(A.T) + 1e4 * np.eye(n))
    fixed = [name for name in ['dm', 'df', 'qpar'] if rng.rand() < 0.25]
    for name in fixed: like.all_params[name].update(fixed=True)
    like.initialize()
    return like, dict(ells=ells, nk=nk, resolution=resolution, dense=dense, tracers=tracers, n=n, fixed=fixed)


@pytest.mark.parametrize('seed', range(26))
def test_shape_against_the_oracle(seed):
    import bench
    like, info = build(seed)
    rng = np.random.RandomState(100 + seed)
    B = [1, 7, 100, 1000, 2049, 2500][seed % 6]
    theta = bench.sample_theta(like, B, seed=200 + seed)
    ctx, offset = like._get_posterior_context()
    logp, status = ctx.eval_logposterior_host(theta)
    logp = logp + offset
    inside = status == 0
    assert inside.mean() > 0.5 and np.isfinite(logp[inside]).all(), info
    rows = np.flatnonzero(inside)[:: max(1, inside.sum() // 12)][:12]
    ref = oracle_logposterior(like, theta[rows])
    err = np.abs(logp[rows] - ref) / np.maximum(1., np.abs(ref))
    assert (err <= TOL).all(), (info, err.max())
    # the same rows in another batch (other tile heights, the other GEMM path beyond 2048 rows): same numbers to the tolerance, same status
    other = ctx.eval_logposterior_host(np.concatenate([theta[rows], theta[:1].repeat(2100, axis=0)]))[0][:len(rows)] + offset
    assert np.allclose(other, logp[rows], rtol=TOL, atol=TOL), info
    print('seed {:d}: {} B = {:d}: max relative error {:.1e}'.format(seed, info, B, err.max()))
