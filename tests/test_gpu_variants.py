"""GPU (-m gpu): variants of the full-shape path the headline configuration does not exercise, each against the NumPy oracle on seeded points:
AP parametrisations (qiso / qap / qisoqap, theories/galaxy_clustering/base.py:341-350), Standard / Fixed templates (power_template.py:592-596, 198-202),
FoG damping, the cubic observable transform (power_spectrum.py:402-404), Hartlap / Percival factors (likelihoods/base.py:623-656), diagonal precision,
a ragged number of multipoles.  Tolerance: 1e-10 on logL (relative above 1), 1e-11 on the theory vector."""
import numpy as np
import pytest

from oracle import np_oracle as orc

pytestmark = pytest.mark.gpu

KEDGES = np.linspace(0.01, 0.2, 20)


def build(template='shapefit', apmode='qparqper', ells=(0, 2, 4), transform=None, covariance='dense', nobs=None, damping=False, seed=3):
    from desilike_amd.theories.galaxy_clustering import (ShapeFitPowerSpectrumTemplate, StandardPowerSpectrumTemplate, FixedPowerSpectrumTemplate,
                                                         KaiserTracerPowerSpectrumMultipoles)
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    cls = {'shapefit': ShapeFitPowerSpectrumTemplate, 'standard': StandardPowerSpectrumTemplate, 'fixed': FixedPowerSpectrumTemplate}[template]
    tpl = cls(z=0.8) if template == 'fixed' else cls(z=0.8, apmode=apmode)
    theory = KaiserTracerPowerSpectrumMultipoles(template=tpl)
    if damping:
        for name, value in [('sigmapar', 4.), ('sigmaper', 2.5)]:
            theory.init.params[name].update(fixed=False, value=value, prior=dict(limits=[0., 10.]), ref=dict(limits=[value - 1., value + 1.]))
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.8}, kedges=KEDGES, ells=ells, wmatrix={'resolution': 3}, theory=theory, shotnoise=5e3, transform=transform)
    n = (len(KEDGES) - 1) * len(ells)
    rng = np.random.RandomState(seed)
    kwargs = {}
    if covariance == 'dense':
        A = rng.standard_normal((n, n)) * 20.
        kwargs['covariance'] = A.dot(A.T) + 2e4 * np.eye(n)
    else:
        kwargs['precision'] = 1. / rng.uniform(1e4, 5e4, size=n)   # 1-D: diagonal precision (likelihoods/base.py:15-16)
    if nobs is not None:
        kwargs['correct_covariance'] = {'nobs': nobs, 'correction': 'hartlap-percival2014'}
    like = ObservablesGaussianLikelihood(observables=[obs], **kwargs)
    like.initialize()
    return like, obs, theory, tpl


def oracle_loglike(like, obs, theory, tpl, names, row, template, apmode, transform):
    p = dict(zip(names, row))
    wm = obs.wmatrix
    c = dict(template='shapefit' if template == 'shapefit' else 'fixed', k11=tpl.k, pk_dd_fid=tpl.pk_dd_fid, f_fid=tpl.f_fid, kp=getattr(tpl, 'kp', 0.03), a=getattr(tpl, 'a', 0.6),
             kin=theory.k, mu=theory.mu, wmu_ell=theory.wmu, ellsin=theory.ells, nd=theory.nd, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein,
             shotnoiseout=wm.shotnoiseout, flatdata=obs.flatdata, transform=transform)
    q = dict(p)
    if template != 'fixed':
        qpar, qper = orc.ap_qparqper(apmode, 1. / 3., **{name: p[name] for name in ['qpar', 'qper', 'qiso', 'qap'] if name in p})
        q.update(qpar=qpar, qper=qper)
    q['b1'] = (p['b1'], p['b1'])
    out = orc.fullshape_observable(c, q)
    return out['flattheory'], orc.gaussian_loglikelihood(out['flattheory'], obs.flatdata, like.precision)[0]


CASES = [dict(template='shapefit', apmode='qiso'), dict(template='shapefit', apmode='qap'), dict(template='shapefit', apmode='qisoqap', damping=True),
         dict(template='standard', apmode='qparqper'), dict(template='fixed'), dict(template='shapefit', transform='cubic'),
         dict(template='shapefit', covariance='diag'), dict(template='shapefit', nobs=400), dict(template='standard', apmode='qisoqap', ells=(0, 2)),
         dict(template='shapefit', transform='cubic', covariance='diag', ells=(0, 2))]


@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join('{}={}'.format(k, v) for k, v in c.items()))
def test_variant_vs_oracle(case):
    like, obs, theory, tpl = build(**case)
    names = like.varied_params.names()
    rng = np.random.RandomState(17)
    theta = np.column_stack([np.clip(param.ref.sample(size=48, random_state=rng), *param.prior.limits) for param in like.varied_params])
    theta[5] = [param.value for param in like.varied_params]
    loglike, logprior, status, flat = like._get_context().eval_batch_host(theta, return_flattheory=True)
    template, apmode, transform = case.get('template'), case.get('apmode', 'qparqper'), case.get('transform', None)
    if transform is None: assert (status == 0).all()
    nfinite = 0
    for i in range(0, 48, 3 if transform is None else 1):
        with np.errstate(invalid='ignore'):
            ref_flat, ref_ll = oracle_loglike(like, obs, theory, tpl, names, theta[i], template, apmode, transform)
        if not np.isfinite(ref_ll):
            # the cubic transform takes a real cube root of theory / data: where a multipole changes sign relative to the data the reference's numpy path
            # yields NaN as well (power_spectrum.py:404) -- reported as status 2 here, mapped to -inf by the samplers
            assert status[i] == 2 and not np.isfinite(loglike[i]), (case, i)
            assert np.array_equal(np.isnan(flat[i]), np.isnan(ref_flat))
            continue
        nfinite += 1
        assert status[i] == 0
        assert np.allclose(flat[i], ref_flat, rtol=1e-11, atol=1e-12 * np.abs(ref_flat).max()), (case, i)
        assert abs(loglike[i] - ref_ll) <= 1e-10 * max(1., abs(ref_ll)), (case, i, loglike[i], ref_ll)
    assert nfinite >= 8
    if case.get('nobs'):
        n = like.precision.shape[0]
        hartlap = (case['nobs'] - n - 2.) / (case['nobs'] - 1.)
        assert np.isclose(like.hartlap2007_factor, hartlap)


def _oracle_constants(like, obs, theory, tpl):
    wm = obs.wmatrix
    return dict(template='shapefit', k11=tpl.k, pk_dd_fid=tpl.pk_dd_fid, f_fid=tpl.f_fid, kp=tpl.kp, a=tpl.a, kin=theory.k, mu=theory.mu, wmu_ell=theory.wmu,
                ellsin=theory.ells, nd=theory.nd, matrix_full=wm.matrix_full, kmask=wm.kmask, offset=wm.offset, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout,
                flatdata=obs.flatdata)


def _check_against_oracle(like, obs, theory, tpl, nrows=24):
    names = like.varied_params.names()
    rng = np.random.RandomState(23)
    theta = np.column_stack([np.clip(param.ref.sample(size=nrows, random_state=rng), *param.prior.limits) for param in like.varied_params])
    loglike, logprior, status, flat = like._get_context().eval_batch_host(theta, return_flattheory=True)
    assert (status == 0).all()
    c = _oracle_constants(like, obs, theory, tpl)
    for i in range(0, nrows, 2):
        p = dict(zip(names, theta[i])); p['b1'] = (p['b1'], p['b1'])
        out = orc.fullshape_observable(c, p)
        assert np.allclose(flat[i], out['flattheory'], rtol=1e-11, atol=1e-12 * np.abs(out['flattheory']).max()), i
        ref = orc.gaussian_loglikelihood(out['flattheory'], obs.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)


def test_window_variants_vs_oracle():
    """Row selection without a window matrix (different k ranges per multipole: klim), and a user-provided dense matrix with rebinned input k,
    more input than output multipoles and a window shot-noise vector (window.py:249-324, 445-457)."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    rng = np.random.RandomState(9)
    # (1) k-range cuts per multipole on measured k's: theory evaluated on the union of k's, rows selected by kmask
    k = np.arange(0.015, 0.2, 0.01)
    tpl = ShapeFitPowerSpectrumTemplate(z=0.8)
    theory = KaiserTracerPowerSpectrumMultipoles(template=tpl)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.7}, k=k, ells=(0, 2, 4), klim={0: (0.02, 0.2), 2: (0.02, 0.15), 4: (0.05, 0.1)}, theory=theory, shotnoise=2e3)
    obs.initialize()
    n = sum(len(kk) for kk in obs.wmatrix.k)
    like = ObservablesGaussianLikelihood(observables=[obs], precision=1. / rng.uniform(1e4, 4e4, size=n))
    like.initialize()
    assert obs.wmatrix.kmask is not None and n < 3 * len(k)
    _check_against_oracle(like, obs, theory, tpl)
    # (2) dense user matrix: 3 input multipoles on a fine grid (rebinned by 2) -> 2 output multipoles, window shot-noise vector
    kout = np.arange(0.025, 0.2, 0.01)
    kin = np.linspace(0.001, 0.35, 140)
    nout = 2 * len(kout)
    wmat = np.abs(rng.standard_normal((nout, 3 * len(kin)))) * 0.02
    for ill in range(2):
        for i, kk in enumerate(kout):
            wmat[ill * len(kout) + i, ill * len(kin) + np.argmin(np.abs(kin - kk))] += 1.
    wmat /= wmat.sum(axis=1)[:, None]
    tpl = ShapeFitPowerSpectrumTemplate(z=0.8)
    theory = KaiserTracerPowerSpectrumMultipoles(template=tpl)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.7}, k=kout, ells=(0, 2), wmatrix=wmat, kin=kin, kinrebin=2, ellsin=(0, 2, 4), theory=theory, shotnoise=2e3,
                                                  wshotnoise=0.1 * rng.uniform(size=nout))
    A = rng.standard_normal((nout, nout)) * 15.
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + 1e4 * np.eye(nout))
    like.initialize()
    assert theory.k.size == 70 and obs.wmatrix.matrix_full.shape == (nout, 210)
    _check_against_oracle(like, obs, theory, tpl)


def test_chunked_batch_boundaries():
    """Batches above the 32768-point internal pass are looped inside dl_eval_batch: results must not depend on where the chunk boundary falls."""
    like, obs, theory, tpl = build(template='shapefit')
    rng = np.random.RandomState(31)
    B = 32768 + 77
    theta = np.column_stack([np.clip(param.ref.sample(size=B, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert (status == 0).all()
    pick = np.r_[0:5, 32760:32768 + 77]
    small = ctx.eval_batch_host(theta[pick])
    assert (np.abs(loglike[pick] - small[0]) <= 1e-10 * np.maximum(1., np.abs(small[0]))).all()
    assert np.array_equal(logprior[pick], small[1])


def test_xcd_local_point_assignment_is_a_pure_permutation():
    """Batches that are multiples of 256 take the XCD-local point assignment of the theory kernel (a row block is produced on the XCD whose chi2 GEMM workgroups consume
    it); other sizes keep launch order.  Same numbers either way, two observables included."""
    from test_host_api import make_cfg5
    g, like = make_cfg5()
    rng = np.random.RandomState(21)
    theta = np.column_stack([np.clip(param.ref.sample(size=768, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    full = ctx.eval_batch_host(theta)               # 768 = 3 x 256: XCD-local
    part = ctx.eval_batch_host(theta[:767])         # 767: launch order
    one = ctx.eval_batch_host(theta[512:513])
    assert np.array_equal(full[0][:767], part[0]) and np.array_equal(full[1][:767], part[1]) and np.array_equal(full[2][:767], part[2])
    assert full[0][512] == one[0][0]
