"""GPU (-m gpu): the PNG theory (scale-dependent bias from local primordial non-Gaussianity, csrc/dl_kernels.hip::dl_png_kernel) through the C ABI against golden vectors of
the reference itself (tests/golden/make_png_fixture.py) and, through the host mirror, against the NumPy oracle.  Tolerance: 1e-10 on logL (relative above 1)."""
import numpy as np
import pytest

from test_oracle_png import load, png_oracle_point, FIXTURES

pytestmark = pytest.mark.gpu


def spec_from_png_golden(g):
    from oracle import np_oracle as oc
    names = [str(n) for n in g['names']]

    def inp(name, default):
        return (names.index(name), default) if name in names else (-1, default)

    kt = g['c.k11']
    alpha = oc.png_alpha_prim(kt, g['c.pk_dd_fid'], g['pk_prim'], float(g['h']))
    inputs = {'qpar': inp('qpar', 1.), 'qper': inp('qper', 1.), 'df': inp('df', 1.), 'dm': inp('dm', 0.), 'dn': inp('dn', 0.), 'b1X': inp('b1', 1.), 'b1Y': inp('b1', 1.), 'sn0': inp('sn0', 0.),
              'fnl_loc': inp('fnl_loc', 0.), 'pX': inp('p', 1.), 'pY': inp('p', 1.), 'bphiX': inp('bphi', 1.), 'bphiY': inp('bphi', 1.), 'sigmas': inp('sigmas', 0.), 'sigmasY': inp('sigmas', 0.)}
    obs = dict(theory=np.array([5]), template=np.array([1 if str(g['c.template']).startswith('ShapeFit') else 0]), apmode=np.array([0]), transform=np.array([0]), eta=[1. / 3.],
               f_fid=[float(g['c.f_fid'])], a=[float(g['c.a']) if 'c.a' in g else 0.6], kp=[float(g['c.kp']) if 'c.kp' in g else 0.03], nd=[float(g['c.nd'])],
               ells_in=np.asarray(g['c.ellsin'], dtype='i4'), kin=g['c.kin'], mu=g['c.mu'], wmu_ell=g['c.wmu_ell'], k_t=kt[1:], pk_dd_fid=g['c.pk_dd_fid'][1:],
               wmatrix=g['c.matrix_full'], kmask=None, offset=None, shotnoise_in=g['c.shotnoisein'], shotnoise_out=g['c.shotnoiseout'], flatdata=g['c.flatdata'],
               png_alpha=alpha[1:], png_mode=np.array([{'bphi': 0, 'b-p': 1}[str(g['mode'])]], dtype='i4'), inputs=inputs)
    return dict(n_params=np.array([len(names)]), priors=g['priors'], precision=g['precision'], observables=[obs])


@pytest.mark.parametrize('name', FIXTURES)
def test_vs_reference(name):
    from desilike_amd._lib import Context
    g = load(name)
    ctx = Context(spec_from_png_golden(g), device=0)
    nint = len(g['int_power'])
    rows = g['theta'][:nint].copy()
    power = ctx.eval_theory_host(rows, iobs=0)
    assert np.allclose(power, g['int_power'], rtol=1e-10, atol=1e-12 * np.abs(g['int_power']).max())
    loglike, logprior, status, flat = ctx.eval_batch_host(g['theta'], return_flattheory=True)
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(loglike - g['loglikelihood']) <= tol).all(), (np.abs(loglike - g['loglikelihood']) / np.maximum(1., np.abs(g['loglikelihood']))).max()
    finite = np.isfinite(g['logprior'])
    assert np.allclose(logprior[finite], g['logprior'][finite], rtol=1e-13, atol=1e-13) and np.array_equal(status == 1, ~finite)
    # a ragged larger batch gives the same numbers
    theta = np.tile(g['theta'][finite], (40, 1))[:517]
    big = ctx.eval_batch_host(theta)[0]
    assert np.array_equal(big[:finite.sum()], loglike[finite])
    ctx.close()


def test_host_mirror_matches_oracle():
    from oracle import np_oracle as oc
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, PNGTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    template = ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic')
    theory = PNGTracerPowerSpectrumMultipoles(template=template, mode='bphi')
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'fnl_loc': 15., 'bphi': 2.}, kedges=np.linspace(0.002, 0.102, 26), ells=(0, 2), wmatrix={'resolution': 2}, theory=theory, shotnoise=1e4)
    rng = np.random.RandomState(2)
    A = rng.standard_normal((50, 50)) * 300.
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + 4e6 * np.eye(50))
    like.initialize()
    names = like.varied_params.names()
    assert {'fnl_loc', 'bphi', 'b1', 'sn0', 'sigmas'} <= set(names) and 'p' not in names
    theta = np.column_stack([param.ref.sample(size=6, random_state=rng) for param in like.varied_params])
    loglike = np.array([like(**dict(zip(names, row))) - like.logprior for row in theta])
    wm, kt, fid = obs.wmatrix, template.k, template.fiducial
    for i, row in enumerate(theta):
        p = dict(zip(names, row))
        pk_dd = template.pk_dd_fid * oc.shapefit_factor(kt, template.kp, template.a, dm=p.get('dm', 0.), dn=p.get('dn', 0.))
        alpha = oc.png_alpha_prim(kt, pk_dd, fid.pk_prim(kt), fid.h)
        bf = oc.png_bfnl('bphi', p['b1'], fnl_loc=p['fnl_loc'], bphi=p['bphi'])
        power = oc.png_tracer_power(theory.k, theory.mu, theory.wmu, kt, pk_dd, alpha, template.f_fid * p.get('df', 1.), theory.nd, p['b1'], p['b1'], bf, bf, sn0=p['sn0'],
                                    sigmasX=p['sigmas'], sigmasY=p['sigmas'], qpar=p.get('qpar', 1.), qper=p.get('qper', 1.))
        flat = oc.window_apply(power, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout)
        ref = oc.gaussian_loglikelihood(flat, obs.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (i, loglike[i], ref)
