"""The tracer-velocity variant of the PNG theory (reference primordial_non_gaussianity.py:196-330): the oracle on the CPU and the device path on the GPU against multipoles
of the reference's own class (tests/golden/make_png_velocity_fixture.py: 81 trapezoid nodes in mu on [-1, 1]; the device folds them onto mu >= 0)."""
import os

import numpy as np
import pytest

from oracle import np_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return dict(np.load(os.path.join(HERE, 'golden', 'png_velocity.npz')))


def oracle_power(g, row, mu=None, wmu_ell=None):
    names = [str(n) for n in g['names']]
    p = dict(zip(names, row))
    k11 = g['k11']
    factor = orc.shapefit_factor(k11, float(g['kp']), float(g['a']), dm=p['dm'])
    pk = g['pk_dd_fid'] * factor
    alpha = g['alpha_fid'] / np.sqrt(factor)          # alpha ~ 1 / sqrt(P_dd) (primordial_non_gaussianity.py:291-292)
    f = float(g['f_fid']) * p['df']
    bfnl = 2. * 1.686 * (p['b1'] - p['p']) * p['fnl_loc']
    return orc.png_velocity_power(g['k'], g['mu'] if mu is None else mu, g['wmu_ell'] if wmu_ell is None else wmu_ell, k11[1:], pk[1:], alpha[1:], f, float(g['z']), p['b1'], bfnl,
                                  bv=p['bv'], sigmas=p['sigmas'], sigmau=p['sigmau'], qpar=p['qpar'], qper=p['qper'])


def folded_grid(g):
    mu, w = g['mu'], g['wmu_ell']
    half = mu >= -1e-12
    return np.abs(mu[half]), np.where(np.abs(mu[half]) > 0., 2., 1.) * w[:, half]


def test_oracle_against_the_reference():
    g = load()
    assert tuple(g['ells']) == (1, 3) and g['mu'].size == 81
    scale = np.abs(g['power']).max()
    for i, row in enumerate(g['theta']):
        assert np.allclose(oracle_power(g, row), g['power'][i], rtol=1e-11, atol=1e-13 * scale)
    # the integrand times an odd Legendre polynomial is even in mu: the 41 nodes mu >= 0 with the mirror weights added give the same sums
    mu, wmu = folded_grid(g)
    assert mu.size == 41
    for i in [0, 5]:
        assert np.allclose(oracle_power(g, g['theta'][i], mu=mu, wmu_ell=wmu), g['power'][i], rtol=1e-11, atol=1e-13 * scale)


def test_host_mirror_grid_and_parameters():
    from desilike_amd.theories.galaxy_clustering import PNGTracerVelocityPowerSpectrumMultipoles, ShapeFitPowerSpectrumTemplate
    g = load()
    theory = PNGTracerVelocityPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'), k=g['k'])
    theory.initialize()
    mu, wmu = folded_grid(g)
    assert theory.ells == (1, 3) and np.allclose(theory.mu, mu, atol=1e-15) and np.allclose(theory.wmu, wmu, rtol=1e-13, atol=1e-16)
    names = [param.name for param in theory._all_params() if param.varied]
    assert set(names) == {'qpar', 'qper', 'dm', 'df', 'fnl_loc', 'p', 'b1', 'bv', 'sigmas', 'sigmau'} and 'sn0' not in theory._input_map()
    spec = theory._theory_spec()
    assert int(spec['png_velocity'][0]) == 1 and np.isclose(spec['png_velfac'][0], 100. / 1.5)
    with pytest.raises(ValueError): PNGTracerVelocityPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'), ells=(0, 2)).initialize()


@pytest.mark.gpu
def test_device_against_the_reference():
    from desilike_amd._lib import Context
    g = load()
    names = [str(n) for n in g['names']]
    mu, wmu = folded_grid(g)
    nk, nell = g['k'].size, 2

    def inp(name, default):
        return (names.index(name), default) if name in names else (-1, default)

    inputs = {'qpar': inp('qpar', 1.), 'qper': inp('qper', 1.), 'df': inp('df', 1.), 'dm': inp('dm', 0.), 'dn': inp('dn', 0.), 'b1X': inp('b1', 1.), 'b1Y': inp('b1', 1.),
              'fnl_loc': inp('fnl_loc', 0.), 'pX': inp('p', 1.), 'pY': inp('p', 1.), 'sigmas': inp('sigmas', 0.), 'sigmasY': inp('sigmas', 0.), 'bv': inp('bv', 1.), 'sigmau': inp('sigmau', 0.)}
    obs = dict(theory=np.array([5]), template=np.array([1]), apmode=np.array([0]), transform=np.array([0]), eta=[1. / 3.], f_fid=[float(g['f_fid'])], a=[float(g['a'])], kp=[float(g['kp'])], nd=[1.],
               ells_in=np.asarray(g['ells'], dtype='i4'), kin=g['k'], mu=mu, wmu_ell=wmu, k_t=g['k11'][1:], pk_dd_fid=g['pk_dd_fid'][1:], wmatrix=None, kmask=None, offset=None,
               flatdata=np.zeros(nell * nk), png_alpha=g['alpha_fid'][1:], png_mode=np.array([1], dtype='i4'), png_velocity=np.array([1], dtype='i4'), png_velfac=[100. / (1. + float(g['z']))],
               inputs=inputs)
    priors = np.array([[0., -1e3, 1e3, 0., 1.]] * len(names))
    ctx = Context(dict(n_params=np.array([len(names)]), priors=priors, precision=np.ones(nell * nk), observables=[obs]), device=0)
    power = ctx.eval_theory_host(g['theta'], iobs=0)
    scale = np.abs(g['power']).max()
    assert power.shape == g['power'].shape
    assert np.allclose(power, g['power'], rtol=1e-10, atol=1e-12 * scale), np.abs(power - g['power']).max() / scale
    ctx.close()
    # the host mirror compiles the same pipeline: its standalone multipoles against the oracle on the mirror's own fiducial
    from desilike_amd.theories.galaxy_clustering import PNGTracerVelocityPowerSpectrumMultipoles, ShapeFitPowerSpectrumTemplate
    theory = PNGTracerVelocityPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'), k=g['k'])
    row = dict(zip(names, g['theta'][0]))
    got = theory(**row).power
    template = theory.template
    m = dict(g)
    m.update(k11=np.concatenate([[template.k[0] / 2.], template.k]), pk_dd_fid=np.concatenate([[1.], template.pk_dd_fid]), alpha_fid=np.concatenate([[1.], theory.alpha_fid]),
             f_fid=template.f_fid, kp=template.kp, a=template.a)
    expected = oracle_power(m, g['theta'][0])
    assert np.allclose(got, expected, rtol=1e-10, atol=1e-12 * np.abs(expected).max())


@pytest.mark.gpu
def test_likelihood_on_the_odd_multipoles():
    """End to end: windowed observable of the odd multipoles, mock data from the theory, Gaussian likelihood -- zero at the truth, the batch equals the scalar calls."""
    from desilike_amd import vmap
    from desilike_amd.theories.galaxy_clustering import PNGTracerVelocityPowerSpectrumMultipoles, ShapeFitPowerSpectrumTemplate
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    theory = PNGTracerVelocityPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'))
    truth = {'b1': 2., 'bv': 1.1, 'fnl_loc': 20., 'sigmau': 3.}
    obs = TracerPowerSpectrumMultipolesObservable(data=truth, kedges=np.linspace(0.005, 0.105, 21), ells=(1, 3), wmatrix={'resolution': 2}, theory=theory)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=np.diag(np.full(40, 1e8)))
    like.initialize()
    assert like.flatdata.shape == (40,) and np.isfinite(like.flatdata).all() and np.abs(like.flatdata).max() > 0.
    assert abs(like(**truth)) < 1e-12 and like(**dict(truth, bv=1.3)) < -1e-3
    names = like.varied_params.names()
    rng = np.random.RandomState(1)
    theta = np.column_stack([np.clip(param.ref.sample(size=9, random_state=rng), *param.prior.limits) for param in like.varied_params])
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: theta[:, i] for i, name in enumerate(names)})
    for i in [0, 4, 8]:
        assert np.isclose(like(**dict(zip(names, theta[i]))), derived[like._param_loglikelihood][i], rtol=1e-12, atol=1e-12)
