"""Damped BAO with the kernel broadbands of the reference (bao.py:43-60, 468-523, 833-905): 'pcs' for P_ell, 'pcs2' for xi_ell (Fourier-space kernels taken
through the Hankel transform + powers of s).  Fixtures from the reference (tests/golden/make_golden.py cfg4_pcs).
CPU: host-built broadband matrices and the oracle chain; GPU (-m gpu): the call surface (broadband terms as pass-through columns), marginalised broadband."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, prior_list


def make_likelihood(space, data=None):
    from desilike_amd.theories.galaxy_clustering import BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles, DampedBAOWigglesTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable, TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg4_bao_{}_pcs'.format(space))
    template = BAOPowerSpectrumTemplate(z=0.5)
    data = g['obs0']['flatdata'] if data is None else data
    if space == 'xi':
        theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='recsym', broadband='pcs2')
        obs = TracerCorrelationFunctionMultipolesObservable(data=data, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
    else:
        theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template, broadband='pcs')
        obs = TracerPowerSpectrumMultipolesObservable(data=data, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
    return g, theory, ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])


@pytest.mark.parametrize('space', ['xi', 'pk'])
def test_kernel_broadband_matrices_and_oracle_chain(space):
    g, theory, like = make_likelihood(space)
    c = g['obs0']
    names = [str(n) for n in g['names']]
    assert like.varied_params.names() == names
    knames = [str(n) for n in c['kernel_params']]
    nk = len(knames)
    if space == 'xi':
        assert theory._broadband_names == knames + [str(n) for n in c['s_params']]
        kmat = theory._fourier_broadband
        assert np.allclose(theory._s_broadband, c['s_matrix'].reshape(-1, len(c['s_params'])), rtol=1e-13, atol=0.)
    else:
        assert theory._broadband_names == knames
        kmat = theory.broadband_matrix
    assert np.allclose(theory.kin, c['kin'], rtol=1e-14) and np.isclose(theory.kp, c['kp'], rtol=1e-13)
    assert np.allclose(kmat, c['kernel_matrix'].reshape(-1, nk), rtol=1e-11, atol=1e-14 * np.abs(c['kernel_matrix']).max())
    # oracle chain on the reference's own intermediate (wiggle multipoles): + kernels, Hankel / window, + powers of s
    priors = prior_list(g)
    for i, row in enumerate(g['theta']):
        p = dict(zip(names, row))
        power = g['wiggle_power'][i] + c['kernel_matrix'].dot(np.array([p.get(n, 0.) for n in knames]))
        if space == 'xi':
            theo = orc.get_corr(power, c['kin'], c['s'], (0, 2)) + c['s_matrix'].dot(np.array([p[str(n)] for n in c['s_params']]))
            flat = np.ravel(theo)
        else:
            theo = power
            flat = orc.window_apply(power, matrix_full=c['matrix_full'], shotnoisein=c['shotnoisein'], shotnoiseout=c['shotnoiseout'])
        assert np.allclose(theo, g['theory'][i], rtol=1e-11, atol=1e-13 * np.abs(g['theory'][i]).max())
        logl = orc.gaussian_loglikelihood(flat, c['flatdata'], g['precision'])[0]
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))
        assert np.isclose(orc.logprior(row, priors), g['logprior'][i], rtol=1e-13, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize('space', ['xi', 'pk'])
def test_kernel_broadband_call_surface_vs_reference(space):
    from desilike_amd import vmap
    g, theory, like = make_likelihood(space)
    names = [str(n) for n in g['names']]
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert errors == {}
    assert (np.abs(derived['loglikelihood'] - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all()
    assert np.allclose(derived['logprior'], g['logprior'], rtol=1e-13, atol=1e-13)
    flat = like._get_context().eval_batch_host(g['theta'], return_flattheory=True)[3]
    assert np.allclose(flat, g['flattheory'], rtol=1e-9, atol=1e-12 * np.abs(g['flattheory']).max())
    g2, theory2, like2 = make_likelihood(space, data={'b1': 2., 'sigmas': 2.})
    assert abs(like2(b1=2., sigmas=2.)) < 1e-12


@pytest.mark.gpu
def test_xi_kernel_broadband_marginalised():
    """The DESI-style fit: every broadband term of the 'pcs2' correlation function model solved analytically."""
    g, theory, like = make_likelihood('xi')
    like.initialize()
    for param in theory.init.params.select(basename=['al*', 'bl*']):
        if param.varied: param.update(derived='.marg')
    like._invalidate()
    names = [str(n) for n in g['names']]
    solved, vnames = like.solved_params.names(), like.varied_params.names()
    assert solved == ['al2_0', 'al2_1', 'bl0_0', 'bl0_2', 'bl2_0', 'bl2_2']
    sub = g['theta'][:, [names.index(n) for n in vnames]]
    loglike, logprior, status, xs = like._get_context().eval_batch_host(sub, return_solved=True)
    assert (status == 0).all()
    c = g['obs0']
    knames, snames = [str(n) for n in c['kernel_params']], [str(n) for n in c['s_params']]
    kmat = c['kernel_matrix'].reshape(-1, len(knames))
    # derivative rows through the oracle: Hankel transform of the Fourier kernel of each solved al*, then the powers of s of each bl*
    Tk = np.array([np.ravel(orc.get_corr(kmat[:, knames.index(name)].reshape(2, -1), c['kin'], c['s'], (0, 2))) for name in solved if name.startswith('al')])
    Ts = np.array([c['s_matrix'].reshape(-1, len(snames))[:, snames.index(name)] for name in solved if name.startswith('bl')])
    T = np.vstack([Tk, Ts])
    for i in range(len(sub)):
        f0 = np.ravel(orc.get_corr(g['wiggle_power'][i], c['kin'], c['s'], (0, 2)))
        sol = orc.solve_marginalized(f0 - c['flatdata'], T, like.precision, x0=np.zeros(6), prior_loc=np.zeros(6), prior_scale=np.full(6, np.inf), marg_mask=np.ones(6, dtype='?'))
        assert abs(loglike[i] - sol['loglikelihood']) <= 1e-8 * max(1., abs(sol['loglikelihood'])), (loglike[i], sol['loglikelihood'])
