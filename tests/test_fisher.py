"""Fisher algebra (fisher.py:731-750, 216-257): CPU test of the restatement; GPU test of the batched finite-difference driver."""
import numpy as np
import pytest

from oracle import np_oracle as orc


def test_fisher_algebra_linear_model():
    # linear model: Fisher is exact; mean = generalised least squares solution, chi2min = residual chi2
    rng = np.random.RandomState(0)
    n, p = 40, 3
    A = rng.standard_normal((p, n))
    data = rng.standard_normal(n) * 2.
    C = rng.standard_normal((n, n)); prec = np.linalg.inv(C.dot(C.T) + n * np.eye(n))
    center = np.array([0.3, -0.2, 1.])
    flatdiff = center.dot(A) - data
    offset, gradient, hessian = orc.fisher_gaussian(flatdiff, A, prec)
    assert np.isclose(offset, -flatdiff.dot(prec).dot(flatdiff))          # no 1/2: fisher.py:746
    mean, chi2min = orc.fisher_mean_chi2min(center, offset, gradient, hessian)
    gls = np.linalg.solve(A.dot(prec).dot(A.T), A.dot(prec).dot(data))
    assert np.allclose(mean, gls, rtol=1e-10)
    # with the reference's convention chi2min = -2 (offset + ...) uses offset = -chi2 (not -chi2 / 2): reproduce as is
    resid = gls.dot(A) - data
    d = -np.linalg.solve(hessian, gradient)
    assert np.isclose(chi2min, -2. * (offset + gradient.dot(d) + 0.5 * d.dot(hessian).dot(d)))
    from desilike_amd.fisher import LikelihoodFisher
    fisher = LikelihoodFisher(center, ['a', 'b', 'c'], offset, gradient, hessian)
    assert np.allclose(fisher.mean(), mean) and np.isclose(fisher.chi2min, chi2min)
    assert np.allclose(fisher.covariance(), np.linalg.inv(A.dot(prec).dot(A.T)))


def _oracle_fisher(c, names, precision, center, steps):
    """fisher.py:739-748 on vectors computed by the ORACLE (the restatement pinned on the reference's outputs of this fixture): flattheory on the stencil,
    central differences, then the reference's algebra."""
    def flattheory(row):
        p = dict(zip(names, row)); p['b1'] = (p['b1'], p['b1'])
        return orc.fullshape_observable(c, p)['flattheory']

    flatdiff = flattheory(center) - c['flatdata']
    flatderiv = []
    for ip in range(len(names)):
        up, dn = center.copy(), center.copy()
        up[ip] += steps[ip, 1]; dn[ip] -= steps[ip, 0]
        flatderiv.append((flattheory(up) - flattheory(dn)) / (steps[ip, 0] + steps[ip, 1]))
    return orc.fisher_gaussian(flatdiff, np.array(flatderiv), precision)


@pytest.mark.gpu
def test_fisher_kernel_vs_oracle():
    """dl_eval_fisher (stencil through the theory kernels + whitened GEMM, MFMA Gram product per centre) against the oracle's vectors and algebra."""
    from test_host_api import make_cfg2
    from golden_utils import observable_constants
    from desilike_amd.fisher import Fisher
    g, like = make_cfg2(dense=False)
    fisher = Fisher(like)
    names = like.varied_params.names()
    c = observable_constants(g)
    rng = np.random.RandomState(5)
    centers = np.array([[1.01, 0.995, 0.01, 1.02, 1.9, 0.1], [1., 1., 0., 1., 2., 0.]] + [[param.ref.sample(random_state=rng) for param in like.varied_params] for _ in range(3)])
    centers[4, 0] = like.varied_params['qpar'].prior.limits[1] - 1e-4          # next to a prior bound: the upper step is shortened
    steps = fisher.steps(centers)
    assert steps[4, 0, 1] == pytest.approx(1e-4) and (steps > 0.).all()
    offset, gradient, hessian = fisher.evaluate(centers)
    for ib, center in enumerate(centers):
        ref = _oracle_fisher(c, names, like.precision, center, steps[ib])
        assert np.isclose(offset[ib], ref[0], rtol=1e-10, atol=1e-10), ib
        # (at the best fit -- centre 1 -- the oracle's gradient is exactly 0: the scale of the comparison is |dD| |D| ~ sqrt(|hessian| |offset|), floored)
        gscale = max(np.abs(ref[1]).max(), 0.1 * np.sqrt(np.abs(ref[2]).max()))
        assert np.allclose(gradient[ib], ref[1], rtol=1e-8, atol=1e-8 * gscale), ib
        assert np.allclose(hessian[ib], ref[2], rtol=1e-8, atol=1e-8 * np.abs(ref[2]).max()), ib
        assert np.array_equal(hessian[ib], hessian[ib].T)
    # offset = -D P D = 2 logL (no 1/2, fisher.py:746); a batch of one centre gives the same bits as the batch of five
    ctx = like._get_context()
    assert np.allclose(offset, 2. * ctx.eval_batch_host(centers)[0], rtol=1e-10, atol=1e-9)
    single = fisher.evaluate(centers[2])
    assert np.array_equal(single[2][0], hessian[2]) and np.array_equal(single[1][0], gradient[2])
    # the call surface: data generated at b1 = 2 -> the centre is the best fit: mean = centre, chi2min = 0, positive-definite precision
    result = fisher(qpar=1., qper=1., dm=0., df=1., b1=2., sn0=0.)
    assert np.abs(result.mean() - centers[1]).max() < 1e-6 and abs(result.chi2min) < 1e-8
    assert (np.linalg.eigvalsh(result.precision()) > 0).all()
    # sn0 has a Gaussian prior (scale 1000): its precision adds to the diagonal (fisher.py:712-714)
    assert np.isclose(result.precision()[5, 5] - fisher.likelihood_fisher.precision()[5, 5], 1e-6, rtol=1e-9)
    # the likelihood gradient is the derivative of logL: differences of the GPU loglikelihood
    x0 = centers[0]
    eps = np.array([1e-4, 1e-4, 1e-3, 1e-3, 1e-3, 1e-2])
    grad = np.zeros(6)
    for i in range(6):
        up, dn = x0.copy(), x0.copy()
        up[i] += eps[i]; dn[i] -= eps[i]
        ll = ctx.eval_batch_host(np.array([up, dn]))[0]
        grad[i] = (ll[0] - ll[1]) / (2 * eps[i])
    assert np.allclose(gradient[0], grad, rtol=5e-3, atol=1e-4 * np.abs(grad).max())


@pytest.mark.gpu
def test_fisher_varies_solved_parameters():
    """Analytically solved parameters are varied in the Fisher estimate, like the reference (fisher.py:688-695: warning + derived=False on a copy)."""
    from golden_utils import load_golden, observable_constants
    from test_gpu_marg import make_marg_likelihood
    from desilike_amd.fisher import Fisher
    g = load_golden('marg_sn0_grid')
    like = make_marg_likelihood(g, solved='.marg')
    with pytest.warns(UserWarning, match='solved parameters'):
        fisher = Fisher(like)
    names = fisher.varied_params.names()
    assert names[-1] == 'sn0' and 'sn0' not in like.varied_params.names()
    c = observable_constants(g)
    center = np.array([1.005, 0.998, 0.004, 1.01, 1.95, 0.3])
    steps = fisher.steps(center)
    offset, gradient, hessian = fisher.evaluate(center)
    ref = _oracle_fisher(c, names, like.precision, center, steps[0])
    assert np.isclose(offset[0], ref[0], rtol=1e-10) and np.allclose(gradient[0], ref[1], rtol=1e-8, atol=1e-8 * np.abs(ref[1]).max())
    assert np.allclose(hessian[0], ref[2], rtol=1e-8, atol=1e-8 * np.abs(ref[2]).max())
    result = fisher(**dict(zip(names, center)))
    prior = like.all_params['sn0'].prior
    assert np.isclose(result.precision()[5, 5] - fisher.likelihood_fisher.precision()[5, 5], prior.scale**-2, rtol=1e-9)


@pytest.mark.gpu
def test_logposterior_value_and_grad():
    """Batched finite-difference gradient (the value_and_grad gradient-based samplers need, samplers/hmc.py:194): against the oracle's log-posterior on the same stencil,
    and against the analytic gradient along a parameter the model is linear in (sn0)."""
    from desilike_amd.fisher import logposterior_value_and_grad
    from golden_utils import load_golden, observable_constants, prior_list
    from test_gpu_marg import make_marg_likelihood
    from oracle import np_oracle as orc
    g = load_golden('marg_sn0_grid')
    like = make_marg_likelihood(g, solved=False)
    names = like.varied_params.names()
    rnames = [str(n) for n in g['names']]
    theta = g['theta'][:6][:, [rnames.index(n) for n in names]]
    value, grad = logposterior_value_and_grad(like, theta, method='finite')
    c = observable_constants(g)
    priors = [dict(dist=param.prior.dist, limits=param.prior.limits, loc=getattr(param.prior, 'loc', 0.) if param.prior.dist == 'norm' else 0.,
                   scale=getattr(param.prior, 'scale', 1.) if param.prior.dist == 'norm' else 1.) for param in like.varied_params]

    def logpost(row):
        p = dict(zip(names, row)); p['b1'] = (p['b1'], p['b1'])
        return orc.gaussian_loglikelihood(orc.fullshape_observable(c, p)['flattheory'], c['flatdata'], like.precision)[0] + orc.logprior(row, priors)

    for i, row in enumerate(theta):
        assert abs(value[i] - logpost(row)) <= 1e-10 * max(1., abs(value[i]))
        for ip, param in enumerate(like.varied_params):
            _, lo, hi = param.delta
            lo, hi = min(lo, row[ip] - param.prior.limits[0]), min(hi, param.prior.limits[1] - row[ip])
            up, dn = row.copy(), row.copy()
            up[ip] += hi; dn[ip] -= lo
            ref = (logpost(up) - logpost(dn)) / (lo + hi)
            assert abs(grad[i, ip] - ref) <= 1e-6 * max(1., abs(ref)), (i, param.name, grad[i, ip], ref)
    # analytic: d logL / d sn0 = -T P Delta (T = d flattheory / d sn0, constant), + Gaussian prior term
    isn = names.index('sn0')
    p0 = dict(zip(names, theta[0])); p0['b1'] = (p0['b1'], p0['b1'])
    f0 = orc.fullshape_observable(c, p0)['flattheory']
    T = orc.fullshape_observable(c, dict(p0, sn0=p0['sn0'] + 1.))['flattheory'] - f0
    prior = like.varied_params['sn0'].prior
    analytic = -T.dot(like.precision).dot(f0 - c['flatdata']) - (theta[0, isn] - prior.loc) / prior.scale**2
    assert abs(grad[0, isn] - analytic) <= 1e-7 * max(1., abs(analytic))


@pytest.mark.gpu
def test_analytic_gradient_vs_oracle_and_finite_differences():
    """dl_eval_logposterior_grad (the value_and_grad of the gradient-based samplers): against the five-point stencil of the NumPy ORACLE's
    log-posterior (1e-8 of the largest component), against the library's own central differences, and for two tracers sharing template parameters."""
    import torch
    from desilike_amd.fisher import logposterior_value_and_grad
    from golden_utils import load_golden, observable_constants, prior_list
    from oracle import np_oracle as orc
    from test_host_api import make_cfg2, make_cfg5
    g, like = make_cfg2()
    names = like.varied_params.names()
    theta = g['theta'][:6, [[str(n) for n in g['names']].index(name) for name in names]]
    value, grad = logposterior_value_and_grad(like, theta, method='analytic')
    c, priors = observable_constants(g), prior_list(g)

    def oracle(th):
        out = []
        for row in th:
            p = dict(zip(names, row)); p['b1'] = (p['b1'], p['b1'])
            out.append(orc.gaussian_loglikelihood(orc.fullshape_observable(c, p)['flattheory'], c['flatdata'], like.precision)[0])
        return np.array(out) + orc.logprior(th, [priors[[str(n) for n in g['names']].index(name)] for name in names])

    assert (np.abs(value - oracle(theta)) <= 1e-10 * np.maximum(1., np.abs(value))).all()
    fd = np.zeros_like(grad)
    for p in range(theta.shape[1]):
        h = 1e-3

        def f(x):
            th = theta.copy(); th[:, p] += x
            return oracle(th)

        fd[:, p] = (-f(2 * h) + 8 * f(h) - 8 * f(-h) + f(-2 * h)) / (12 * h)
    assert (np.abs(grad - fd).max(axis=0) <= 1e-8 * np.abs(fd).max(axis=0)).all(), np.abs(grad - fd).max(axis=0) / np.abs(fd).max(axis=0)
    value_fd, grad_fd = logposterior_value_and_grad(like, theta, method='finite')
    assert np.array_equal(np.isfinite(value), np.isfinite(value_fd)) and np.allclose(value, value_fd, rtol=1e-12, atol=1e-9)
    assert np.allclose(grad, grad_fd, rtol=1e-3, atol=1e-5 * np.abs(grad).max())        # (central differences with the parameters' own, large, steps: O(h^2) off)
    # rows outside the prior: -inf and a zero gradient; NaN inputs likewise
    bad = theta[:2].copy(); bad[0, 0] = 5.; bad[1, 1] = np.nan
    value, grad = logposterior_value_and_grad(like, bad, method='analytic')
    assert np.isneginf(value).all() and (grad == 0.).all()
    # two tracers (shared qpar, qper, dm, df; namespaced b1, sn0): the columns shared by both observables receive both contributions
    g5, like5 = make_cfg5()
    names5 = like5.varied_params.names()
    theta5 = g5['theta'][:4, [[str(n) for n in g5['names']].index(name) for name in names5]]
    value5, grad5 = logposterior_value_and_grad(like5, theta5, method='analytic')
    _, grad5_fd = logposterior_value_and_grad(like5, theta5, method='finite')
    assert np.allclose(grad5, grad5_fd, rtol=1e-3, atol=1e-5 * np.abs(grad5).max())
    ref = g5['loglikelihood'][:4] + g5['logprior'][:4]
    assert (np.abs(value5 - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all()
    # out of scope -> the caller is told (and 'auto' falls back to central differences)
    from test_host_api import make_cfg4
    g4, like4 = make_cfg4('pk')
    with pytest.raises(NotImplementedError):
        logposterior_value_and_grad(like4, g4['theta'][:2, [[str(n) for n in g4['names']].index(name) for name in like4.varied_params.names()]], method='analytic')
