"""The batched Levenberg-Marquardt maximiser of desilike_amd/profilers.py: CPU tests of the iteration on analytic least-squares problems (the device supplies value,
gradient and Gauss-Newton curvature; here NumPy does), GPU tests on likelihoods whose maximum is known exactly (data generated from the theory itself)."""
import numpy as np
import pytest


def _least_squares(residual, jacobian):
    def evaluate(points):
        f, g, H = [], [], []
        for x in points:
            r, J = residual(x), jacobian(x)
            f.append(-0.5 * r.dot(r)); g.append(-J.T.dot(r)); H.append(J.T.dot(J))
        return np.array(f), np.array(g), np.array(H)
    return evaluate


def test_levenberg_marquardt_on_analytic_problems():
    from desilike_amd.profilers import levenberg_marquardt
    # Rosenbrock as a least-squares problem: maximum of f = -1/2 |r|^2 at (1, 1), from several starts at once
    evaluate = _least_squares(lambda x: np.array([10. * (x[1] - x[0]**2), 1. - x[0]]), lambda x: np.array([[-20. * x[0], 10.], [-1., 0.]]))
    start = np.array([[-1.2, 1.], [2., -1.], [0., 0.], [3., 3.]])
    x, f, g, H, iterations, converged = levenberg_marquardt(evaluate, start, np.full(2, -np.inf), np.full(2, np.inf), max_iterations=200)
    assert converged.all() and np.allclose(x, 1., atol=1e-6) and np.all(f > -1e-12) and iterations.max() < 60
    # exponential fit with bounds: the unconstrained optimum lies outside, the iteration ends on the bound
    t = np.linspace(0., 2., 30)
    data = 2. * np.exp(-1.5 * t)
    evaluate = _least_squares(lambda x: x[0] * np.exp(-x[1] * t) - data, lambda x: np.column_stack([np.exp(-x[1] * t), -x[0] * t * np.exp(-x[1] * t)]))
    x, f, g, H, iterations, converged = levenberg_marquardt(evaluate, np.array([[1., 1.], [3., 0.5]]), np.array([0., 0.]), np.array([10., 10.]), max_iterations=200)
    assert np.allclose(x, [2., 1.5], atol=1e-6)
    x, f, g, H, iterations, converged = levenberg_marquardt(evaluate, np.array([[1., 0.5]]), np.array([0., 0.]), np.array([10., 1.2]), max_iterations=200)
    assert x[0, 1] <= 1.2 and abs(x[0, 1] - 1.2) < 1e-6 and np.isfinite(f).all()
    with pytest.raises(ValueError):
        levenberg_marquardt(lambda p: (np.full(len(p), np.nan), np.zeros((len(p), 2)), np.tile(np.eye(2), (len(p), 1, 1))), np.zeros((1, 2)), np.full(2, -1.), np.full(2, 1.))


def test_profile_of_a_gaussian_is_its_marginal_parabola():
    """Coordinates held fixed (profiles along a parameter): the maximum over the others of a correlated Gaussian is the parabola of the marginal variance."""
    from desilike_amd.profilers import levenberg_marquardt
    cov = np.array([[1., 0.8, 0.1], [0.8, 2., 0.3], [0.1, 0.3, 0.5]])
    prec, mean = np.linalg.inv(cov), np.array([0.3, -1., 2.])

    def evaluate(points):
        d = points - mean
        return -0.5 * np.einsum('ij,jk,ik->i', d, prec, d), -d.dot(prec), np.tile(prec, (len(points), 1, 1))

    grid = np.linspace(-2., 2., 9)
    start = np.tile(mean + 0.2, (9, 1)); start[:, 0] = mean[0] + grid
    free = np.ones((9, 3), dtype='?'); free[:, 0] = False
    x, f, g, H, iterations, converged = levenberg_marquardt(evaluate, start, np.full(3, -np.inf), np.full(3, np.inf), free=free)
    assert converged.all() and np.array_equal(x[:, 0], start[:, 0])
    assert np.allclose(f, -0.5 * grid**2 / cov[0, 0], atol=1e-10)
    assert np.allclose(x[:, 1:], mean[1:] + np.outer(grid, cov[1:, 0] / cov[0, 0]), atol=1e-6)      # the conditional mean: the ridge


@pytest.mark.gpu
def test_maximum_of_a_likelihood_generated_from_its_theory():
    """config-5 shape (two tracers, 8 parameters), data = theory at known parameters, flat priors: the posterior maximum is there, chi2 = 0."""
    from test_host_api import make_cfg5
    from desilike_amd.profilers import GaussNewtonProfiler, Profiles
    g, like = make_cfg5()
    names = like.varied_params.names()
    truth = np.array([param.value for param in like.varied_params], dtype='f8')
    truth = truth * (1. + 0.01 * np.random.RandomState(0).standard_normal(truth.size))
    # replace the data by the theory at `truth`
    ctx = like._get_context()
    flat = ctx.eval_batch_host(truth[None, :], return_flattheory=True)[3][0]
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    sizes = [obs.wmatrix.size for obs in like.observables]
    start = 0
    for obs, n in zip(like.observables, sizes):
        obs.flatdata = flat[start:start + n].copy(); start += n
    like2 = ObservablesGaussianLikelihood(observables=like.observables, covariance=like.covariance)
    profiler = GaussNewtonProfiler(like2, seed=3)
    profiles = profiler.maximize(niterations=6)
    assert isinstance(profiles, Profiles) and profiles.bestfit['logposterior'].shape == (6,)
    best = profiles.choice()
    logprior_at_truth = sum(float(param.prior(np.array([value]))[0]) for param, value in zip(like2.varied_params, truth))
    assert abs(best['logposterior'] - logprior_at_truth) < 1e-6, (best['logposterior'], logprior_at_truth)
    errors = np.array([profiles.error[name][profiles.argmax()] for name in names])
    got = np.array([best[name] for name in names])
    assert np.all(np.abs(got - truth) < 1e-3 * errors), (got - truth) / errors
    # most starts end at the same maximum
    assert np.sum(np.abs(profiles.bestfit['logposterior'] - best['logposterior']) < 1e-5) >= 4
    assert all(profiles.attrs['converged'][i] for i in [profiles.argmax()])
    # errors = the Fisher matrix at the best fit (+ prior curvature)
    from desilike_amd.fisher import Fisher
    fisher = Fisher(like2)(**{name: best[name] for name in names})
    assert np.allclose(np.sqrt(np.diag(np.linalg.inv(-fisher._hessian))), errors, rtol=1e-5)
    # profiles along two parameters and their intervals: near the maximum the posterior is Gaussian in the data, -2 Delta log L = (Delta x / sigma)^2
    tables = profiler.profile(params=['LRG.b1', 'qpar'], size=9, cl=1.5)
    for name in ['LRG.b1', 'qpar']:
        values, logp = tables[name][:, 0], tables[name][:, 1]
        sigma = profiles.error[name][profiles.argmax()]
        expected = best['logposterior'] - 0.5 * ((values - best[name]) / sigma)**2
        assert np.all(logp <= best['logposterior'] + 1e-7)
        inner = np.abs(values - best[name]) <= 0.4 * sigma
        assert inner.sum() >= 3 and np.allclose(logp[inner], expected[inner], rtol=0., atol=0.25 * 0.5 * 0.4**2)       # the Gaussian parabola close to the maximum
        assert np.all(np.diff(logp[values <= best[name]]) > 0.) and np.all(np.diff(logp[values >= best[name]]) < 0.)  # one maximum
        # maximised over the others >= the slice with the others kept at the best fit
        points = np.tile(got, (len(values), 1)); points[:, names.index(name)] = values
        assert np.all(logp >= profiler.logposterior(points) - 1e-8)
    intervals = profiler.interval(params=['LRG.b1'], cl=1., size=13)
    lo, hi = intervals['LRG.b1']
    sigma = profiles.error['LRG.b1'][profiles.argmax()]
    assert -2. * sigma < lo < -0.5 * sigma and 0.5 * sigma < hi < 2. * sigma, (lo, hi, sigma)
    # a two-parameter grid: nothing above the maximum, the highest cell is the one next to the best fit, its maxima along one axis are the 1D profile of the other
    axes, surface = profiler.grid(['LRG.b1', 'qpar'], size=5, cl=1.)
    nearest = tuple(int(np.argmin(np.abs(axis - best[name]))) for axis, name in zip(axes, ['LRG.b1', 'qpar']))
    assert surface.shape == (5, 5) and np.all(surface <= best['logposterior'] + 1e-7) and tuple(int(i) for i in np.unravel_index(np.argmax(surface), surface.shape)) == nearest
    line = profiler.profile(params=['LRG.b1'], size=5, cl=1.)['LRG.b1']
    assert np.allclose(axes[0], line[:, 0]) and np.allclose(surface.max(axis=1), line[:, 1], atol=0.06)      # (a grid of 5 values of the other parameter against its continuum)
    # the profiles seed a Metropolis-Hastings sampler's proposal
    from desilike_amd.samplers import MCMCSampler
    sampler = MCMCSampler(like2, chains=4, covariance=profiles, seed=1, learn=False)
    assert np.allclose(sampler.covariance, profiles.covariance[1], rtol=1e-12)
    sampler.run(check_every=100, max_iterations=100, start=np.tile(got, (4, 1)))
    assert 0.02 < np.nanmean(sampler.acceptance_rate) < 0.7      # (several parameters are limited by their uniform priors: the Gaussian width overshoots there)
