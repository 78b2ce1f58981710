"""CPU: pin the restated analytic marginalisation (oracle.np_oracle.solve_marginalized, likelihoods/base.py:129-200, 314-413).

The reference's own implementation needs jax (absent here), so the pin is the Gaussian-integral identity against the reference's
NON-marginalised posterior evaluated on a grid of the linear parameter (fixture marg_sn0_grid.npz, generated from the reference):
log-posterior(sn0) is an exact parabola c - a (s - s*)^2 / 2; the reference's convention (396-401: no (2 pi)^(n/2)) gives
logposterior_marg = c - log(a) / 2."""
import numpy as np

from oracle import np_oracle as orc
from golden_utils import load_golden, observable_constants


def reference_parabola(g, ip):
    grid, lp = g['grid'], g['logposterior_grid'][ip]
    imax = np.argmax(lp)
    sl = slice(max(imax - 5, 0), imax + 6)
    x0 = grid[imax]
    c2, c1, c0 = np.polyfit(grid[sl] - x0, lp[sl] - lp[imax], 2)
    a = -2. * c2
    smax = x0 + c1 / a
    cmax = lp[imax] + c0 + 0.5 * c1**2 / a
    resid = np.abs(np.polyval([c2, c1, c0], grid[sl] - x0) - (lp[sl] - lp[imax])).max()
    return a, smax, cmax, resid


def test_marginalisation_vs_reference_grid():
    g = load_golden('marg_sn0_grid')
    c = observable_constants(g)
    names = [str(n) for n in g['names']]
    loc, scale = g['sn0_prior']
    for ip, row in enumerate(g['theta']):
        a, smax, cmax, resid = reference_parabola(g, ip)
        assert resid < 1e-7 * max(1., abs(cmax))    # the reference posterior is exactly Gaussian in sn0
        T = (g['flattheory_sn0_1'][ip] - g['flattheory_sn0_0'][ip])[None, :]
        flatdiff = g['flattheory_sn0_0'][ip] - c['flatdata']          # Delta at x0 = 0
        for marg in [True, False]:
            sol = orc.solve_marginalized(flatdiff, T, g['precision'], x0=[0.], prior_loc=[loc], prior_scale=[scale], marg_mask=[marg])
            total = sol['loglikelihood'] + sol['logprior_solved'] + g['logprior_others'][ip]
            expected = cmax - (0.5 * np.log(a) if marg else 0.)
            assert np.isclose(sol['x'][0], smax, rtol=1e-6, atol=1e-8)
            assert abs(total - expected) < 1e-6 * max(1., abs(expected)), (ip, marg, total, expected)
        # Delta evaluated at another x0 gives the same answer (linearity)
        sol2 = orc.solve_marginalized(flatdiff + 0.3 * T[0], T, g['precision'], x0=[0.3], prior_loc=[loc], prior_scale=[scale], marg_mask=[True])
        assert np.isclose(sol2['loglikelihood'] + sol2['logprior_solved'], sol['loglikelihood'] * 0 + orc.solve_marginalized(flatdiff, T, g['precision'], [0.], [loc], [scale], [True])['loglikelihood']
                          + orc.solve_marginalized(flatdiff, T, g['precision'], [0.], [loc], [scale], [True])['logprior_solved'], rtol=1e-11, atol=1e-9)


def test_marginalisation_closed_form_flat_prior():
    # flat prior, diagonal precision, two solved parameters: chi2 at the best fit + log det(T P T^T)
    rng = np.random.RandomState(0)
    n = 30
    T = rng.standard_normal((2, n))
    d = rng.standard_normal(n) * 3.
    prec = rng.uniform(0.5, 2., n)
    sol = orc.solve_marginalized(d, T, prec, x0=[0., 0.], prior_loc=[0., 0.], prior_scale=[np.inf, np.inf], marg_mask=[True, True])
    F = (T * prec).dot(T.T)
    xbest = -np.linalg.solve(F, (T * prec).dot(d))
    resid = d + xbest.dot(T)
    assert np.allclose(sol['x'], xbest, rtol=1e-12)
    assert np.isclose(sol['loglikelihood'], -0.5 * (resid * prec).dot(resid) - 0.5 * np.linalg.slogdet(F)[1], rtol=1e-12)
    assert sol['logprior_solved'] == 0.
    # '.prec': marginalised precision gives the same chi2 at fixed x = 0 (likelihoods/base.py:280-309)
    P2 = orc.marginalize_precision(prec, T, [np.inf, np.inf])
    assert np.isclose(-0.5 * d.dot(P2).dot(d), -0.5 * (resid * prec).dot(resid), rtol=1e-10)
