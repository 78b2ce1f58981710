"""Posterior maximisation on the device: all starting points advance together, each iteration is ONE batch.

The reference's profilers (desilike/profilers/: ``MinuitProfiler``, ``ScipyProfiler``, ``BOBYQAProfiler``, ``OptaxProfiler``) hand ``-logposterior`` to third-party
minimisers, one evaluation at a time per start (profilers/base.py: the starts are spread over MPI ranks).  The Gaussian likelihoods of this package expose more than
values: ``dl_eval_fisher`` returns, for a whole batch of centres, the log-likelihood, its gradient and the Gauss-Newton curvature ``-dD P dD^T`` (csrc/dl_fisher.hip:
the finite-difference stencil of every centre through the theory kernels and the whitened window product as one launch sequence).  :class:`GaussNewtonProfiler` runs
Levenberg-Marquardt on that: per iteration one Fisher batch over the candidates of ALL starts, a P x P solve per start on the host, acceptance per start.

``maximize`` returns :class:`Profiles` with the reference's attribute names (``bestfit`` incl. ``logposterior``, ``error``, ``covariance``, ``start``;
samples/profiles.py); errors and covariance are the inverse Gauss-Newton curvature (+ prior curvature) at the best fit -- what ``Fisher`` gives at that point.
"""
import numpy as np

from .fisher import Fisher
from .samplers import BasePosteriorSampler


class Profiles(object):
    """Result of a maximisation (the subset of desilike/samples/profiles.py the maximiser fills): ``start`` / ``bestfit``: dict name -> array [nstarts]
    (``bestfit['logposterior']`` included), ``error``: dict name -> array [nstarts], ``covariance``: (names, matrix [P, P]) at the best of the starts."""

    def __init__(self, start, bestfit, error, covariance, attrs=None):
        self.start, self.bestfit, self.error, self.covariance, self.attrs = start, bestfit, error, covariance, dict(attrs or {})

    def argmax(self):
        return int(np.argmax(self.bestfit['logposterior']))

    def choice(self, index='argmax'):
        """Best-fit values of one start (default: the best) as dict name -> float (samples/profiles.py ``ParameterBestFit.choice``)."""
        index = self.argmax() if index == 'argmax' else int(index)
        return {name: float(values[index]) for name, values in self.bestfit.items()}

    def to_dict(self):
        return {'start': self.start, 'bestfit': self.bestfit, 'error': self.error, 'covariance': self.covariance, 'attrs': self.attrs}

    def save(self, filename):
        np.save(filename, self.to_dict(), allow_pickle=True)

    @classmethod
    def load(cls, filename):
        state = np.load(filename, allow_pickle=True)[()]
        return cls(state['start'], state['bestfit'], state['error'], tuple(state['covariance']), attrs=state.get('attrs', None))


def levenberg_marquardt(evaluate, start, lower, upper, max_iterations=100, xtol=1e-7, ftol=1e-9, damping=1e-3, free=None):
    """Maximise f over a batch of starts.  ``evaluate(points [S, P]) -> (f [S], gradient [S, P], curvature [S, P, P])`` with ``curvature`` positive semi-definite
    (minus the Gauss-Newton Hessian); ``lower`` / ``upper``: bounds [P].  Per iteration ONE call of ``evaluate`` on the candidates of the starts still running: a
    candidate that improves f is taken (its derivatives are already there) and the damping relaxed, otherwise the damping grows and the step is redone from the kept
    derivatives.  Stops a start when an accepted step is below ``xtol`` in units of the curvature's standard deviations or improves f by less than ``ftol``.
    ``free``: boolean mask [P] (or [S, P]) of the coordinates that move; the others keep their starting values (profiles along a parameter).
    Returns (points, f, gradient, curvature, iterations [S], converged [S])."""
    x = np.array(start, dtype='f8')
    S, P = x.shape
    free = np.ones((S, P), dtype='?') if free is None else np.broadcast_to(np.asarray(free, dtype='?'), (S, P))
    f, g, H = evaluate(x)
    f, g, H = np.array(f, dtype='f8'), np.array(g, dtype='f8'), np.array(H, dtype='f8')
    if not np.isfinite(f).all(): raise ValueError('the objective is not finite at a starting point')
    lam = np.full(S, float(damping))
    active, converged, iterations = np.ones(S, dtype='?'), np.zeros(S, dtype='?'), np.zeros(S, dtype='i8')
    lower, upper = np.asarray(lower, dtype='f8'), np.asarray(upper, dtype='f8')
    width = np.where(np.isfinite(lower) & np.isfinite(upper), np.abs(np.where(np.isfinite(upper), upper, 0.) - np.where(np.isfinite(lower), lower, 0.)), 1.)
    eps = 1e-12 * np.maximum(1., width)
    lo, hi = np.where(np.isfinite(lower), lower + eps, -np.inf), np.where(np.isfinite(upper), upper - eps, np.inf)
    for iteration in range(max_iterations):
        index = np.flatnonzero(active)
        if not index.size: break
        candidates, scaled = np.empty((index.size, P)), np.empty(index.size)
        for slot, s in enumerate(index):
            diag = np.maximum(np.diag(H[s]), 1e-300)
            # moving coordinates: the free ones, minus those that sit on a bound with the gradient pointing outwards (active set: solving for them and clipping the
            # step would distort the step of the others and stall the iteration against the bound)
            m = free[s] & ~(((x[s] <= lo) & (g[s] < 0.)) | ((x[s] >= hi) & (g[s] > 0.)))
            if not m.any():
                candidates[slot], scaled[slot] = x[s], 0.
                continue
            step = np.zeros(P)
            step[m] = np.linalg.solve((H[s] + lam[s] * np.diag(diag))[np.ix_(m, m)], g[s][m])      # (H + lam diag H) step = gradient, in the moving coordinates
            candidates[slot] = np.where(m, np.clip(x[s] + step, lo, hi), x[s])
            scaled[slot] = np.max(np.abs(candidates[slot] - x[s]) * np.sqrt(diag))   # step in units of the (conditional) standard deviations
        fc, gc, Hc = evaluate(candidates)
        iterations[index] += 1
        for slot, s in enumerate(index):
            if np.isfinite(fc[slot]) and fc[slot] >= f[s]:
                gain = fc[slot] - f[s]
                x[s], f[s], g[s], H[s] = candidates[slot], fc[slot], gc[slot], Hc[slot]
                lam[s] = max(lam[s] / 5., 1e-12)
                if scaled[slot] < xtol or gain < ftol: active[s], converged[s] = False, True
            else:
                lam[s] *= 7.
                if lam[s] > 1e12: active[s] = False        # no uphill step left at any damping: a maximum to rounding, or a discontinuity
                if scaled[slot] < 1e-3 * xtol: active[s], converged[s] = False, True
    return x, f, g, H, iterations, converged


class GaussNewtonProfiler(BasePosteriorSampler):
    """``GaussNewtonProfiler(likelihood, seed=None, ref_scale=1., save_fn=None).maximize(niterations=4, start=None)`` -- the call surface of the reference's profilers
    (profilers/base.py: ``maximize(niterations, start)``; ``niterations`` = number of independent starts drawn from the parameters' ``ref`` distributions).
    Analytically solved parameters are varied with the others (as ``Fisher`` does, following the reference: fisher.py:688-695)."""

    def __init__(self, likelihood, save_fn=None, **kwargs):
        super(GaussNewtonProfiler, self).__init__(likelihood, **kwargs)
        self.fisher = Fisher(likelihood)
        self.params = self.fisher.varied_params
        self.save_fn = save_fn
        self.profiles = None

    def _prior_terms(self, points):
        """Log-prior, its gradient and minus its second derivative (diagonal) at ``points [S, P]``: exact for uniform / normal priors, by differences otherwise."""
        S, P = points.shape
        value, gradient, curvature = np.zeros(S), np.zeros((S, P)), np.zeros((S, P))
        for ip, param in enumerate(self.params):
            column, prior = points[:, ip], param.prior
            value += prior(column)
            if prior.dist == 'norm':
                gradient[:, ip], curvature[:, ip] = -(column - prior.loc) / prior.scale**2, 1. / prior.scale**2
            elif prior.dist != 'uniform':
                h = 1e-4 * float(getattr(prior, 'scale', 1.))
                up, mid, dn = prior(column + h), prior(column), prior(column - h)
                ok = np.isfinite(up) & np.isfinite(dn) & np.isfinite(mid)
                gradient[ok, ip] = (up[ok] - dn[ok]) / (2. * h)
                curvature[ok, ip] = np.maximum(-(up[ok] - 2. * mid[ok] + dn[ok]) / h**2, 0.)
        return value, gradient, curvature

    def _evaluate(self, points):
        points = np.ascontiguousarray(points, dtype='f8')
        offset, gradient, hessian = self.fisher.evaluate(points)
        value, pgradient, pcurvature = self._prior_terms(points)
        curvature = -hessian
        index = np.arange(points.shape[1])
        curvature[:, index, index] += pcurvature
        f = 0.5 * offset + value                       # (the offset is -D P D: the reference's convention, fisher.py:746)
        f[np.isnan(f)] = -np.inf
        return f, gradient + pgradient, curvature

    def _get_start_points(self, size):
        """Starts from the ``ref`` distributions with a finite posterior (samplers/base.py:274-323); solved parameters start at their values."""
        varied = [param.name for param in self.varied_params]
        start = np.empty((size, len(self.params)))
        coords, _ = self._get_start(size)
        for ip, param in enumerate(self.params):
            start[:, ip] = coords[:, varied.index(param.name)] if param.name in varied else param.value
        return start

    def maximize(self, niterations=None, start=None, max_iterations=100, xtol=1e-7, ftol=1e-9):
        """Maximise the log-posterior from ``niterations`` starts (default 4; or the rows of ``start [nstarts, P]``, columns ordered as ``self.params``)."""
        names = [param.name for param in self.params]
        if start is None: start = self._get_start_points(4 if niterations is None else int(niterations))
        start = np.atleast_2d(np.asarray(start, dtype='f8'))
        if start.shape[1] != len(names): raise ValueError('start must have {:d} columns ({})'.format(len(names), names))
        lower = np.array([param.prior.limits[0] for param in self.params], dtype='f8')
        upper = np.array([param.prior.limits[1] for param in self.params], dtype='f8')
        x, f, g, H, iterations, converged = levenberg_marquardt(self._evaluate, start, lower, upper, max_iterations=max_iterations, xtol=xtol, ftol=ftol)
        covariances = []
        for s in range(x.shape[0]):
            try: covariances.append(np.linalg.inv(H[s]))
            except np.linalg.LinAlgError: covariances.append(np.full(H[s].shape, np.nan))
        covariances = np.array(covariances)
        best = int(np.argmax(f))
        bestfit = {name: x[:, ip].copy() for ip, name in enumerate(names)}
        bestfit['logposterior'] = f.copy()
        error = {name: np.sqrt(np.abs(covariances[:, ip, ip])) for ip, name in enumerate(names)}
        self.profiles = Profiles({name: start[:, ip].copy() for ip, name in enumerate(names)}, bestfit, error, (names, covariances[best]),
                                 attrs={'iterations': iterations.tolist(), 'converged': converged.tolist(), 'gradient_norm': np.abs(g / np.sqrt(np.maximum(H[:, np.arange(len(names)), np.arange(len(names))], 1e-300))).max(axis=1).tolist()})
        if self.save_fn is not None: self.profiles.save(self.save_fn)
        return self.profiles

    def profile(self, params=None, size=30, cl=2., max_iterations=100, xtol=1e-7, ftol=1e-9):
        """1D profiles (profilers/base.py ``profile``): for ``size`` values of each parameter within ``cl`` errors of the best fit, the posterior maximised over all the
        OTHER parameters -- every grid point of every profiled parameter is a row of the same Levenberg-Marquardt batch.  Needs :meth:`maximize` first.  Fills and
        returns ``profiles.profile``: name -> array [size, 2] of (value, maximised log-posterior)."""
        if self.profiles is None: raise ValueError('run maximize first')
        names = [param.name for param in self.params]
        params = names if params is None else [str(name) for name in (params if isinstance(params, (list, tuple)) else [params])]
        best = self.profiles.choice()
        center = np.array([best[name] for name in names])
        index = self.profiles.argmax()
        lower = np.array([param.prior.limits[0] for param in self.params], dtype='f8')
        upper = np.array([param.prior.limits[1] for param in self.params], dtype='f8')
        starts, free, grids = [], [], {}
        for name in params:
            ip = names.index(name)
            error = float(self.profiles.error[name][index])
            grid = np.linspace(max(center[ip] - cl * error, lower[ip]), min(center[ip] + cl * error, upper[ip]), size)
            if np.isfinite(lower[ip]) and grid[0] <= lower[ip]: grid[0] = lower[ip] + 1e-9 * error
            if np.isfinite(upper[ip]) and grid[-1] >= upper[ip]: grid[-1] = upper[ip] - 1e-9 * error
            grids[name] = grid
            block = np.tile(center, (size, 1))
            block[:, ip] = grid
            # the other parameters start on the Gaussian ridge of the best fit: x_j = x^_j + C_jp / C_pp (x_p - x^_p)
            cov = self.profiles.covariance[1]
            if np.isfinite(cov).all() and cov[ip, ip] > 0.:
                block += np.outer(grid - center[ip], cov[:, ip] / cov[ip, ip]) * (np.arange(len(names)) != ip)
                block = np.clip(block, np.where(np.isfinite(lower), lower + 1e-12, -np.inf), np.where(np.isfinite(upper), upper - 1e-12, np.inf))
                block[:, ip] = grid
            mask = np.ones((size, len(names)), dtype='?'); mask[:, ip] = False
            starts.append(block); free.append(mask)
        x, f, g, H, iterations, converged = levenberg_marquardt(self._evaluate, np.concatenate(starts), lower, upper, max_iterations=max_iterations, xtol=xtol, ftol=ftol,
                                                                free=np.concatenate(free))
        self.profiles.profile = getattr(self.profiles, 'profile', None) or {}
        for iname, name in enumerate(params):
            self.profiles.profile[name] = np.column_stack([grids[name], f[iname * size:(iname + 1) * size]])
        return self.profiles.profile

    def grid(self, params, size=10, cl=2., max_iterations=100, xtol=1e-7, ftol=1e-9):
        """Best fits on a tensor grid of ``params`` (profilers/base.py ``grid``): ``size`` values per parameter within ``cl`` errors of the best fit, the posterior
        maximised over all the other parameters at every grid point -- the whole grid is one Levenberg-Marquardt batch (two parameters: the surface whose level lines
        are the confidence contours).  Fills and returns ``profiles.grid``: (list of the grid axes, maximised log-posterior of shape [size] * len(params))."""
        if self.profiles is None: raise ValueError('run maximize first')
        names = [param.name for param in self.params]
        params = [str(name) for name in (params if isinstance(params, (list, tuple)) else [params])]
        sizes = [int(size)] * len(params) if np.ndim(size) == 0 else [int(n) for n in size]
        best, index = self.profiles.choice(), self.profiles.argmax()
        center = np.array([best[name] for name in names])
        lower = np.array([param.prior.limits[0] for param in self.params], dtype='f8')
        upper = np.array([param.prior.limits[1] for param in self.params], dtype='f8')
        columns, axes = [names.index(name) for name in params], []
        for ip, n, name in zip(columns, sizes, params):
            error = float(self.profiles.error[name][index])
            axis = np.linspace(max(center[ip] - cl * error, lower[ip]), min(center[ip] + cl * error, upper[ip]), n)
            if np.isfinite(lower[ip]) and axis[0] <= lower[ip]: axis[0] = lower[ip] + 1e-9 * error
            if np.isfinite(upper[ip]) and axis[-1] >= upper[ip]: axis[-1] = upper[ip] - 1e-9 * error
            axes.append(axis)
        mesh = np.stack(np.meshgrid(*axes, indexing='ij'), axis=-1).reshape(-1, len(params))
        start = np.tile(center, (mesh.shape[0], 1))
        start[:, columns] = mesh
        cov = self.profiles.covariance[1]
        if np.isfinite(cov).all():
            # the other parameters start on the Gaussian ridge: x_o = x^_o + C_og C_gg^-1 (x_g - x^_g)
            others = [i for i in range(len(names)) if i not in columns]
            if others:
                shift = np.linalg.solve(cov[np.ix_(columns, columns)], (mesh - center[columns]).T).T.dot(cov[np.ix_(columns, others)])
                start[:, others] = np.clip(center[others] + shift, np.where(np.isfinite(lower[others]), lower[others] + 1e-12, -np.inf), np.where(np.isfinite(upper[others]), upper[others] - 1e-12, np.inf))
        free = np.ones(start.shape, dtype='?'); free[:, columns] = False
        x, f, g, H, iterations, converged = levenberg_marquardt(self._evaluate, start, lower, upper, max_iterations=max_iterations, xtol=xtol, ftol=ftol, free=free)
        self.profiles.grid = (axes, f.reshape(sizes))
        return self.profiles.grid

    def interval(self, params=None, cl=1., size=30):
        """Lower and upper limits where the profile drops by ``cl^2 / 2`` below the maximum (profilers/base.py ``interval``; from :meth:`profile`, by linear interpolation
        between grid points).  Returns and stores ``profiles.interval``: name -> (lower - bestfit, upper - bestfit); ``nan`` where the profile does not reach the level."""
        profile = self.profile(params=params, size=size, cl=2. * cl + 1.)
        best = self.profiles.choice()
        self.profiles.interval = getattr(self.profiles, 'interval', None) or {}
        for name, table in profile.items():
            values, logp = table[:, 0], table[:, 1]
            level = best['logposterior'] - 0.5 * cl**2
            limits = []
            for side in (values < best[name], values > best[name]):
                v, l = values[side], logp[side]
                order = np.argsort(np.abs(v - best[name]))
                v, l = np.concatenate([[best[name]], v[order]]), np.concatenate([[best['logposterior']], l[order]])
                below = np.flatnonzero(l < level)
                if not below.size: limits.append(np.nan); continue
                i = below[0]
                limits.append(v[i - 1] + (level - l[i - 1]) / (l[i] - l[i - 1]) * (v[i] - v[i - 1]) - best[name])
            self.profiles.interval[name] = tuple(limits)
        return self.profiles.interval
