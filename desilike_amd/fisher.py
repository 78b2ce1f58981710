"""Fisher matrix of a Gaussian likelihood (reference: desilike/fisher.py:642-809, 63-640; differentiation.py).

``Fisher(likelihood)(**center)`` evaluates the theory vector on the whole finite-difference stencil of the varied parameters as ONE GPU
batch (the reference's ``Differentiation`` scatters the stencil points over MPI ranks, differentiation.py:394-398) and ``dl_eval_fisher``
(csrc/dl_fisher.hip) forms ``d(flatdiff)/d(theta)`` by central differences of step ``Parameter.delta`` and the reference's Gaussian finalisation
(fisher.py:731-750) on the device: ``hessian = -dD P dD^T``, ``gradient = -dD P D``, ``offset = -D P D`` (no 1/2, as in the reference, line 746).
The host adds the Gaussian prior terms (fisher.py:706-716) and keeps the P x P ``LikelihoodFisher`` algebra (216-257).
"""
import numpy as np


class LikelihoodFisher(object):
    """Gaussian approximation of a likelihood around ``center`` (fisher.py:63-640, the parts the hot path produces / consumes)."""

    def __init__(self, center, params, offset=0., gradient=None, hessian=None, with_prior=False):
        self._center = np.asarray(center, dtype='f8')
        self._params = list(params)
        n = self._center.size
        self._offset = float(offset)
        self._gradient = np.zeros(n) if gradient is None else np.asarray(gradient, dtype='f8')
        self._hessian = np.zeros((n, n)) if hessian is None else np.asarray(hessian, dtype='f8')
        self.with_prior = bool(with_prior)

    def names(self):
        return [str(param) for param in self._params]

    def _solve(self):
        return np.linalg.solve(self._hessian, self._gradient)   # fisher.py:216-221

    @property
    def chi2min(self):
        flatdiff = -self._solve()   # fisher.py:224-227
        return -2. * (self._offset + self._gradient.dot(flatdiff) + 0.5 * flatdiff.dot(self._hessian).dot(flatdiff))

    def mean(self):
        return self._center - self._solve()   # fisher.py:229-243

    def center(self):
        return self._center.copy()

    def precision(self):
        return -self._hessian

    def covariance(self):
        return np.linalg.inv(-self._hessian)

    def std(self):
        return np.diag(self.covariance())**0.5

    def __add__(self, other):
        """Sum of independent Gaussian likelihoods expanded around the same centre."""
        assert self.names() == other.names() and np.allclose(self._center, other._center)
        return LikelihoodFisher(self._center, self._params, self._offset + other._offset, self._gradient + other._gradient, self._hessian + other._hessian,
                                with_prior=self.with_prior or other.with_prior)

    def to_likelihood(self):
        return FisherGaussianLikelihood(self)


class FisherGaussianLikelihood(object):
    """Gaussian-in-parameters likelihood built from a :class:`LikelihoodFisher` (likelihoods/galaxy_clustering/fisher.py:31-60 / fisher.py 'to_likelihood'):
    ``loglikelihood(theta) = offset + g (theta - c) + 1/2 (theta - c) H (theta - c)``."""

    def __init__(self, fisher):
        self.fisher = fisher

    def __call__(self, **params):
        diff = np.array([params[name] for name in self.fisher.names()]) - self.fisher._center
        return self.fisher._offset + self.fisher._gradient.dot(diff) + 0.5 * diff.dot(self.fisher._hessian).dot(diff)


class Fisher(object):
    """Estimate the Fisher matrix of ``likelihood`` (a Gaussian likelihood of this package) by finite differences, entirely on the GPU: the stencil of all
    centres goes through the theory kernels and the whitened window GEMM as one batch and ``dl_eval_fisher`` forms ``offset``, ``gradient`` and ``hessian``
    (fisher.py:739-748) with an fp64 MFMA Gram product per centre.

    Analytically solved parameters ('.marg' / '.best' / '.auto' / '.prec') are VARIED, as in the reference, which warns and works on a copy of the likelihood with
    ``derived=False`` for them (fisher.py:688-695)."""

    def __init__(self, likelihood, delta_scale=1.):
        self.likelihood = likelihood
        likelihood.initialize()
        self.delta_scale = float(delta_scale)
        solved = [param for param in likelihood.all_params if param.solved]
        if solved:
            import warnings
            warnings.warn('solved parameters: {}; cannot proceed with solved parameters, so they are varied in the Fisher estimate'.format([param.name for param in solved]))
        from .parameter import ParameterCollection
        self.varied_params = ParameterCollection(list(likelihood.varied_params) + solved)
        self._solved_names = [param.name for param in solved]
        self._ctx = None

    def _get_context(self):
        if self._ctx is None:
            like = self.likelihood
            if self._solved_names:
                from ._lib import Context
                self._ctx = Context(like._spec({}, like._flatdata_list(), like._precision_input, vary_solved=True), device=like.device)
            else:
                self._ctx = like._get_context()
        return self._ctx

    def steps(self, centers):
        """Lower / upper finite-difference steps ``[B, P, 2]`` at ``centers [B, P]``: ``Parameter.delta`` = (value, step below, step above) (parameter.py:898-915)
        scaled by ``delta_scale``, shortened where a prior bound is closer."""
        centers = np.atleast_2d(centers)
        steps = np.empty(centers.shape + (2,), dtype='f8')
        for ip, param in enumerate(self.varied_params):
            _, lower, upper = param.delta
            steps[:, ip, 0] = np.minimum(lower * self.delta_scale, centers[:, ip] - param.prior.limits[0])
            steps[:, ip, 1] = np.minimum(upper * self.delta_scale, param.prior.limits[1] - centers[:, ip])
        if not (steps.sum(axis=-1) > 0.).all():
            raise ValueError('zero finite-difference interval: a centre sits on both prior bounds of a parameter')
        return np.clip(steps, 0., None)

    def evaluate(self, centers):
        """``centers [B, P]`` (columns ordered as :attr:`varied_params`) -> (offset [B], gradient [B, P], hessian [B, P, P]) of the likelihood term, one GPU batch."""
        import torch
        centers = np.ascontiguousarray(np.atleast_2d(centers), dtype='f8')
        ctx = self._get_context()
        device = torch.device('cuda', ctx.device)
        hessian, gradient, offset = ctx.eval_fisher(torch.as_tensor(centers, device=device), torch.as_tensor(self.steps(centers), device=device).contiguous())
        return offset.cpu().numpy(), gradient.cpu().numpy(), hessian.cpu().numpy()

    def __call__(self, **params):
        varied = self.varied_params
        center = np.array([params.get(param.name, param.value) for param in varied], dtype='f8')
        offset, gradient, hessian = (array[0] for array in self.evaluate(center[None, :]))
        likelihood_fisher = LikelihoodFisher(center, varied, offset=offset, gradient=gradient, hessian=hessian)
        # Gaussian priors (fisher.py:706-716)
        poffset, pgradient, phessian = 0., [], []
        for param, value in zip(varied, center):
            loc, scale = (param.prior.loc, param.prior.scale) if param.prior.dist == 'norm' else (0., np.inf)
            prec = scale**(-2)
            poffset += -0.5 * (value - loc)**2 * prec
            pgradient.append(-(value - loc) * prec)
            phessian.append(-prec)
        prior_fisher = LikelihoodFisher(center, varied, offset=poffset, gradient=pgradient, hessian=np.diag(phessian), with_prior=True)
        self.likelihood_fisher, self.prior_fisher = likelihood_fisher, prior_fisher
        return likelihood_fisher + prior_fisher


def logposterior_value_and_grad(likelihood, theta, method='auto'):
    r"""Log-posterior and its gradient w.r.t. the varied parameters for a batch of points.

    ``method='analytic'``: ``dl_eval_logposterior_grad`` -- the gradient is formed on the device from the theory's own derivatives (csrc/dl_fullshape_grad.h:
    one forward pass, one extra GEMM, one gradient pass: about 2.5 evaluations instead of 2 P + 1), for Kaiser full-shape likelihoods with uniform / Gaussian priors;
    ``'finite'``: central differences evaluated as ONE GPU batch (below); ``'auto'``: analytic where the context supports it, else central differences.

    What gradient-based samplers ask of the likelihood (``jax.value_and_grad(logposterior)``: desilike/samplers/hmc.py:194, nuts.py:205, mclmc.py); the reference
    obtains it from jax tracing, here the stencil ``theta_b \pm h_p e_p`` of all B points and P parameters (B (2 P + 1) rows) goes through ``dl_eval_logposterior``
    in one call.  Steps are ``Parameter.delta`` (parameter.py:898-915), shortened where a prior bound is closer (one-sided there).  Analytically solved parameters
    are marginalised inside every evaluation.  Returns ``(logposterior [B], gradient [B, P])``; rows outside the prior get ``-inf`` and a NaN gradient.
    """
    import torch
    if method not in ('auto', 'analytic', 'finite'): raise ValueError('method must be one of auto, analytic, finite')
    likelihood.initialize()
    varied = likelihood.varied_params
    theta = np.atleast_2d(np.asarray(theta, dtype='f8'))
    B, P = theta.shape
    assert P == len(varied)
    if method != 'finite':
        ctx = likelihood._get_context()
        th = torch.as_tensor(theta, dtype=torch.float64, device=torch.device('cuda', ctx.device)).contiguous()
        out = ctx.eval_logposterior_grad(th) if not len(likelihood.solved_params) else None
        if out is not None:
            return out[0].cpu().numpy(), out[1].cpu().numpy()
        if method == 'analytic':
            raise NotImplementedError('the analytic gradient covers Kaiser full-shape likelihoods (uniform template knots, no counter terms, no damping, no transform, no solved '
                                      'parameters, uniform / norm priors): use method="finite"')
    lower, upper = np.empty((B, P)), np.empty((B, P))
    for ip, param in enumerate(varied):
        _, lo, hi = param.delta
        lower[:, ip] = np.clip(np.minimum(lo, theta[:, ip] - param.prior.limits[0]), 0., None)
        upper[:, ip] = np.clip(np.minimum(hi, param.prior.limits[1] - theta[:, ip]), 0., None)
    points = np.repeat(theta[:, None, :], 2 * P + 1, axis=1)                 # [B, 1 + 2 P, P]
    index = np.arange(P)
    points[:, 1 + 2 * index, index] -= lower
    points[:, 2 + 2 * index, index] += upper
    ctx = likelihood._get_context()
    device = torch.device('cuda', ctx.device)
    pts = torch.as_tensor(points.reshape(-1, P), dtype=torch.float64, device=device).contiguous()
    out = torch.empty(pts.shape[0], dtype=torch.float64, device=device)
    ctx.eval_logposterior(pts, out)
    logpost = out.cpu().numpy().reshape(B, 2 * P + 1)
    with np.errstate(invalid='ignore', divide='ignore'):
        gradient = (logpost[:, 2::2] - logpost[:, 1::2]) / (lower + upper)
    return logpost[:, 0], gradient
