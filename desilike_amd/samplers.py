"""Posterior samplers driving the batched likelihood: the callers of the hot path.

``BasePosteriorSampler.logposterior`` reproduces the input / output conventions of the reference
(desilike/samplers/base.py:45-66, 144-205): rows containing NaN -> -inf, rows outside the prior are not evaluated,
NaN results -> -inf, the returned value is ``loglikelihood + logprior`` of the likelihood's derived outputs.
``EmceeSampler`` mirrors desilike/samplers/emcee.py (``emcee.EnsembleSampler(..., vectorize=True)``): it uses ``emcee`` when it is
installed and otherwise the built-in affine-invariant stretch move (Goodman & Weare 2010), which calls ``logposterior`` once per
half-ensemble exactly like emcee's default ``StretchMove``.  With ``torch.distributed`` initialised (one process per GPU) the walkers of
every call are sharded across ranks and the log-posteriors all-gathered (``desilike_amd.parallel.WalkerSharding``).
"""
import numpy as np

from .base import vmap
from .parameter import Samples
from .parallel import WalkerSharding


class BasePosteriorSampler(object):

    def __init__(self, likelihood, rng=None, seed=None, max_tries=1000, ref_scale=1., sharding=None):
        self.likelihood = likelihood
        self.varied_params = likelihood.varied_params
        self.max_tries = int(max_tries)
        self.ref_scale = float(ref_scale)
        self.rng = rng if rng is not None else np.random.RandomState(seed=seed)
        self.sharding = sharding if sharding is not None else WalkerSharding()
        self._vlikelihood = vmap(likelihood, errors='return', return_derived=True)
        self.derived = None

    def logprior(self, values):
        """Sum of the priors of the varied parameters (samplers/base.py:202-205)."""
        values = np.atleast_2d(values)
        toret = np.zeros(values.shape[0])
        for param, column in zip(self.varied_params, values.T):
            toret += param.prior(column)
        return toret

    def _logposterior_local(self, values):
        """samplers/base.py:144-193 for the rows handled by this process."""
        values = np.atleast_2d(np.asarray(values, dtype='f8'))
        toret = np.full(values.shape[0], -np.inf)
        mask = ~np.isnan(values).any(axis=1)                      # bcast_values, samplers/base.py:57-61
        if not mask.any():
            return toret
        logprior = self.logprior(values[mask])
        finite = ~np.isinf(logprior)                              # samplers/base.py:147-149
        logposterior = logprior.copy()
        if finite.any():
            points = Samples(values[mask][finite].T, params=self.varied_params)
            (_, derived), errors = self._vlikelihood(points.to_dict())
            total = np.zeros(finite.sum())
            for param in [self.likelihood._param_loglikelihood, self.likelihood._param_logprior]:
                column = np.array(derived[param], dtype='f8')
                column[np.isnan(column)] = -np.inf                # samplers/base.py:187-189
                total += column
            for ipoint in errors:                                 # non-finite evaluations are caught up with -inf (samplers/base.py:166-177)
                total[ipoint] = -np.inf
            logposterior[finite] = total
        toret[mask] = logposterior
        return toret

    def logposterior(self, values):
        """values [B, ndim] (or [ndim]) -> log-posterior [B]; rows sharded over the process group, results all-gathered."""
        values = np.asarray(values, dtype='f8')
        isscalar = values.ndim == 1
        values = np.atleast_2d(values)
        toret = self.sharding.map(self._logposterior_local, values)
        return toret[0] if isscalar else toret

    def _get_start(self, size):
        """Draw starting points from the parameters' ``ref`` distributions until the posterior is finite (samplers/base.py:274-323)."""
        start = np.full((size, len(self.varied_params)), np.nan)
        logposterior = np.full(size, -np.inf)
        for _ in range(self.max_tries):
            mask = ~np.isfinite(logposterior)
            if not mask.any():
                break
            for iparam, param in enumerate(self.varied_params):
                if param.ref.is_proper():
                    draw = param.ref.sample(size=int(mask.sum()), random_state=self.rng)
                    if self.ref_scale != 1.:
                        draw = param.value + self.ref_scale * (draw - param.value)
                else:
                    draw = np.full(int(mask.sum()), param.value)
                start[mask, iparam] = draw
            start = self.sharding.broadcast(start)
            logposterior[mask] = self.logposterior(start[mask])
        if not np.isfinite(logposterior).all():
            raise ValueError('Could not find finite log posterior after {:d} tries'.format(self.max_tries))
        return start, logposterior


class EnsembleStretchMove(object):
    """Affine-invariant ensemble sampler (Goodman & Weare 2010, stretch move with a = 2), vectorised: ``log_prob_fn(coords [n, ndim]) -> [n]``
    is called once per half-ensemble, like ``emcee.EnsembleSampler(vectorize=True)`` with its default move."""

    def __init__(self, nwalkers, ndim, log_prob_fn, a=2., rng=None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError('nwalkers must be even and at least 2 * ndim')
        self.nwalkers, self.ndim, self.log_prob_fn, self.a = int(nwalkers), int(ndim), log_prob_fn, float(a)
        self.rng = rng if rng is not None else np.random.RandomState()
        self.naccepted = np.zeros(self.nwalkers)
        self.niterations = 0

    def step(self, coords, log_prob):
        coords, log_prob = coords.copy(), log_prob.copy()
        half = self.nwalkers // 2
        perm = self.rng.permutation(self.nwalkers)
        for first, second in [(perm[:half], perm[half:]), (perm[half:], perm[:half])]:
            zz = ((self.a - 1.) * self.rng.uniform(size=half) + 1.)**2 / self.a          # g(z) ~ 1 / sqrt(z) on [1 / a, a]
            partners = coords[second][self.rng.randint(half, size=half)]
            proposal = partners - (partners - coords[first]) * zz[:, None]
            new_log_prob = self.log_prob_fn(proposal)
            lnpdiff = (self.ndim - 1.) * np.log(zz) + new_log_prob - log_prob[first]
            accepted = np.log(self.rng.uniform(size=half)) < lnpdiff
            idx = first[accepted]
            coords[idx], log_prob[idx] = proposal[accepted], new_log_prob[accepted]
            self.naccepted[idx] += 1
        self.niterations += 1
        return coords, log_prob

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.niterations, 1)


class EmceeSampler(BasePosteriorSampler):
    """Ensemble sampler with the reference's constructor surface (desilike/samplers/emcee.py:8-69)."""
    name = 'emcee'

    def __init__(self, likelihood, nwalkers=None, use_emcee=None, **kwargs):
        super(EmceeSampler, self).__init__(likelihood, **kwargs)
        ndim = len(self.varied_params)
        if nwalkers is None:
            nwalkers = 2 * max((int(2.5 * ndim) + 1) // 2, 2)                            # samplers/emcee.py:66
        if isinstance(nwalkers, str):
            nwalkers = int(eval(nwalkers, {'ndim': ndim}))
        self.nwalkers = int(nwalkers)
        self.chain = None
        emcee = None
        if use_emcee is not False:
            try:
                import emcee
            except ImportError:
                if use_emcee: raise
        self._emcee = emcee
        if emcee is not None:
            self.sampler = emcee.EnsembleSampler(self.nwalkers, ndim, self.logposterior, vectorize=True)   # samplers/emcee.py:69
        else:
            self.sampler = EnsembleStretchMove(self.nwalkers, ndim, self.logposterior, rng=self.rng)

    def run(self, niterations=300, thin_by=1, start=None):
        """Run ``niterations`` ensemble updates; returns dict(name -> [niterations, nwalkers]) incl. 'logposterior' (cf. samplers/emcee.py:101-111)."""
        if start is None:
            if self.chain is not None:
                start, logposterior = self._last
            else:
                start, logposterior = self._get_start(self.nwalkers)
        else:
            start = np.asarray(start, dtype='f8')
            logposterior = self.logposterior(start)
        if self._emcee is not None:
            self.sampler._random = self.rng
            state = self._emcee.State(start, log_prob=logposterior)
            for state in self.sampler.sample(initial_state=state, iterations=niterations, thin_by=thin_by, store=True):
                pass
            coords, logp = self.sampler.get_chain(), self.sampler.get_log_prob()
            self._last = (coords[-1], logp[-1])
        else:
            coords, logp = [], []
            for it in range(niterations * thin_by):
                start, logposterior = self.sampler.step(start, logposterior)
                if (it + 1) % thin_by == 0:
                    coords.append(start.copy()); logp.append(logposterior.copy())
            coords, logp = np.array(coords), np.array(logp)
            self._last = (start, logposterior)
        chain = {param.name: coords[..., iparam] for iparam, param in enumerate(self.varied_params)}
        chain['logposterior'] = logp
        if self.chain is None:
            self.chain = chain
        else:
            self.chain = {name: np.concatenate([self.chain[name], chain[name]], axis=0) for name in chain}
        return self.chain

    @property
    def acceptance_fraction(self):
        return self.sampler.acceptance_fraction

    def save(self, fn):
        """Chains as ``.npz`` (the reference's checkpoint format, parameter.py:2164-2182): resume with ``load``."""
        np.savez(fn, **self.chain)

    def load(self, fn):
        data = np.load(fn)
        self.chain = {name: data[name] for name in data.files}
        last = np.column_stack([self.chain[param.name][-1] for param in self.varied_params])
        self._last = (last, self.chain['logposterior'][-1])
