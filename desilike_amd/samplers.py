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


class CounterRNG(object):
    """Counter-based random numbers for the ensemble sampler: Philox4x32-10 (Salmon et al. 2011, Random123) keyed by a 64-bit ``seed``; the draw for
    (iteration, stream, index) is a pure function of the seed.  The device-resident sampler (csrc/dl_ensemble.hip) evaluates the same function, so the
    host and the GPU drivers -- and every rank of a sharded run -- produce the same chain without exchanging random state."""
    PERM, MOVE, ACCEPT = 0, 1, 3   # stream ids (+ half-step for the last two)

    def __init__(self, seed=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    @staticmethod
    def philox4x32(counter, key):
        """``counter [..., 4]``, ``key [..., 2]`` uint32 -> uint32 ``[..., 4]`` (10 rounds)."""
        c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
        k0, k1 = np.asarray(key[..., 0], dtype=np.uint64), np.asarray(key[..., 1], dtype=np.uint64)
        mask = np.uint64(0xFFFFFFFF)
        for _ in range(10):
            p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
            c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask]
            k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
        return np.stack(c, axis=-1).astype(np.uint32)

    def draw(self, iteration, stream, n):
        """uint32 ``[n, 4]``: words of index 0 .. n - 1 of ``stream`` at ``iteration``."""
        counter = np.empty((n, 4), dtype=np.uint32)
        counter[:, 0], counter[:, 1] = int(iteration) & 0xFFFFFFFF, (int(iteration) >> 32) & 0xFFFFFFFF
        counter[:, 2], counter[:, 3] = np.arange(n, dtype=np.uint32), stream
        key = np.array([self.seed & 0xFFFFFFFF, self.seed >> 32], dtype=np.uint32)
        return self.philox4x32(counter, key)

    @staticmethod
    def uniform53(hi, lo):
        """53-bit uniform on [0, 1) from two 32-bit words (numpy's ``random_sample`` construction)."""
        return ((hi >> np.uint32(5)).astype('f8') * 67108864. + (lo >> np.uint32(6)).astype('f8')) / 9007199254740992.

    def permutation(self, iteration, n):
        """Random split of the ensemble: position r of the permutation holds walker F(r), F a keyed bijection of [0, n) -- four rounds of (odd multiplier, offset)
        mod 2^m followed by a right xor-shift, m = bit length of n - 1, keys = eight Philox words of the iteration, cycle-walked back into [0, n).  Every element
        is a pure function of (seed, iteration, r): no sort (the argsort of random keys this replaces was an O(n^2) ranking on one workgroup of the device, 5.7 us
        of every other half-step at 512 walkers; csrc/dl_ensemble.hip computes the same values)."""
        ka, kb = self.draw(iteration, self.PERM, 2)
        m = max(int(n - 1).bit_length(), 1)
        mask, shift = np.uint32((1 << m) - 1), np.uint32((m + 1) // 2)
        x = np.arange(n, dtype=np.uint32)
        todo = np.ones(n, dtype='?')
        while todo.any():
            y = x[todo]
            for r in range(4):
                y = (y * (ka[r] | np.uint32(1)) + kb[r]) & mask
                y = y ^ (y >> shift)
            x[todo] = y
            todo[todo] = y >= n
        return x.astype(int)

    def move(self, iteration, half, n):
        """(uniform [n], partner index in [0, n) [n]) of a half-step's stretch proposals."""
        words = self.draw(iteration, self.MOVE + half, n)
        return self.uniform53(words[:, 0], words[:, 1]), (words[:, 2] % np.uint32(n)).astype(int)

    def accept(self, iteration, half, n):
        words = self.draw(iteration, self.ACCEPT + half, n)
        return self.uniform53(words[:, 0], words[:, 1])


def _sync_random_state(rng, seed, sharding):
    """One random stream for all ranks of a sharded sampler (the reference broadcasts rng / seed: samplers/base.py:213-217, 278): without it every rank would
    propose different walkers and evaluate slices of different ensembles."""
    if rng is None and seed is None:
        seed = int(np.random.SeedSequence().entropy % (2**32))
    if sharding.active and sharding.world > 1:
        if rng is not None:
            kind, keys, pos, has_gauss, cached = rng.get_state()
            state = sharding.broadcast(np.concatenate([keys.astype('f8'), [pos, has_gauss, cached]]))
            rng = np.random.RandomState()
            rng.set_state((kind, state[:-3].astype(np.uint32), int(state[-3]), int(state[-2]), float(state[-1])))
        else:
            seed = int(sharding.broadcast(np.array([float(seed)]))[0])
    if rng is None:
        rng = np.random.RandomState(seed=seed)
    return rng, seed


class BasePosteriorSampler(object):

    def __init__(self, likelihood, rng=None, seed=None, max_tries=1000, ref_scale=1., sharding=None):
        self.likelihood = likelihood
        self.varied_params = likelihood.varied_params
        self.max_tries = int(max_tries)
        self.ref_scale = float(ref_scale)
        self.sharding = sharding if sharding is not None else WalkerSharding()
        self.rng, self.seed = _sync_random_state(rng, seed, self.sharding)
        self._vlikelihood = vmap(likelihood, errors='return', return_derived=True)
        self.derived = None
        self.fast = True   # False: always go through vmap(likelihood) (derived parameters kept by the likelihood's own call surface)

    def logprior(self, values):
        """Sum of the priors of the varied parameters (samplers/base.py:202-205)."""
        values = np.atleast_2d(values)
        toret = np.zeros(values.shape[0])
        for param, column in zip(self.varied_params, values.T):
            toret += param.prior(column)
        return toret

    def _posterior_context(self):
        """(device context, offset) of a GPU likelihood, or ``None``."""
        get_context = getattr(self.likelihood, '_get_posterior_context', None)
        if self.fast and get_context is not None:
            return get_context()
        return None

    def _logposterior_local(self, values):
        """samplers/base.py:144-193 for the rows handled by this process."""
        values = np.atleast_2d(np.asarray(values, dtype='f8'))
        context = self._posterior_context()
        if context is not None and values.shape[0]:
            # GPU likelihoods: the same conventions (NaN rows, rows outside the prior, non-finite results -> -inf) are applied by the finalize kernels:
            # ONE C-ABI call per batch instead of the dictionary round trip below; linear parameters with constant derivative rows are marginalised once, into the
            # precision matrix, instead of at every point
            ctx, offset = context
            return ctx.eval_logposterior_host(values)[0] + offset
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
        context = self._posterior_context()
        if context is not None and values.shape[0] and self.sharding.sharded(values.shape[0]):
            toret = self.sharding.map_logposterior(context[0], values, offset=context[1])   # device-resident exchange (RCCL through the C ABI)
        else:
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
    is called once per half-ensemble, like ``emcee.EnsembleSampler(vectorize=True)`` with its default move.

    ``rng``: a ``numpy.random.RandomState`` (sequential draws) or a :class:`CounterRNG` (counter-based draws: the chain is then bit-identical to the
    device-resident sampler's, ``dl_ensemble_*``)."""

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
        counter = isinstance(self.rng, CounterRNG)
        perm = self.rng.permutation(self.niterations, self.nwalkers) if counter else self.rng.permutation(self.nwalkers)
        for ihalf, (first, second) in enumerate([(perm[:half], perm[half:]), (perm[half:], perm[:half])]):
            if counter:
                uz, partner = self.rng.move(self.niterations, ihalf, half)
            else:
                uz = self.rng.uniform(size=half)
                partner = self.rng.randint(half, size=half)
            zz = ((self.a - 1.) * uz + 1.)**2 / self.a          # g(z) ~ 1 / sqrt(z) on [1 / a, a]
            partners = coords[second][partner]
            proposal = partners - (partners - coords[first]) * zz[:, None]
            new_log_prob = self.log_prob_fn(proposal)
            with np.errstate(invalid='ignore', divide='ignore'):
                lnpdiff = ((self.ndim - 1.) * np.log(zz) + new_log_prob) - log_prob[first]
                ua = self.rng.accept(self.niterations, ihalf, half) if counter else self.rng.uniform(size=half)
                accepted = np.log(ua) < lnpdiff
            idx = first[accepted]
            coords[idx], log_prob[idx] = proposal[accepted], new_log_prob[accepted]
            self.naccepted[idx] += 1
        self.niterations += 1
        return coords, log_prob

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.niterations, 1)


class EmceeSampler(BasePosteriorSampler):
    """Ensemble sampler with the reference's constructor surface (desilike/samplers/emcee.py:8-69).

    ``device_resident`` (default: True for GPU likelihoods unless ``use_emcee=True``): the whole ensemble update runs on the GPU (``dl_ensemble_*``: stretch
    proposals, log-posterior, accept / reject and the counter-based random generator on the device, one all-gather of log-posteriors per half-step over RCCL when
    the sampler's group is an :class:`~desilike_amd.parallel.RcclGroup`); the host drains the chain once per ``run``.  Otherwise ``emcee`` if installed, else the
    built-in :class:`EnsembleStretchMove` on the host."""
    name = 'emcee'

    def __init__(self, likelihood, nwalkers=None, use_emcee=None, device_resident=None, a=2., **kwargs):
        super(EmceeSampler, self).__init__(likelihood, **kwargs)
        ndim = len(self.varied_params)
        if nwalkers is None:
            nwalkers = 2 * max((int(2.5 * ndim) + 1) // 2, 2)                            # samplers/emcee.py:66
        if isinstance(nwalkers, str):
            nwalkers = int(eval(nwalkers, {'ndim': ndim}))
        self.nwalkers = int(nwalkers)
        self.chain = None
        self.a = float(a)
        if device_resident is None:
            # (parameters derived by an expression are computed by the host wrapper of the context: the device-resident ensemble does not see them)
            device_resident = use_emcee is not True and getattr(likelihood, '_get_posterior_context', None) is not None and not len(getattr(likelihood, 'dependent_params', []))
        self.device_resident = bool(device_resident)
        self._ensemble = None
        emcee = None
        if use_emcee is not False and not self.device_resident:
            try:
                import emcee
            except ImportError:
                if use_emcee: raise
        self._emcee = emcee
        if self.device_resident:
            if self.nwalkers % 2 or self.nwalkers < 2 * ndim:
                raise ValueError('nwalkers must be even and at least 2 * ndim')
            self.sampler = None
        elif emcee is not None:
            self.sampler = emcee.EnsembleSampler(self.nwalkers, ndim, self.logposterior, vectorize=True)   # samplers/emcee.py:69
        else:
            self.sampler = EnsembleStretchMove(self.nwalkers, ndim, self.logposterior, a=self.a, rng=self.rng)

    def _get_ensemble(self):
        if self._ensemble is None:
            from ._lib import DeviceEnsemble
            from .parallel import RcclGroup
            ctx, offset = self.likelihood._get_posterior_context()
            import os
            forced = os.environ.get('DL_ENS_FORCE_COMM', None) is not None and isinstance(self.sharding.group, RcclGroup)   # single-rank smoke test of the in-stream collective
            group = self.sharding.group if ((self.sharding.sharded(self.nwalkers // 2) or forced) and isinstance(self.sharding.group, RcclGroup)) else None
            # one 64-bit key for the device generator, drawn from the (rank-synchronised) host generator
            key = int(self.rng.randint(0, 2**32, dtype=np.uint64)) | (int(self.rng.randint(0, 2**32, dtype=np.uint64)) << 32)
            self._ensemble = DeviceEnsemble(ctx, self.nwalkers, a=self.a, seed=key, offset=offset, group=group)
            self.counter_seed = key
        return self._ensemble

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
        if self.device_resident:
            import torch
            ens = self._get_ensemble()
            device = torch.device('cuda', ens.device)
            if self.chain is None or self._last[0] is not start:
                ens.set_state(start, logposterior)
            coords = torch.empty((niterations, self.nwalkers, ens.n_params), dtype=torch.float64, device=device)
            logp = torch.empty((niterations, self.nwalkers), dtype=torch.float64, device=device)
            ens.run(niterations * thin_by, thin_by=thin_by, chain=coords, chain_logp=logp)
            coords, logp = coords.cpu().numpy(), logp.cpu().numpy()    # the one synchronisation of the run
            self._last = (coords[-1], logp[-1]) if niterations else (start, logposterior)
            self._naccepted = ens.get_state()[2]
            self._niterations = ens.info('iteration')
        elif self._emcee is not None:
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
        if self.device_resident:
            return self._naccepted / max(self._niterations, 1)
        return self.sampler.acceptance_fraction

    def save(self, fn):
        """Chains as ``.npz`` (the reference's checkpoint format, parameter.py:2164-2182): resume with ``load``."""
        np.savez(fn, **self.chain)

    def load(self, fn):
        data = np.load(fn)
        self.chain = {name: data[name] for name in data.files}
        last = np.column_stack([self.chain[param.name][-1] for param in self.varied_params])
        self._last = (last, self.chain['logposterior'][-1])
        if self._ensemble is not None:
            self._ensemble.set_state(*self._last)


def _expand_dict(values, names):
    """Value or {name / wildcard pattern: value} -> {name: value} (utils.expand_dict of the reference)."""
    import fnmatch
    if not isinstance(values, dict):
        return {name: values for name in names}
    toret = {name: None for name in names}
    for pattern, value in values.items():
        for name in fnmatch.filter(names, str(pattern)):
            toret[name] = value
    return toret


class _BatchEvaluator(object):
    """Common part of the grid / QMC samplers: ONE batched evaluation of the likelihood on all points (rows sharded over the process group if any),
    derived outputs and fixed parameters attached like the reference (samplers/grid.py:108-117, samplers/qmc.py:126-141)."""

    def _evaluate(self, samples, errors='raise'):
        calculator = self.calculator
        shape = samples.shape
        flat = {name: np.ravel(value) for name, value in samples.to_dict().items()}
        names = list(flat)
        size = flat[names[0]].size if names else 1

        def local(rows):
            (logposterior, derived), errs = vmap(calculator, errors='return', return_derived=True)({name: rows[:, i] for i, name in enumerate(names)})
            out = np.column_stack([np.asarray(derived[key], dtype='f8') for key in sorted(derived)])
            for ipoint in errs: out[ipoint] = np.nan
            self._derived_names = sorted(derived)
            return out

        values = np.column_stack([flat[name] for name in names]) if names else np.zeros((size, 0))
        out = self.sharding.map(local, values)
        bad = np.isnan(out).any(axis=1)
        if errors == 'raise' and bad.any():
            raise ValueError('non-finite evaluation at points {}'.format(np.flatnonzero(bad).tolist()))
        for param in calculator.all_params:
            if param.fixed and param.derived is False:
                samples[param] = np.full(shape, param.value, dtype='f8')
        for iname, name in enumerate(self._derived_names):
            samples[name] = out[:, iname].reshape(shape)
        return samples


class GridSampler(_BatchEvaluator):
    """Evaluate the likelihood on a grid (desilike/samplers/grid.py): ``size`` (int or {name: int}), ``ref_scale``, ``grid`` ({name: values}) as in the reference;
    ``run()`` returns ``Samples`` with the varied parameters (meshgrid, 'ij'), the fixed ones, and the derived ``loglikelihood`` / ``logprior`` (+ solved parameters)."""
    name = 'grid'

    def __init__(self, calculator, sharding=None, save_fn=None, **kwargs):
        self.calculator = calculator
        self.varied_params = calculator.varied_params
        self.sharding = sharding if sharding is not None else WalkerSharding()
        self.save_fn = save_fn
        self.set_grid(**kwargs)

    def set_grid(self, size=1, ref_scale=1., grid=None):
        from .parameter import ParameterError
        self.ref_scale = float(ref_scale)
        names = self.varied_params.names()
        grids, sizes = _expand_dict(grid, names), _expand_dict(size, names)
        self.grid = []
        for param in self.varied_params:
            grid, size = grids[param.name], sizes[param.name]
            if grid is None:
                if size is None:
                    raise ValueError('size (and grid) not specified for parameter {}'.format(param.name))
                size = int(size)
                if size < 1:
                    raise ValueError('size is {} < 1 for parameter {}'.format(size, param.name))
                center, limits = param.value, np.array(param.ref.limits, dtype='f8')
                if not limits[0] <= center <= limits[1]:
                    raise ParameterError('Parameter {} value {} is not in reference limits {}'.format(param.name, center, param.ref.limits))
                if size == 1:
                    grid = [center]
                else:   # samplers/grid.py:80-93
                    if param.ref.is_limited() and param.ref.dist == 'uniform':
                        edges = self.ref_scale * (limits - center) + center
                    elif param.proposal:
                        edges = self.ref_scale * np.array([-param.proposal, param.proposal]) + center
                    else:
                        raise ParameterError('Provide proper parameter reference distribution or proposal for {}'.format(param.name))
                    low, high = np.linspace(edges[0], center, size // 2 + 1), np.linspace(center, edges[1], size // 2 + 1)
                    grid = np.concatenate([low, high[1:]]) if size % 2 else np.concatenate([low[:-1], high[1:]])
            else:
                grid = np.sort(np.ravel(grid))
            self.grid.append(np.asarray(grid, dtype='f8'))
        self.samples = Samples(np.meshgrid(*self.grid, indexing='ij'), params=self.varied_params)

    def run(self, **kwargs):
        if kwargs: self.set_grid(**kwargs)
        self.samples = self._evaluate(self.samples)
        if self.save_fn is not None:
            np.savez(self.save_fn, **self.samples)
        return self.samples


class RQuasiRandomSequence(object):
    r"""Roberts' additive recurrence R_d: :math:`x_n = (s + n \alpha) \bmod 1`, :math:`\alpha_j = \phi_d^{-(j + 1)}`, :math:`\phi_d` the real root of
    :math:`x^{d+1} = x + 1` (the reference's default QMC engine, samplers/qmc.py:12-36)."""

    def __init__(self, d, seed=0.5):
        self.d, self.seed = int(d), float(seed)
        phi = 1.
        while abs(phi**(self.d + 1) - phi - 1.) > 1e-12:   # Newton's method
            phi -= (phi**(self.d + 1) - phi - 1.) / ((self.d + 1) * phi**self.d - 1.)
        self.alpha = np.array([phi**(-(1 + j)) for j in range(self.d)])
        self.num_generated = 0

    def random(self, n=1):
        toret = (self.seed + np.arange(self.num_generated + 1, self.num_generated + n + 1)[:, None] * self.alpha) % 1.
        self.num_generated += n
        return toret

    def reset(self):
        self.num_generated = 0
        return self

    def fast_forward(self, n):
        self.num_generated += n
        return self


class QMCSampler(_BatchEvaluator):
    """Quasi Monte-Carlo sequences (desilike/samplers/qmc.py): engines 'rqrs' (default), 'sobol', 'halton', 'lhs' (scipy.stats.qmc) scaled to
    ``value +- proposal`` of each varied parameter; non-finite evaluations are kept as NaN (``errors='nan'``), resumable through ``samples`` / ``offset``."""
    name = 'qmc'

    def __init__(self, calculator, samples=None, sharding=None, engine='rqrs', save_fn=None, **kwargs):
        self.calculator = calculator
        self.varied_params = calculator.varied_params
        self.sharding = sharding if sharding is not None else WalkerSharding()
        ndim = len(self.varied_params)
        if engine == 'rqrs':
            self.engine = RQuasiRandomSequence(ndim, **kwargs)
        elif isinstance(engine, str):
            from scipy.stats import qmc
            self.engine = {'sobol': qmc.Sobol, 'halton': qmc.Halton, 'lhs': qmc.LatinHypercube}[engine](d=ndim, **kwargs)
        else:
            self.engine = engine
        if isinstance(samples, str):
            data = np.load(samples)
            samples = Samples({name: data[name] for name in data.files})
        self.samples = samples
        self.save_fn = save_fn

    def run(self, niterations=300, offset=None):
        lower = [param.value - param.proposal for param in self.varied_params]
        upper = [param.value + param.proposal for param in self.varied_params]
        self.engine.reset()
        if offset is None:
            offset = len(next(iter(self.samples.values()))) if self.samples else 0
        if offset: self.engine.fast_forward(offset)
        unit = self.engine.random(n=niterations)
        samples = Samples((np.asarray(lower) + unit * (np.asarray(upper) - np.asarray(lower))).T, params=self.varied_params)
        samples = self._evaluate(samples, errors='nan')
        if self.samples:
            self.samples = Samples({name: np.concatenate([self.samples[name], samples[name]]) for name in samples})
        else:
            self.samples = samples
        if self.save_fn is not None:
            np.savez(self.save_fn, **self.samples)
        return self.samples
