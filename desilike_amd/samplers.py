"""Posterior samplers driving the batched likelihood: the callers of the hot path.

``BasePosteriorSampler.logposterior`` reproduces the input / output conventions of the reference
(desilike/samplers/base.py:45-66, 144-205): rows containing NaN -> -inf, rows outside the prior are not evaluated,
NaN results -> -inf, the returned value is ``loglikelihood + logprior`` of the likelihood's derived outputs.
``EmceeSampler`` mirrors desilike/samplers/emcee.py (``emcee.EnsembleSampler(..., vectorize=True)``): it uses ``emcee`` when it is
installed and otherwise the built-in affine-invariant stretch move (Goodman & Weare 2010), which calls ``logposterior`` once per
half-ensemble exactly like emcee's default ``StretchMove``.  With ``torch.distributed`` initialised (one process per GPU) the walkers of
every call are sharded across ranks and the log-posteriors all-gathered (``desilike_amd.parallel.WalkerSharding``).
"""
import os

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


class _HostChain(object):
    """One chain driven on the host: :class:`EnsembleStretchMove` over a vectorised log-posterior function."""
    device_resident = False

    def __init__(self, nwalkers, ndim, log_prob_fn, a, rng):
        self.move = EnsembleStretchMove(nwalkers, ndim, log_prob_fn, a=a, rng=rng)
        self.coords = self.logp = None
        self._out = None

    def set_state(self, coords, logp, iteration=0, naccepted=None):
        self.coords, self.logp = np.array(coords, dtype='f8'), np.array(logp, dtype='f8')
        self.move.niterations = int(iteration)
        self.move.naccepted = np.zeros(self.move.nwalkers) if naccepted is None else np.array(naccepted, dtype='f8')

    def enqueue(self, niterations, thin_by=1):
        coords, logp = [], []
        for it in range(niterations * thin_by):
            self.coords, self.logp = self.move.step(self.coords, self.logp)
            if (it + 1) % thin_by == 0:
                coords.append(self.coords.copy()); logp.append(self.logp.copy())
        ndim = self.move.ndim
        self._out = (np.array(coords).reshape(niterations, self.move.nwalkers, ndim), np.array(logp).reshape(niterations, self.move.nwalkers))

    def collect(self):
        out, self._out = self._out, None
        return out

    @property
    def iteration(self):
        return self.move.niterations

    @property
    def naccepted(self):
        return self.move.naccepted


class _DeviceChain(object):
    """One chain resident on the GPU (``dl_ensemble_*``) with its own HIP stream: several chains of one process run concurrently, each a sequence of small
    latency-bound launches whose ramps and tails overlap the other chains' kernels."""
    device_resident = True

    def __init__(self, ctx, offset, nwalkers, a, key, group=None, own_stream=False):
        import torch
        from ._lib import DeviceEnsemble
        self.ens = DeviceEnsemble(ctx, nwalkers, a=a, seed=key, offset=offset, group=group)
        self.device = torch.device('cuda', self.ens.device)
        self.stream = torch.cuda.Stream(device=self.device) if own_stream else None
        self._buffers = self._device_buffers = self._host_buffers = None
        self._naccepted = np.zeros(nwalkers)

    def _cuda_stream(self):
        return None if self.stream is None else self.stream.cuda_stream

    def set_state(self, coords, logp, iteration=0, naccepted=None):
        self.ens.set_state(coords, logp, stream=self._cuda_stream())
        self.ens.set_counter(iteration, naccepted=None if naccepted is None else np.asarray(naccepted, dtype='i8'), stream=self._cuda_stream())

    def enqueue(self, niterations, thin_by=1):
        """``niterations`` recorded updates and the copy of the recorded chain to (pinned) host memory, all enqueued on the chain's stream: nothing waits here, the copies
        of one chain overlap the updates of the others."""
        import torch
        shape = (niterations, self.ens.nwalkers, self.ens.n_params)
        if self._device_buffers is None or tuple(self._device_buffers[0].shape) != shape:
            self._device_buffers = (torch.empty(shape, dtype=torch.float64, device=self.device), torch.empty(shape[:2], dtype=torch.float64, device=self.device))
            self._host_buffers = (torch.empty(shape, dtype=torch.float64, pin_memory=True), torch.empty(shape[:2], dtype=torch.float64, pin_memory=True))
        coords, logp = self._device_buffers
        stream = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))   # (buffers made / last read on the current stream)
        self.ens.run(niterations * thin_by, thin_by=thin_by, chain=coords, chain_logp=logp, stream=stream.cuda_stream)
        with torch.cuda.stream(stream):
            self._host_buffers[0].copy_(coords, non_blocking=True)
            self._host_buffers[1].copy_(logp, non_blocking=True)
        self._buffers = (coords, logp)

    def collect(self, device=False):
        """The chain of the last ``enqueue`` (host arrays; ``device=True``: the device tensors), once the chain's stream has finished."""
        import torch
        coords, logp = self._buffers
        self._buffers = None
        (self.stream if self.stream is not None else torch.cuda.current_stream(self.device)).synchronize()
        if device: return coords, logp
        return self._host_buffers[0].numpy(), self._host_buffers[1].numpy()     # (views of the pinned staging buffers: valid until the next enqueue; the caller copies)

    @property
    def iteration(self):
        return self.ens.info('iteration')

    @property
    def naccepted(self):
        return self.ens.get_state(stream=self._cuda_stream())[2].astype('f8')


class _ChainStore(object):
    """Samples of one chain, growing by batches: coords [n, nwalkers, ndim], logposterior [n, nwalkers] in arrays with amortised doubling -- appending a batch is ONE
    copy of the batch (straight out of the pinned staging buffer of the device chain), whatever the length of the chain so far.

    A doubling is as much fresh memory as the whole chain so far, and fresh memory costs its first touch (0.5 ms per MB here: 39 ms for the 78 MB of 2400 updates of 512
    walkers -- three batches' worth of sampling; round 6: the K = 1 figure of the bench moved between 7.8 and 12.2 M evaluations / s depending on whether a doubling fell
    among its four timed batches).  So the next buffer is prepared IN THE BACKGROUND as soon as the current one is half full: a thread allocates it, touches its pages and
    copies the rows that exist by then (rows never change once appended); the swap copies the few rows appended since."""

    def __init__(self):
        self.size = 0
        self._coords = self._logp = None
        self._prep = None          # (thread, coords, logp, rows copied by the thread) of the buffer being prepared

    def __bool__(self):
        return self._coords is not None

    def _prepare(self, cap):
        import threading
        old_c, old_l, copied = self._coords, self._logp, self.size
        new_c, new_l = np.empty((cap,) + old_c.shape[1:], dtype='f8'), np.empty((cap,) + old_l.shape[1:], dtype='f8')

        def work():
            new_c[:copied], new_l[:copied] = old_c[:copied], old_l[:copied]
            step = max(1, (1 << 21) // max(1, new_c[0].size))      # ~16 MB at a time (NumPy releases the interpreter lock inside the assignment)
            for i in range(copied, cap, step):
                new_c[i:i + step] = 0.
                new_l[i:i + step] = 0.

        thread = threading.Thread(target=work, daemon=True)
        thread.start()
        self._prep = (thread, new_c, new_l, copied)

    def _grow(self, need):
        if self._prep is not None and self._prep[1].shape[0] >= need:
            thread, new_c, new_l, copied = self._prep
            thread.join()
            new_c[copied:self.size], new_l[copied:self.size] = self._coords[copied:self.size], self._logp[copied:self.size]
            self._coords, self._logp = new_c, new_l
        else:                      # (a batch larger than the chain so far: grow at once)
            if self._prep is not None: self._prep[0].join()
            cap = max(2 * self._coords.shape[0], need)
            for name in ['_coords', '_logp']:
                old = getattr(self, name)
                new = np.empty((cap,) + old.shape[1:], dtype='f8')
                new[:self.size] = old[:self.size]
                new[self.size:] = 0.
                setattr(self, name, new)
        self._prep = None

    def reserve(self, n, shapes=None):
        """Room for ``n`` more samples in memory that has been touched: called while the device runs the batch, so that the append that follows is one copy."""
        if self._coords is None:
            if shapes is None: return
            cap = 4 * max(n, 1)
            self._coords, self._logp = np.zeros((cap,) + tuple(shapes[0]), dtype='f8'), np.zeros((cap,) + tuple(shapes[1]), dtype='f8')
            self._coords.fill(0.); self._logp.fill(0.)
        cap = self._coords.shape[0]
        if self.size + n > cap: self._grow(self.size + n)
        cap = self._coords.shape[0]
        if self._prep is None and 2 * (self.size + n) > cap: self._prepare(2 * cap)

    def append(self, coords, logp):
        n = coords.shape[0]
        self.reserve(n, shapes=(coords.shape[1:], logp.shape[1:]))
        self._coords[self.size:self.size + n], self._logp[self.size:self.size + n] = coords, logp
        self.size += n

    @property
    def coords(self):
        return self._coords[:self.size]

    @property
    def logp(self):
        return self._logp[:self.size]


def _batch_iterate(func, min_iterations=0, max_iterations=None, check_every=300):
    """samplers/base.py:28-42: batches of ``check_every`` iterations until ``func`` reports convergence (not before ``min_iterations``) or ``max_iterations``."""
    import sys
    if max_iterations is None: max_iterations = sys.maxsize
    if max_iterations < 0: raise ValueError('max_iterations must be positive')
    if check_every < 1: raise ValueError('check_every must be >= 1, found {:d}'.format(check_every))
    count, converged = 0, False
    while not converged:
        niter = min(max_iterations - count, check_every)
        count += niter
        converged = func(niter)
        if count < min_iterations: converged = False
        if count >= max_iterations: converged = True
    return count


class EmceeSampler(BasePosteriorSampler):
    """Ensemble sampler with the reference's constructor surface (desilike/samplers/emcee.py:8-69; ``chains``, ``run(min_iterations, max_iterations, check_every,
    check)`` and ``check`` of desilike/samplers/base.py:409-724).

    ``device_resident`` (default: True for GPU likelihoods unless ``use_emcee=True``): the whole ensemble update runs on the GPU (``dl_ensemble_*``: stretch
    proposals, log-posterior, accept / reject and the counter-based random generator on the device); the host drains the chain once per batch.  Otherwise
    ``emcee`` if installed, else the built-in :class:`EnsembleStretchMove` on the host.

    ``chains``: number of independent chains (or a list of chain files written by :meth:`save` to resume from).  **Chains are the unit of parallelism**, as in the
    reference (one chain per group of MPI ranks, desilike/utils.py:1040-1148): with a process group (one process per GPU) chain c lives on rank ``c % world``;
    a rank that holds several chains runs them concurrently on separate HIP streams.  Every chain has its own device-resident ensemble and its own Philox key
    (drawn from the rank-synchronised host generator, or given as ``counter_seeds``): a chain is bit-identical to the single-chain run with the same key and start,
    whatever the number of ranks.  Nothing is exchanged inside a batch of ``check_every`` iterations; at its end the new samples of all chains are all-gathered
    (one collective), so that every rank holds every chain and evaluates the same convergence tests (:meth:`check`: Gelman-Rubin across chains ...).
    With ``chains=1`` a process group shards the WALKERS of the one ensemble instead (BASELINE configs[4] as written; see ``WalkerSharding``)."""
    name = 'emcee'

    def __init__(self, likelihood, nwalkers=None, chains=1, use_emcee=None, device_resident=None, a=2., counter_seeds=None, save_fn=None, **kwargs):
        super(EmceeSampler, self).__init__(likelihood, **kwargs)
        ndim = len(self.varied_params)
        if nwalkers is None:
            nwalkers = 2 * max((int(2.5 * ndim) + 1) // 2, 2)                            # samplers/emcee.py:66
        if isinstance(nwalkers, str):
            nwalkers = int(eval(nwalkers, {'ndim': ndim}))
        self.nwalkers = int(nwalkers)
        self.a = float(a)
        resume = None
        if not isinstance(chains, (int, np.integer)):
            resume = [chains] if isinstance(chains, (str, dict)) or hasattr(chains, 'arrays') else list(chains)
            chains = len(resume)
        self.nchains = int(chains)
        if self.nchains < 1: raise ValueError('chains must be >= 1')
        self.chain_parallel = self.nchains > 1
        self.chain_group = None
        if self.chain_parallel:
            # chains are distributed over the ranks; inside a chain nothing is exchanged (every evaluation is local)
            self.chain_group = self.sharding.group if self.sharding.active and self.sharding.world > 1 else None
            self.sharding = WalkerSharding(group=False)
        self.chain_rank = self.chain_group.rank if self.chain_group is not None else 0
        self.chain_world = self.chain_group.world if self.chain_group is not None else 1
        if device_resident is None:
            # (parameters derived by an expression are computed by the host wrapper of the context: the device-resident ensemble does not see them)
            device_resident = use_emcee is not True and getattr(likelihood, '_get_posterior_context', None) is not None and not len(getattr(likelihood, 'dependent_params', []))
        self.device_resident = bool(device_resident)
        emcee = None
        if use_emcee is not False and not self.device_resident and not self.chain_parallel and counter_seeds is None:
            try:
                import emcee
            except ImportError:
                if use_emcee: raise
        self._emcee = emcee
        if (self.device_resident or emcee is None) and (self.nwalkers % 2 or self.nwalkers < 2 * ndim):
            raise ValueError('nwalkers must be even and at least 2 * ndim')
        self.sampler = emcee.EnsembleSampler(self.nwalkers, ndim, self.logposterior, vectorize=True) if emcee is not None else None   # samplers/emcee.py:69
        self._counter_given = counter_seeds is not None    # host-driven chains then draw from the counter-based generator too (same chain as the device's)
        self.counter_seeds = None if counter_seeds is None else [int(key) & 0xFFFFFFFFFFFFFFFF for key in np.atleast_1d(counter_seeds)]
        if self.counter_seeds is not None and len(self.counter_seeds) != self.nchains:
            raise ValueError('provide one counter seed per chain')
        if save_fn is not None:
            if isinstance(save_fn, str):
                save_fn = [save_fn.replace('*', str(ichain)) for ichain in range(self.nchains)]
            save_fn = list(save_fn)
            if len(save_fn) != self.nchains or len(set(save_fn)) != self.nchains:
                raise ValueError('provide one file name per chain (or a template with *)')
        self.save_fn = save_fn
        self._blocks = [_ChainStore() for _ in range(self.nchains)]   # per chain: coords [n, nwalkers, ndim], logposterior [n, nwalkers] so far, on every rank
        self._state = [None] * self.nchains           # per chain: (coords [nwalkers, ndim], logposterior [nwalkers]) after the last update
        self._iterations = [0] * self.nchains         # updates done (the counter of the chain's generator)
        self._accepted = [np.zeros(self.nwalkers) for _ in range(self.nchains)]
        self._runners = {}                            # ichain -> runner, for the chains of this rank
        self._runner_signature = None
        self._holds = {}                              # ichain -> the coords array the runner currently continues from (identity)
        self._refresh = set()                         # chains whose stored log-posteriors must be re-evaluated (the likelihood's parameters changed)
        self.diagnostics = {}
        if resume is not None:
            for ichain, source in enumerate(resume):
                self._load_one(ichain, source)

    @property
    def chains(self):
        """Per chain: dict name -> [niterations, nwalkers] (incl. 'logposterior'; views of the chain's store), or None before the first update."""
        out = []
        for store in self._blocks:
            if not store:
                out.append(None); continue
            chain = {param.name: store.coords[..., iparam] for iparam, param in enumerate(self.varied_params)}
            chain['logposterior'] = store.logp
            out.append(chain)
        return out

    # ---- single-chain views (the surface of the one-chain sampler) -----------------------------------------------------------------------------------------
    @property
    def chain(self):
        return self.chains[0]

    @property
    def _last(self):
        return self._state[0]

    @property
    def counter_seed(self):
        return None if self.counter_seeds is None else self.counter_seeds[0]

    def local_chains(self):
        """Indices of the chains this rank runs."""
        return [ichain for ichain in range(self.nchains) if ichain % self.chain_world == self.chain_rank]

    def _draw_keys(self):
        if self.counter_seeds is None:
            # one 64-bit key per chain for the counter-based generator, drawn from the (rank-synchronised) host generator
            self.counter_seeds = [int(self.rng.randint(0, 2**32, dtype=np.uint64)) | (int(self.rng.randint(0, 2**32, dtype=np.uint64)) << 32) for _ in range(self.nchains)]
        return self.counter_seeds

    def _likelihood_signature(self):
        """Changes when the likelihood's parameters do (the compiled device contexts follow ``all_params``; so must the ensembles built on them)."""
        check = getattr(self.likelihood, '_check_params', None)
        if check is None: return None
        check()
        return getattr(self.likelihood, '_params_signature', None)

    def _get_runner(self, ichain):
        signature = self._likelihood_signature()
        if self._runners and signature != self._runner_signature:
            for runner in self._runners.values():
                if runner.device_resident: runner.ens.close()
            self._runners, self._holds = {}, {}
            self._refresh = set(range(self.nchains))     # the log-posteriors of the current positions belong to the old parameters: re-evaluated at hand-over
        self._runner_signature = signature
        if ichain not in self._runners:
            keys = self._draw_keys()
            ndim = len(self.varied_params)
            if self.device_resident:
                from .parallel import RcclGroup
                import os
                local = self.local_chains()
                # one context per chain of this rank: the chains run concurrently on separate streams and a context's workspaces serve one call at a time
                replica = local.index(ichain) if ichain in local else 0
                ctx, offset = self.likelihood._get_posterior_context(replica=replica) if replica else self.likelihood._get_posterior_context()
                group = None
                if not self.chain_parallel:
                    forced = os.environ.get('DL_ENS_FORCE_COMM', None) is not None and isinstance(self.sharding.group, RcclGroup)   # single-rank smoke test of the in-stream collective
                    group = self.sharding.group if ((self.sharding.sharded(self.nwalkers // 2) or forced) and isinstance(self.sharding.group, RcclGroup)) else None
                self._runners[ichain] = _DeviceChain(ctx, offset, self.nwalkers, self.a, keys[ichain], group=group, own_stream=len(self.local_chains()) > 1)
            else:
                rng = CounterRNG(keys[ichain]) if (self.chain_parallel or self._counter_given) else self.rng
                self._runners[ichain] = _HostChain(self.nwalkers, ndim, self.logposterior, self.a, rng)
        return self._runners[ichain]

    def _get_ensemble(self, ichain=0):
        """The device ensemble of chain ``ichain`` (``DeviceEnsemble``)."""
        return self._get_runner(ichain).ens

    def _starts(self, start=None):
        """(coords, logposterior) per chain that has none yet: drawn like the reference, chain after chain from the synchronised generator (samplers/base.py:274-323)."""
        if start is not None:
            start = np.asarray(start, dtype='f8')
            if start.ndim == 2 and self.nchains == 1: start = start[None]
            if start.shape != (self.nchains, self.nwalkers, len(self.varied_params)):
                raise ValueError('Provide start with shape {}'.format((self.nchains, self.nwalkers, len(self.varied_params))))
            logp = self.logposterior(start.reshape(-1, start.shape[-1])).reshape(self.nchains, self.nwalkers)
            for ichain in range(self.nchains):
                self._state[ichain] = (start[ichain], logp[ichain])
                self._holds.pop(ichain, None)
            return
        for ichain in range(self.nchains):
            if self._state[ichain] is None:
                self._state[ichain] = self._get_start(self.nwalkers)

    def _run_batch(self, niterations, thin_by=1):
        """``niterations`` recorded updates of every chain: the chains of this rank concurrently, then one all-gather of the new samples."""
        local = self.local_chains()
        ndim = len(self.varied_params)
        if self._emcee is not None:
            start, logposterior = self._state[0]
            self.sampler._random = self.rng
            state = self._emcee.State(start, log_prob=logposterior)
            first = self.sampler.iteration
            for state in self.sampler.sample(initial_state=state, iterations=niterations, thin_by=thin_by, store=True):
                pass
            new = {0: (self.sampler.get_chain()[first:], self.sampler.get_log_prob()[first:])}
            self._accepted[0] = self.sampler.acceptance_fraction * self.sampler.iteration
            self._iterations[0] = self.sampler.iteration
        else:
            runners = []
            for ichain in local:
                runner = self._get_runner(ichain)
                coords, logp = self._state[ichain]
                if ichain in self._refresh:
                    self._refresh.discard(ichain)
                    logp = self.logposterior(coords)
                    self._state[ichain] = (coords, logp)
                if self._holds.get(ichain, None) is not coords:      # a new runner, a loaded chain, a new start: hand over positions, log-posteriors and counters
                    runner.set_state(coords, logp, iteration=self._iterations[ichain], naccepted=self._accepted[ichain])
                runners.append(runner)
            for runner in runners:
                runner.enqueue(niterations, thin_by=thin_by)        # (device chains: asynchronous, one stream each)
            for store in self._blocks:                              # (while the device runs: the stores of ALL chains -- gathered ones too -- grow now, not inside the appends below)
                store.reserve(niterations, shapes=((self.nwalkers, ndim), (self.nwalkers,)))
            new = {}
            for ichain, runner in zip(local, runners):
                new[ichain] = runner.collect()                       # the one synchronisation of the batch
                self._accepted[ichain] = np.asarray(runner.naccepted, dtype='f8')
                self._iterations[ichain] = runner.iteration
        if self.chain_group is not None:
            new = self._gather_chains(new, niterations, ndim)
        for ichain in range(self.nchains):
            coords, logp = new[ichain]
            if niterations:
                self._state[ichain] = (coords[-1].copy(), logp[-1].copy())
            if ichain in self._runners: self._holds[ichain] = self._state[ichain][0]
            self._blocks[ichain].append(coords, logp)

    def _gather_chains(self, new, niterations, ndim):
        """All-gather of the batch's samples: every rank ends up with every chain (one collective per batch: [chains per rank, niterations, nwalkers, ndim + 1 + 2]
        doubles per rank; the two extra columns carry the accepted counts and the iteration counter)."""
        nmax = (self.nchains + self.chain_world - 1) // self.chain_world
        block = np.zeros((nmax, niterations + 1, self.nwalkers, ndim + 1), dtype='f8')
        for slot, ichain in enumerate(self.local_chains()):
            coords, logp = new[ichain]
            block[slot, :niterations, :, :ndim], block[slot, :niterations, :, ndim] = coords, logp
            block[slot, niterations, :, 0], block[slot, niterations, 0, 1] = self._accepted[ichain], self._iterations[ichain]
        gathered = np.asarray(self.chain_group.allgather(block)).reshape(self.chain_world, nmax, niterations + 1, self.nwalkers, ndim + 1)
        out = {}
        for ichain in range(self.nchains):
            rank, slot = ichain % self.chain_world, ichain // self.chain_world
            out[ichain] = (gathered[rank, slot, :niterations, :, :ndim].copy(), gathered[rank, slot, :niterations, :, ndim].copy())
            self._accepted[ichain], self._iterations[ichain] = gathered[rank, slot, niterations, :, 0].copy(), int(gathered[rank, slot, niterations, 0, 1])
        return out

    def run(self, niterations=300, thin_by=1, start=None, min_iterations=0, max_iterations=None, check_every=None, check=None):
        """Run the chains.  ``niterations`` updates in one batch (``check_every=None``; cf. samplers/emcee.py:101-111), or batches of ``check_every`` updates until
        the convergence tests of :meth:`check` pass (not before ``min_iterations``) or ``max_iterations`` is reached (samplers/base.py:409-502); ``check``: ``True`` /
        dict of criteria / ``False``.  Chains are saved to ``save_fn`` after every batch (rank 0).  Returns the chain (dict name -> [niterations, nwalkers]) for one
        chain, the list of chains otherwise."""
        self._starts(start)
        if check_every is None:
            self._run_batch(int(niterations), thin_by=thin_by)
            if self.save_fn is not None: self.save()
        else:
            run_check = bool(check) or isinstance(check, dict)
            criteria = check if isinstance(check, dict) else {}

            def batch(niter):
                self._run_batch(niter, thin_by=thin_by)
                if self.save_fn is not None: self.save()
                return self.check(**criteria) if run_check else False

            _batch_iterate(batch, min_iterations=min_iterations, max_iterations=niterations if max_iterations is None else max_iterations, check_every=int(check_every))
        return self.chains[0] if self.nchains == 1 else self.chains

    def check(self, nsplits=4, burnin=0.5, stable_over=2, max_eigen_gr=0.03, max_diag_gr=None, max_geweke=None, max_geweke_pvalue=None, min_ess=None, reliable_ess=50,
              max_dact=None, min_eigen_gr=None, min_diag_gr=None, min_geweke=None, min_geweke_pvalue=None, max_ess=None, min_dact=None, quiet=True):
        """Convergence tests on the chains gathered so far (samplers/base.py:504-690; every rank holds every chain, so every rank gets the same answer): Gelman-Rubin
        (eigenvalues and diagonal) across the chains split in ``nsplits``, Geweke, integrated autocorrelation time per walker; each criterion must hold over
        ``stable_over`` consecutive calls.  Statistics are appended to ``self.diagnostics``.  (The reference's 'cl_diag_gr' -- Gelman-Rubin on interval limits -- is
        not computed.)"""
        from . import diagnostics as diag
        if not isinstance(self.diagnostics, diag.Diagnostics): self.diagnostics = diag.Diagnostics(self.diagnostics)
        d = self.diagnostics
        if any(not store for store in self._blocks): return False
        assert nsplits > 1
        names = self.varied_params.names()
        arrays = [store.coords for store in self._blocks]     # [niterations, nwalkers, ndim] per chain
        size = arrays[0].shape[0]
        if 0 < burnin < 1: burnin = int(burnin * size + 0.5)
        burnin = int(burnin)
        nsplits = int((nsplits + self.nchains - 1) / self.nchains)
        lensplits = (size - burnin) // nsplits
        split = [array[burnin + islab * lensplits:burnin + (islab + 1) * lensplits] for islab in range(nsplits) for array in arrays]
        if any(s.shape[0] < 1 for s in split): return False
        kw = dict(stable_over=stable_over, quiet=quiet, log=print)
        toret = True

        def attempt(func, default=np.nan):
            try: return func()
            except (ValueError, np.linalg.LinAlgError): return default

        eigen_gr = attempt(lambda: diag.gelman_rubin(split, method='eigen', check_valid='ignore').max() - 1.)
        toret &= d.add_test('eigen_gr', 'max eigen Gelman-Rubin - 1', eigen_gr, limits=(min_eigen_gr, max_eigen_gr), **kw)
        diag_gr = attempt(lambda: diag.gelman_rubin(split, method='diag').max() - 1.)
        toret &= d.add_test('diag_gr', 'max diag Gelman-Rubin - 1', diag_gr, limits=(min_diag_gr, max_diag_gr), **kw)
        all_geweke = attempt(lambda: diag.geweke(split, first=0.1, last=0.5))
        toret &= d.add_test('geweke', 'max Geweke', np.max(all_geweke), limits=(min_geweke, max_geweke), **kw)
        from scipy import stats
        pvalue = attempt(lambda: stats.normaltest(all_geweke, axis=None).pvalue)
        toret &= d.add_test('geweke_pvalue', 'Geweke p-value', pvalue, limits=(min_geweke_pvalue, max_geweke_pvalue), **kw)
        walkers = np.concatenate([np.moveaxis(array[burnin:], 1, 0) for array in arrays])            # one series per walker (samplers/base.py:646-651)
        iact = attempt(lambda: diag.integrated_autocorrelation_time(walkers), default=np.full(len(names), np.nan))
        d.add('iact', iact)
        nsamples = walkers.shape[1]
        toret &= d.add_test('iterations_over_iact', 'effective sample size = ({:d} iterations / integrated autocorrelation time)'.format(nsamples) + (' (reliable)' if reliable_ess * iact.max() < nsamples else ''),
                            nsamples / iact.max(), limits=(min_ess, max_ess), **kw)
        if len(d['iact']) >= 2:
            rel = np.abs(d['iact'][-2] / d['iact'][-1] - 1.).max()
            toret &= d.add_test('dact', 'max variation of integrated autocorrelation time', rel, limits=(min_dact, max_dact), **kw)
        return bool(toret)

    @property
    def acceptance_fraction(self):
        """Accepted fraction per walker of chain 0 (one chain), or ``[nchains, nwalkers]``."""
        out = np.array([self._accepted[ichain] / max(self._iterations[ichain], 1) for ichain in range(self.nchains)])
        return out[0] if self.nchains == 1 else out

    # ---- checkpoints -----------------------------------------------------------------------------------------------------------------------------------------
    def _chain_file(self, ichain):
        from .io import ChainFile
        attrs = {'sampler': self.name, 'nwalkers': self.nwalkers, 'a': self.a, 'iteration': int(self._iterations[ichain]), 'naccepted': np.asarray(self._accepted[ichain]).tolist(),
                 'counter_seed': None if self.counter_seeds is None else int(self.counter_seeds[ichain])}
        return ChainFile(dict(self.chains[ichain]), params={param.name: param for param in self.varied_params}, attrs=attrs)

    def save(self, fn=None):
        """Chains in the reference's checkpoint format (``Chain.save``, parameter.py:2164-2182: read by the reference's ``Chain.load`` and by :meth:`load`); the
        attributes carry what resuming the very same chain needs (iteration counter and key of the counter-based generator, accepted counts).  One file per chain
        (``fn``: a name for a single chain, a list, or a template with ``*``; default ``save_fn``); written by rank 0 of the chains' group."""
        if fn is None: fn = self.save_fn
        if fn is None: raise ValueError('provide a file name')
        if isinstance(fn, str): fn = [fn.replace('*', str(ichain)) for ichain in range(self.nchains)]
        if len(fn) != self.nchains: raise ValueError('provide one file name per chain')
        if self.chain_rank != 0 or (not self.chain_parallel and self.sharding.rank != 0): return
        for ichain, name in enumerate(fn):
            if self._blocks[ichain]: self._chain_file(ichain).save(name)

    def _load_one(self, ichain, source):
        from .io import ChainFile
        chain = source if hasattr(source, 'arrays') else ChainFile.load(source)
        names = self.varied_params.names()
        coords = np.stack([np.asarray(chain.arrays[name], dtype='f8') for name in names], axis=-1)
        logp = np.asarray(chain.arrays['logposterior'], dtype='f8')
        self._blocks[ichain] = _ChainStore()
        self._blocks[ichain].append(coords, logp)
        self._state[ichain] = (coords[-1].copy(), logp[-1].copy())
        self._holds.pop(ichain, None)            # whatever a runner holds is not this state: handed over at the next run
        attrs = chain.attrs
        self._iterations[ichain] = int(attrs.get('iteration', logp.shape[0]))
        nacc = attrs.get('naccepted', None)
        self._accepted[ichain] = np.zeros(self.nwalkers) if nacc is None else np.asarray(nacc, dtype='f8')
        key = attrs.get('counter_seed', None)
        if key is not None:
            if self.counter_seeds is None: self.counter_seeds = [None] * self.nchains
            self.counter_seeds[ichain] = int(key)
        runner = self._runners.pop(ichain, None)   # an ensemble keeps the key it was created with: the next run builds one on the loaded key
        if runner is not None and runner.device_resident: runner.ens.close()

    def load(self, fn):
        """Resume from files written by :meth:`save` (one name, a list, or a template with ``*``): the next :meth:`run` continues these chains -- positions,
        log-posteriors, generator key and counter are handed to new ensembles, so the continuation is the chain an uninterrupted run would give."""
        if isinstance(fn, str): fn = [fn.replace('*', str(ichain)) for ichain in range(self.nchains)]
        if len(fn) != self.nchains: raise ValueError('provide one file name per chain')
        for ichain, name in enumerate(fn):
            self._load_one(ichain, name)
        if self.counter_seeds is not None and any(key is None for key in self.counter_seeds):
            self.counter_seeds = None    # (files without keys, e.g. written by the reference: new keys are drawn)


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


class ImportanceSampler(_BatchEvaluator):
    """Importance-sample input chains with a (new) likelihood (desilike/samplers/importance.py): every point of every chain is evaluated -- here as ONE batch per chain
    on the device -- and ``aweight`` is multiplied by ``exp(logposterior - max logposterior)`` of the new likelihood (``subtract_input=True``: first divided by the same
    factor of the input chain's stored log-posterior).  As in the reference the chain's ``loglikelihood`` / ``logprior`` (and the solved / derived outputs) are
    replaced by the new likelihood's, its stored ``logposterior`` array is left as it was, fixed parameters are filled in, and ``size`` / ``nvaried`` / ``ndof`` go
    to ``attrs``; analytically solved parameters are marginalised (``solved_default = '.marg'``, importance.py:37).

    ``chains``: :class:`~desilike_amd.io.ChainFile`, path of a chain file (``.npy`` / ``.npz``, written here or by the reference), or a list of these."""
    name = 'importance'

    def __init__(self, likelihood, chains, save_fn=None, sharding=None):
        from .io import ChainFile
        self.likelihood = self.calculator = likelihood
        likelihood.solved_default = '.marg'
        self.varied_params = likelihood.varied_params
        if not len(self.varied_params):
            raise ValueError('No parameters to be varied!')
        self.sharding = sharding if sharding is not None else WalkerSharding()
        if isinstance(chains, (str, os.PathLike, ChainFile)): chains = [chains]
        self.input_chains = [ChainFile.load(str(chain)) if isinstance(chain, (str, os.PathLike)) else chain for chain in chains]
        self.save_fn = save_fn
        if save_fn is not None:
            if isinstance(save_fn, (str, os.PathLike)):
                self.save_fn = [str(save_fn).replace('*', '{}').format(i) for i in range(self.nchains)]
            elif len(save_fn) != self.nchains:
                raise ValueError('Provide {:d} chain file names'.format(self.nchains))
        self.chains = [None] * self.nchains

    @property
    def nchains(self):
        return len(self.input_chains)

    def run(self, subtract_input=False):
        from .io import ChainFile
        names = self.varied_params.names()
        for ichain, chain in enumerate(self.input_chains):
            arrays = {name: np.array(value) for name, value in chain.arrays.items()}
            shape = chain.shape

            def zero_lag(name):
                value = arrays[name]
                return value[..., 0] if name in chain.derivs else value

            aweight = np.array(arrays['aweight'], dtype='f8') if 'aweight' in arrays else np.ones(shape, dtype='f8')
            if subtract_input:     # importance.py:104-109 (a chain without a stored log-posterior counts as zeros: samples/chain.py:168-172)
                logposterior = zero_lag('logposterior') if 'logposterior' in arrays else np.zeros(shape, dtype='f8')
                mask = np.isfinite(logposterior)
                aweight = aweight / np.exp(logposterior - (logposterior[mask].max() if mask.any() else 0.))
            missing = [name for name in names if name not in arrays]
            if missing:
                raise KeyError('chain {:d} lacks the varied parameters {}'.format(ichain, missing))
            samples = Samples([np.asarray(arrays[name], dtype='f8') for name in names], params=self.varied_params)
            samples = self._evaluate(samples, errors='return')
            derivs = {name: value for name, value in chain.derivs.items()}
            for name in samples:
                arrays[name] = np.asarray(samples[name])
                derivs.pop(name, None)       # (zero-lag values: the Hessian entries of a marginalised input chain do not describe the new likelihood)
            for name in ['loglikelihood', 'logprior']:
                arrays[name] = np.where(np.isnan(arrays[name]), -np.inf, arrays[name])      # importance.py:141-143
            logposterior = arrays['loglikelihood'] + arrays['logprior']
            mask = np.isfinite(logposterior)
            with np.errstate(invalid='ignore'):
                aweight = aweight * np.exp(logposterior - (logposterior[mask].max() if mask.any() else 0.))
            arrays['aweight'] = aweight
            attrs = dict(chain.attrs)
            size = int(np.size(self.likelihood.flatdata))
            nvaried = len(names) + len(self.likelihood.solved_params)
            attrs.update(size=size, nvaried=nvaried, ndof=size - nvaried)                       # importance.py:149-155, likelihoods/base.py:445-457
            params = dict(chain.params)
            params.update({param.name: param for param in self.likelihood.all_params if param.name in arrays})
            self.chains[ichain] = ChainFile(arrays, params=params, derivs=derivs, attrs=attrs)
            if self.save_fn is not None:
                self.chains[ichain].save(self.save_fn[ichain])
        return self.chains

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        pass


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


def __getattr__(name):
    # the blocked Metropolis-Hastings sampler lives in its own module (which imports this one)
    if name in ('MCMCSampler', 'MHDraws'):
        from . import mcmc
        return getattr(mcmc, name)
    if name == 'HMCSampler':
        from . import hmc
        return hmc.HMCSampler
    raise AttributeError('module {!r} has no attribute {!r}'.format(__name__, name))
