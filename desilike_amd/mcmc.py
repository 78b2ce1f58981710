"""Blocked Metropolis-Hastings sampler with the reference's constructor surface (desilike/samplers/mcmc.py: ``MCMCSampler`` 331-559 over ``MHSampler`` 25-127 and
``BlockProposer`` 199-328 -- the CosmoMC / cobaya fast-slow blocked proposal).

The reference advances one chain per group of MPI ranks and spends the ranks of the group on speculative proposals (``vectorize = mpicomm.size``, mcmc.py:531).  Here the
unit is the batch: ``chains`` x ``vectorize`` proposals are ONE call of the device likelihood; with a GPU likelihood the whole update (proposals, Metropolis scan, weights,
recorded samples, random draws) is resident on the device (``dl_mh_*``, csrc/dl_mh.hip), the host drains the new samples once per batch of ``check_every`` tries.
The random draws are counter-based (csrc/dl_mh.h; :class:`MHDraws` is the NumPy statement of the same functions): a chain is a pure function of (seed, chain index,
start, proposal covariance), whatever the rank or the driver that runs it.

Not built: dragging (mcmc.py:52-84: it spares evaluations of slow parameters; every parameter of a device likelihood costs the same launch)."""
import sys

import numpy as np

from .samplers import BasePosteriorSampler, CounterRNG, _batch_iterate
from .parallel import WalkerSharding


class MHDraws(object):
    """Counter-based draws of the blocked proposal (csrc/dl_mh.h) for chain ``chain``: cycler position, rotation column, radial scale and Metropolis exponential of
    proposer call ``n`` (``n = try * vectorize + slot``)."""
    PERM_A, PERM_B, RADIAL, RADIAL2, ACCEPT, ROT = 16, 17, 18, 19, 20, 21

    def __init__(self, seed, chain, blocks, oversample_factors=None):
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
        self.chain = int(chain)
        self.blocks = np.array(blocks, dtype='i4')
        self.oversample_factors = np.ones(len(self.blocks), dtype='i4') if oversample_factors is None else np.array(oversample_factors, dtype='i4')
        self.block_starts = np.insert(np.cumsum(self.blocks), 0, 0)
        self.rep_block = np.concatenate([np.full(b * o, ib, dtype='i4') for ib, (b, o) in enumerate(zip(self.blocks, self.oversample_factors))])   # mcmc.py:254-255
        self._cycle = (None, None)

    def _words(self, counters, stream):
        """uint32 [n, 4] for 64-bit ``counters`` [n] and 32-bit ``stream`` (scalar or [n])."""
        counters = np.atleast_1d(np.asarray(counters, dtype=np.uint64))
        c = np.empty((counters.size, 4), dtype=np.uint32)
        c[:, 0], c[:, 1] = (counters & np.uint64(0xFFFFFFFF)).astype(np.uint32), (counters >> np.uint64(32)).astype(np.uint32)
        c[:, 2], c[:, 3] = self.chain, stream
        return CounterRNG.philox4x32(c, self.key)

    def permutation(self, cycle):
        """The cycler's order in cycle ``cycle`` (mcmc.py:150-155): a keyed bijection of [0, n); in order for two entries or fewer."""
        if self._cycle[0] == cycle: return self._cycle[1]
        n = len(self.rep_block)
        if n <= 2:
            perm = np.arange(n)
        else:
            ka, kb = self._words([cycle], self.PERM_A)[0] | np.uint32(1), self._words([cycle], self.PERM_B)[0]
            bits = max(int(n - 1).bit_length(), 1)
            mask, shift = np.uint32((1 << bits) - 1), np.uint32((bits + 1) // 2)
            x = np.arange(n, dtype=np.uint32)
            todo = np.ones(n, dtype='?')
            while todo.any():
                y = x[todo]
                for r in range(4):
                    y = (y * ka[r] + kb[r]) & mask
                    y = y ^ (y >> shift)
                x[todo] = y
                todo[todo] = y >= n
            perm = x.astype(int)
        self._cycle = (cycle, perm)
        return perm

    def _gauss(self, m, ib, refl, size):
        elements = np.arange(size)
        streams = (self.ROT | (ib << 8) | (refl << 14) | ((elements >> 1) << 20)).astype(np.uint32)
        w = self._words(np.full(size, m, dtype=np.uint64), streams)
        rho = np.sqrt(-2. * np.log1p(-CounterRNG.uniform53(w[:, 0], w[:, 1])))
        phi = 2. * np.pi * CounterRNG.uniform53(w[:, 2], w[:, 3])
        return np.where(elements & 1, rho * np.sin(phi), rho * np.cos(phi))

    def rotation_column(self, m, ib, b, j):
        """Column ``j`` of rotation ``m`` of block ``ib``: Haar-distributed, as a product of Householder reflections of Gaussian vectors (Stewart 1980; the construction
        of scipy.stats.special_ortho_group that the reference calls, mcmc.py:172)."""
        y = np.zeros(b); y[j] = 1.
        signs = np.ones(b)
        for k in range(b - 2, -1, -1):
            x = self._gauss(m, ib, k, b - k)
            norm2, x0 = np.sum(x * x), x[0]
            signs[k] = -1. if x0 < 0. else 1.
            x[0] = x0 + signs[k] * np.sqrt(norm2)
            y[k:] -= 2. * x * (np.dot(x, y[k:]) / ((norm2 - x0 * x0) + x[0] * x[0]))
        signs[b - 1] = (-1.)**(b - 1) * np.prod(signs[:b - 1])
        return signs * y

    def direction(self, n):
        """(block, direction x radius) of proposer call ``n`` (mcmc.py:163-183, 270-277)."""
        nrep = len(self.rep_block)
        q, p = divmod(int(n), nrep)
        perm = self.permutation(q)
        ib = int(self.rep_block[perm[p]])
        b = int(self.blocks[ib])
        calls = q * b * int(self.oversample_factors[ib]) + int(np.sum(self.rep_block[perm[:p]] == ib))
        w = self._words([n], self.RADIAL)[0]
        mix, e = CounterRNG.uniform53(w[0:1], w[1:2])[0], -np.log1p(-CounterRNG.uniform53(w[2:3], w[3:4])[0])
        if b >= 2:
            radius = e if mix < 0.33 else np.sqrt(2. * e)                     # exponential | sqrt(chi2(2))
            return ib, self.rotation_column(calls // b, ib, b, calls % b) * radius
        w2 = self._words([n], self.RADIAL2)[0]
        g = np.sqrt(2. * e) * np.cos(2. * np.pi * CounterRNG.uniform53(w2[0:1], w2[1:2])[0])
        radius = e if mix < 0.33 else abs(g)                                  # exponential | sqrt(chi2(1))
        return ib, np.array([(1. if w2[2] & np.uint32(1) else -1.) * radius])

    def exponential(self, n):
        """Standard exponentials of the Metropolis tests of calls ``n`` (array)."""
        w = self._words(n, self.ACCEPT)
        return -np.log1p(-CounterRNG.uniform53(w[:, 0], w[:, 1]))


class _HostMH(object):
    """Chains driven on the host around a batched log-posterior function: the statement of csrc/dl_mh.hip in NumPy (same draws, same chain)."""
    device_resident = False

    def __init__(self, log_prob_fn, ndim, chain_ids, vectorize, blocks, oversample, order, proposal_scale, seed):
        self.log_prob_fn, self.ndim, self.vectorize, self.scale = log_prob_fn, int(ndim), int(vectorize), float(proposal_scale)
        self.order = np.asarray(order, dtype=int)
        self.draws = [MHDraws(seed, chain, blocks, oversample) for chain in chain_ids]
        self.block_starts = self.draws[0].block_starts
        self.nchains = len(self.draws)
        self.tries = 0

    def set_covariance(self, cholesky):
        self.L = np.array(cholesky, dtype='f8')

    def set_state(self, coords, logposterior=None, weight=None, naccepted=None, tries=0):
        self.coords = np.array(coords, dtype='f8')
        self.logp = None if logposterior is None else np.array(logposterior, dtype='f8')
        self.weight = np.ones(self.nchains, dtype='i8') if weight is None else np.array(weight, dtype='i8')
        self.naccepted = np.zeros(self.nchains, dtype='i8') if naccepted is None else np.array(naccepted, dtype='i8')
        self.fails = np.zeros(self.nchains, dtype='i4')
        self.tries = int(tries)

    def run(self, ntries, thin_by=1):
        C, V, P = self.nchains, self.vectorize, self.ndim
        if self.logp is None:
            self.logp = np.asarray(self.log_prob_fn(self.coords), dtype='f8')
            if not np.isfinite(self.logp).all(): raise ValueError('the log-posterior of a starting position is not finite')
        records = [([], [], []) for _ in range(C)]
        for t in range(self.tries, self.tries + ntries):
            prop = np.repeat(self.coords, V, axis=0)
            for c, draws in enumerate(self.draws):
                for v in range(V):
                    ib, direction = draws.direction(t * V + v)
                    start = self.block_starts[ib]
                    jump = self.L[start:, start:start + len(direction)].dot(direction * self.scale)       # mcmc.py:290-296
                    prop[c * V + v, self.order[start:]] += jump
            newlp = np.asarray(self.log_prob_fn(prop), dtype='f8')
            newlp[np.isnan(newlp)] = -np.inf
            for c, draws in enumerate(self.draws):
                lp = newlp[c * V:(c + 1) * V]
                e = draws.exponential(t * V + np.arange(V))
                accept = (lp > -np.inf) & ((lp > self.logp[c]) | (e > self.logp[c] - lp))                 # mcmc.py:107-112
                if accept.any():
                    first = int(np.argmax(accept))
                    if self.naccepted[c] > 0 and self.naccepted[c] % thin_by == 0:                          # mcmc.py:97-99
                        records[c][0].append(self.coords[c].copy()); records[c][1].append(self.logp[c]); records[c][2].append(self.weight[c] + first)
                    self.coords[c], self.logp[c], self.weight[c] = prop[c * V + first], lp[first], 1
                    self.naccepted[c] += 1
                    self.fails[c] = 0
                else:
                    self.weight[c] += V
                    self.fails[c] += 1
        self.tries += ntries
        return [(np.array(r[0], dtype='f8').reshape(len(r[0]), P), np.array(r[1], dtype='f8'), np.array(r[2], dtype='i8')) for r in records]

    def get_state(self):
        return self.coords.copy(), self.logp.copy(), self.weight.copy(), self.naccepted.copy(), self.fails.copy()

    def close(self):
        pass


class _DeviceMH(object):
    """Chains of this rank resident on the GPU (``dl_mh_*``), optionally on a HIP stream of their own."""
    device_resident = True

    def __init__(self, ctx, offset, chain_ids, vectorize, blocks, oversample, order, proposal_scale, seed, max_tries, own_stream=False):
        import torch
        from ._lib import DeviceMH
        self.mh = DeviceMH(ctx, len(chain_ids), vectorize=vectorize, blocks=blocks, oversample=oversample, order=order, chain_ids=chain_ids, proposal_scale=proposal_scale,
                           seed=seed, offset=offset, max_tries=max_tries)
        self.nchains = len(chain_ids)
        self.stream = torch.cuda.Stream(device=torch.device('cuda', self.mh.device)) if own_stream else None
        self._pending = None

    def _cuda_stream(self):
        return None if self.stream is None else self.stream.cuda_stream

    def set_covariance(self, cholesky):
        self.mh.set_covariance(cholesky, stream=self._cuda_stream())

    def set_state(self, coords, logposterior=None, weight=None, naccepted=None, tries=0):
        self.mh.set_state(coords, logposterior=logposterior, weight=weight, naccepted=naccepted, tries=tries, stream=self._cuda_stream())

    def enqueue(self, ntries, thin_by=1):
        import torch
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))
            with torch.cuda.stream(self.stream):      # (the record buffers are allocated and filled on the chains' stream)
                self._pending = self.mh.run(ntries, thin_by=thin_by, stream=self._cuda_stream())
        else:
            self._pending = self.mh.run(ntries, thin_by=thin_by)

    def collect(self):
        import torch
        coords, logp, weight, count = self._pending
        self._pending = None
        if self.stream is not None: self.stream.synchronize()
        count = count.cpu().numpy()                                                          # the one synchronisation of the batch
        nmax = int(count.max()) if count.size else 0
        coords, logp, weight = coords[:, :nmax].cpu().numpy(), logp[:, :nmax].cpu().numpy(), weight[:, :nmax].cpu().numpy()
        return [(coords[c, :count[c]], logp[c, :count[c]], weight[c, :count[c]]) for c in range(self.nchains)]      # views of the batch's host copy

    def run(self, ntries, thin_by=1):
        self.enqueue(ntries, thin_by=thin_by)
        return self.collect()

    def get_state(self):
        return self.mh.get_state(stream=self._cuda_stream())

    @property
    def tries(self):
        return self.mh.info('tries')

    def close(self):
        self.mh.close()


class _DeviceMHGroups(object):
    """The chains of this rank in ``len(parts)`` groups, each a :class:`_DeviceMH` with its own device context and HIP stream: the tries of the groups are enqueued
    one after the other and run concurrently -- the step kernel of one group (a single wavefront's critical path) in the shadow of another group's evaluation."""
    device_resident = True

    def __init__(self, parts):
        self.parts = parts                         # [(runner, slots of its chains in the rank's list)]
        self.nchains = sum(len(slots) for _, slots in parts)

    def set_covariance(self, cholesky):
        for runner, _ in self.parts: runner.set_covariance(cholesky)

    def set_state(self, coords, logposterior=None, weight=None, naccepted=None, tries=0):
        pick = lambda values, slots: None if values is None else np.asarray(values)[slots]
        for runner, slots in self.parts:
            runner.set_state(np.asarray(coords)[slots], logposterior=pick(logposterior, slots), weight=pick(weight, slots), naccepted=pick(naccepted, slots), tries=tries)

    def run(self, ntries, thin_by=1):
        for runner, _ in self.parts: runner.enqueue(ntries, thin_by=thin_by)
        out = [None] * self.nchains
        for runner, slots in self.parts:
            for slot, record in zip(slots, runner.collect()): out[slot] = record
        return out

    def get_state(self):
        states = [runner.get_state() for runner, _ in self.parts]
        out = []
        for i in range(5):
            merged = np.empty((self.nchains,) + states[0][i].shape[1:], dtype=states[0][i].dtype)
            for (runner, slots), state in zip(self.parts, states): merged[slots] = state[i]
            out.append(merged)
        return tuple(out)

    def close(self):
        for runner, _ in self.parts: runner.close()


class _ChainsView(object):
    """``sampler.chains``: a read-only sequence of per-chain dictionaries, built on access."""

    def __init__(self, sampler):
        self._sampler = sampler

    def __len__(self):
        return self._sampler.nchains

    def __getitem__(self, index):
        if isinstance(index, slice): return [self._sampler._chain_dict(i) for i in range(*index.indices(len(self)))]
        if index < 0: index += len(self)
        if not 0 <= index < len(self): raise IndexError(index)
        return self._sampler._chain_dict(index)

    def __iter__(self):
        return (self._sampler._chain_dict(i) for i in range(len(self)))


class _WeightedStore(object):
    """Recorded states of one chain, growing by batches: ``store[0]`` coords [n, ndim], ``store[1]`` log-posteriors [n], ``store[2]`` multiplicities [n].  A batch is
    appended as it comes (no copy of what is there); the pieces are joined when the arrays are asked for."""

    def __init__(self, coords, logp, weight):
        self._pieces, self._joined = [(np.asarray(coords, dtype='f8'), np.asarray(logp, dtype='f8'), np.asarray(weight, dtype='i8'))], None

    def append(self, coords, logp, weight):
        if self._joined is not None: self._pieces, self._joined = [self._joined], None
        self._pieces.append((coords, logp, np.asarray(weight, dtype='i8')))

    def __getitem__(self, index):
        if self._joined is None:
            self._joined = self._pieces[0] if len(self._pieces) == 1 else tuple(np.concatenate([piece[i] for piece in self._pieces]) for i in range(3))
            self._pieces = [self._joined]
        return self._joined[index]


def _format_blocks(blocks, names):
    """mcmc.py:316-328: ``[[oversample factor, [names]], ...]`` -> blocks of names sorted by ascending factor (slowest first), factors."""
    factors, groups = [int(block[0]) for block in blocks], [[str(name) for name in block[1] if str(name) in names] for block in blocks]
    factors, groups = [f for f, g in zip(factors, groups) if g], [g for g in groups if g]
    inblocks = [name for group in groups for name in group]
    if set(inblocks) != set(names) or len(inblocks) != len(names):
        raise ValueError('Missing (or repeated) sampled parameters in provided blocks: {}'.format(sorted(set(names) ^ set(inblocks))))
    argsort = np.argsort(factors, kind='stable')
    return [groups[i] for i in argsort], np.array([factors[i] for i in argsort], dtype='i4')


class MCMCSampler(BasePosteriorSampler):
    """``MCMCSampler(likelihood, blocks=None, covariance=None, proposal_scale=2.4, learn=True, chains=1, vectorize=None, ...)``; ``run(min_iterations, max_iterations,
    check_every, check, thin_by)`` as desilike/samplers/base.py:409-502, with ONE difference of unit: an *iteration* is a try (``vectorize`` proposals per chain, at most
    one accepted move), not an accepted move -- the batch is what the device executes; ``chains[i]['fweight']`` carries the multiplicities as in the reference.

    blocks : ``[[oversample_factor, [names]], ...]`` (mcmc.py:352-361); default one block of all parameters (a device likelihood has no fast / slow hierarchy).
    covariance : proposal covariance: ``None`` (the parameters' ``proposal`` squared), array [ndim, ndim] in the order of ``varied_params``, ``(names, array)``, or chain
        file(s) / :class:`~desilike_amd.io.ChainFile` to estimate it from (second half, weighted), or the :class:`~desilike_amd.profilers.Profiles` of a maximisation;
        parameters it does not cover take ``proposal`` squared.
    learn : update the proposal covariance from the chains before every batch (mcmc.py:467-497); a dict restricts it: ``{'every': '40 * ndim', 'max_eigen_gr': 0.1,
        'min_eigen_gr': 0.03, 'burnin': 0.5}``.
    chains : number of chains, or a list of files written by :meth:`save` to resume from.  Chains are distributed over the ranks of the process group (chain c on rank
        ``c % world``); all chains of a rank advance in the same batch.
    vectorize : speculative proposals per chain and try (mcmc.py:86-105); default: what fills a batch of 256 rows on a device likelihood, 1 on the host.
    streams : groups of chains that run concurrently on HIP streams of their own, each with its own device context (device-resident chains; default 1: one batch for
        all chains measured faster than half-size groups side by side); a chain does not depend on the grouping."""
    name = 'mcmc'

    def __init__(self, likelihood, blocks=None, oversample_power=0.4, covariance=None, proposal_scale=2.4, learn=True, drag=False, chains=1, vectorize=None,
                 device_resident=None, counter_seed=None, save_fn=None, streams=None, **kwargs):
        super(MCMCSampler, self).__init__(likelihood, **kwargs)
        if drag: raise NotImplementedError('dragging (mcmc.py:52-84) is not built: every parameter of a device likelihood costs the same launch')
        names = self.varied_params.names()
        ndim = len(names)
        if blocks is None: groups, factors = [list(names)], np.ones(1, dtype='i4')
        else: groups, factors = _format_blocks(blocks, names)
        self.blocks, self.oversample_factors = [len(group) for group in groups], factors
        self.sorted_names = [name for group in groups for name in group]
        self.order = np.array([names.index(name) for name in self.sorted_names], dtype='i4')       # sorted position -> column of the likelihood's varied parameters
        self.proposal_scale = float(proposal_scale)
        resume = None
        if not isinstance(chains, (int, np.integer)):
            resume = [chains] if isinstance(chains, (str, dict)) or hasattr(chains, 'arrays') else list(chains)
            chains = len(resume)
        self.nchains = int(chains)
        if self.nchains < 1: raise ValueError('chains must be >= 1')
        self.chain_group = self.sharding.group if self.sharding.active and self.sharding.world > 1 else None
        self.sharding = WalkerSharding(group=False)            # nothing is exchanged inside a chain: every evaluation is local
        self.chain_rank = self.chain_group.rank if self.chain_group is not None else 0
        self.chain_world = self.chain_group.world if self.chain_group is not None else 1
        if device_resident is None:
            device_resident = getattr(likelihood, '_get_posterior_context', None) is not None and not len(getattr(likelihood, 'dependent_params', []))
        self.device_resident = bool(device_resident)
        nlocal = len(self.local_chains())
        if vectorize is None: vectorize = max(1, min(64, 256 // max(nlocal, 1))) if self.device_resident else 1
        self.vectorize = int(vectorize)
        if streams is None: streams = 1      # (measured on the config-5 likelihood: two half-size groups are slower than one batch, profiles/r03u_mh_sampler.txt)
        self.streams = max(1, min(int(streams), max(nlocal, 1))) if self.device_resident else 1
        if not 1 <= self.vectorize <= 64: raise ValueError('vectorize must be in [1, 64]')
        if counter_seed is None:
            counter_seed = int(self.rng.randint(0, 2**32, dtype=np.uint64)) | (int(self.rng.randint(0, 2**32, dtype=np.uint64)) << 32)
        self.counter_seed = int(counter_seed) & 0xFFFFFFFFFFFFFFFF
        self.learn, self.learn_check = bool(learn), None
        burnin = 0.5
        if isinstance(learn, dict):
            self.learn, self.learn_check = True, dict(learn)
            burnin = self.learn_check['burnin'] = self.learn_check.get('burnin', burnin)
        self.learn_diagnostics = {}
        self._size_every = 0
        if save_fn is not None:
            if isinstance(save_fn, str): save_fn = [save_fn.replace('*', str(ichain)) for ichain in range(self.nchains)]
            save_fn = list(save_fn)
            if len(save_fn) != self.nchains or len(set(save_fn)) != self.nchains: raise ValueError('provide one file name per chain (or a template with *)')
        self.save_fn = save_fn
        self._store = [None] * self.nchains          # per chain: [coords [n, ndim], logposterior [n], fweight [n]] so far, on every rank
        self._state = [None] * self.nchains          # per chain: (coords, logposterior, weight, naccepted)
        self._tries = 0
        self._runner, self._runner_signature, self._handed = None, None, False
        self.diagnostics = {}
        self.covariance = self._initial_covariance(covariance, burnin)
        if resume is not None:
            for ichain, source in enumerate(resume): self._load_one(ichain, source)

    # ---- proposal covariance ---------------------------------------------------------------------------------------------------------------------------------
    def _initial_covariance(self, source, burnin=0.5):
        """load_source(..., cov=True) of the reference (mcmc.py:445-451): what the source does not provide is filled with ``proposal`` squared."""
        names = self.varied_params.names()
        ndim = len(names)
        cov = np.diag([float(param.proposal)**2 for param in self.varied_params])
        if source is None: return cov
        if isinstance(getattr(source, 'covariance', None), tuple): source = source.covariance      # profiles of a maximisation (profilers.Profiles): (names, matrix)
        given = None
        if isinstance(source, (tuple, list)) and len(source) == 2 and not hasattr(source[0], 'arrays') and np.ndim(source[1]) == 2:
            given = ([str(name) for name in source[0]], np.asarray(source[1], dtype='f8'))
        elif isinstance(source, np.ndarray) or (isinstance(source, (tuple, list)) and np.ndim(source) == 2):
            matrix = np.asarray(source, dtype='f8')
            if matrix.shape != (ndim, ndim): raise ValueError('covariance must have shape ({0:d}, {0:d})'.format(ndim))
            given = (names, matrix)
        else:
            from .io import ChainFile
            sources = [source] if isinstance(source, str) or hasattr(source, 'arrays') else list(source)
            files = [s if hasattr(s, 'arrays') else ChainFile.load(s) for s in sources]
            have = [name for name in names if all(name in f.arrays for f in files)]
            values, weights = [], []
            for f in files:
                x = np.stack([np.asarray(f.arrays[name], dtype='f8').ravel() for name in have], axis=-1)
                w = np.asarray(f.arrays['fweight'], dtype='f8').ravel() if 'fweight' in f.arrays else np.ones(x.shape[0])
                skip = int(burnin * x.shape[0] + 0.5) if 0 < burnin < 1 else int(burnin)
                values.append(x[skip:]); weights.append(w[skip:])
            given = (have, np.atleast_2d(np.cov(np.concatenate(values), rowvar=False, fweights=np.concatenate(weights).astype('i8'), ddof=1)))
        index = [names.index(name) for name in given[0] if name in names]
        sub = [i for i, name in enumerate(given[0]) if name in names]
        cov[np.ix_(index, index)] = given[1][np.ix_(sub, sub)]
        return cov

    def _cholesky_sorted(self, covariance):
        """BlockProposer.set_covariance (mcmc.py:298-313): symmetric positive definite, Cholesky factor in the sorted (block) order."""
        covariance = np.asarray(covariance, dtype='f8')
        if not (np.allclose(covariance.T, covariance) and np.all(np.linalg.eigvalsh((covariance + covariance.T) / 2.) > 0)):
            raise np.linalg.LinAlgError('The given covmat is not a positive-definite, symmetric square matrix.')
        return np.linalg.cholesky(covariance[np.ix_(self.order, self.order)])

    def _weighted_covariance(self, burnin=0.5):
        values, weights = [], []
        for store in self._store:
            n = store[0].shape[0]
            skip = int(burnin * n + 0.5) if 0 < burnin < 1 else int(burnin)
            values.append(store[0][skip:]); weights.append(store[2][skip:])
        values, weights = np.concatenate(values), np.concatenate(weights)
        if values.shape[0] < 2: return None
        return np.atleast_2d(np.cov(values, rowvar=False, fweights=weights, ddof=1))

    def _prepare(self):
        """mcmc.py:467-497: learn the proposal covariance from the chains (all of them, after burn-in) before a batch."""
        if not self.learn or any(store is None for store in self._store): return False
        burnin = 0.5
        if self.learn_check is not None:
            every = self.learn_check.get('every', None)
            if every is not None:
                if isinstance(every, str): every = int(eval(every, {'__builtins__': {}}, {'ndim': len(self.varied_params)}))
                size = sum(store[0].shape[0] for store in self._store)
                if size - self._size_every < int(every): return False
                self._size_every = size
            burnin = self.learn_check['burnin']
            criteria = {key: value for key, value in self.learn_check.items() if key not in ('every',)}
            if not self.check(**criteria, diagnostics=self.learn_diagnostics, quiet=True): return False
        covariance = self._weighted_covariance(burnin)
        if covariance is None: return False
        try:
            cholesky = self._cholesky_sorted(covariance)
        except np.linalg.LinAlgError:
            return False                                     # 'New proposal covariance is ill-conditioned, skipping update.' (mcmc.py:489-491)
        self.covariance = covariance
        if self._runner is not None: self._runner.set_covariance(cholesky)
        return True

    # ---- chains ----------------------------------------------------------------------------------------------------------------------------------------------
    def local_chains(self):
        return [ichain for ichain in range(self.nchains) if ichain % self.chain_world == self.chain_rank]

    def _chain_dict(self, ichain):
        store = self._store[ichain]
        if store is None: return None
        chain = {param.name: store[0][:, iparam] for iparam, param in enumerate(self.varied_params)}
        chain['fweight'], chain['logposterior'] = store[2], store[1]
        return chain

    @property
    def chains(self):
        """Per chain: dict name -> [n] with 'logposterior' and the multiplicities 'fweight' (mcmc.py:544-547), or None before the first recorded state.  A sequence
        that builds a chain's dictionary when it is asked for (hundreds of chains advance per batch: nothing per chain is done on the way)."""
        return _ChainsView(self)

    @property
    def chain(self):
        return self.chains[0]

    def _likelihood_signature(self):
        check = getattr(self.likelihood, '_check_params', None)
        if check is None: return None
        check()
        return getattr(self.likelihood, '_params_signature', None)

    def _get_runner(self):
        signature = self._likelihood_signature()
        if self._runner is not None and signature != self._runner_signature:
            self._runner.close()
            self._runner, self._handed = None, False
            for ichain, state in enumerate(self._state):       # the log-posteriors of the current positions belong to the old parameters
                if state is not None: self._state[ichain] = (state[0], None) + tuple(state[2:])
        self._runner_signature = signature
        if self._runner is None:
            local = self.local_chains()
            kw = dict(vectorize=self.vectorize, blocks=self.blocks, oversample=self.oversample_factors, order=self.order, proposal_scale=self.proposal_scale, seed=self.counter_seed)
            if self.device_resident and self.streams > 1:
                parts = []
                for igroup, slots in enumerate(np.array_split(np.arange(len(local)), self.streams)):
                    ctx, offset = self.likelihood._get_posterior_context(replica=igroup) if igroup else self.likelihood._get_posterior_context()
                    parts.append((_DeviceMH(ctx, offset, [local[slot] for slot in slots], max_tries=self.max_tries, own_stream=True, **kw), slots))
                self._runner = _DeviceMHGroups(parts)
            elif self.device_resident:
                ctx, offset = self.likelihood._get_posterior_context()
                self._runner = _DeviceMH(ctx, offset, local, max_tries=self.max_tries, **kw)
            else:
                self._runner = _HostMH(self.logposterior, len(self.varied_params), local, **kw)
            self._runner.set_covariance(self._cholesky_sorted(self.covariance))
            self._handed = False
        return self._runner

    def _starts(self, start=None):
        ndim = len(self.varied_params)
        if start is not None:
            start = np.asarray(start, dtype='f8').reshape(self.nchains, ndim)
            for ichain in range(self.nchains): self._state[ichain] = (start[ichain].copy(), None, 1, 0)
            self._tries, self._handed = 0, False
            return
        missing = [ichain for ichain in range(self.nchains) if self._state[ichain] is None]
        if missing:
            coords, logp = self._get_start(len(missing))       # drawn chain after chain from the synchronised generator (samplers/base.py:274-323)
            for ichain, x, lp in zip(missing, coords, logp): self._state[ichain] = (x, lp, 1, 0)
            self._handed = False

    def _run_batch(self, ntries, thin_by=1):
        if ntries <= 0:   # (max_iterations reached exactly at a resume: nothing to do -- and the gathered block keeps its record count in row 0, the state in row ntries)
            return
        runner = self._get_runner()
        self._prepare()
        local = self.local_chains()
        ndim = len(self.varied_params)
        if not self._handed:
            coords = np.array([self._state[ichain][0] for ichain in local])
            logp = [self._state[ichain][1] for ichain in local]
            logp = None if any(lp is None for lp in logp) else np.array(logp, dtype='f8')
            runner.set_state(coords, logposterior=logp, weight=[self._state[ichain][2] for ichain in local], naccepted=[self._state[ichain][3] for ichain in local], tries=self._tries)
            self._handed = True
        records = runner.run(ntries, thin_by=thin_by)
        coords, logp, weight, naccepted, fails = runner.get_state()
        if (np.asarray(fails) >= self.max_tries).any():
            raise ValueError('Could not find finite log posterior after {:d} tries'.format(self.max_tries))      # mcmc.py:102-103
        weight, naccepted, logp = np.asarray(weight).tolist(), np.asarray(naccepted).tolist(), np.asarray(logp, dtype='f8').tolist()     # (python scalars in bulk)
        new = {ichain: records[slot] + ((coords[slot], logp[slot], weight[slot], naccepted[slot]),) for slot, ichain in enumerate(local)}
        if self.chain_group is not None: new = self._gather(new, ntries, ndim)
        self._tries += ntries
        state, store = self._state, self._store
        for ichain in range(self.nchains):
            x, lp, w, current = new[ichain]
            state[ichain] = current
            if len(lp):
                if store[ichain] is None: store[ichain] = _WeightedStore(x, lp, w)
                else: store[ichain].append(x, lp, w)
        self.diagnostics['naccepted'] = [int(self._state[ichain][3]) for ichain in range(self.nchains)]

    def _gather(self, new, ntries, ndim):
        """All-gather of the batch: every rank ends up with every chain's new records and current state (one collective of [chains per rank, ntries + 1, ndim + 3])."""
        nmax = (self.nchains + self.chain_world - 1) // self.chain_world
        block = np.zeros((nmax, ntries + 1, ndim + 3), dtype='f8')
        for slot, ichain in enumerate(self.local_chains()):
            x, lp, w, state = new[ichain]
            n = x.shape[0]
            block[slot, :n, :ndim], block[slot, :n, ndim], block[slot, :n, ndim + 1] = x, lp, w
            block[slot, ntries, :ndim], block[slot, ntries, ndim], block[slot, ntries, ndim + 1], block[slot, ntries, ndim + 2] = state[0], state[1], state[2], state[3]
            block[slot, 0, ndim + 2] = n
        gathered = np.asarray(self.chain_group.allgather(block)).reshape(self.chain_world, nmax, ntries + 1, ndim + 3)
        out = {}
        for ichain in range(self.nchains):
            b = gathered[ichain % self.chain_world, ichain // self.chain_world]
            n = int(b[0, ndim + 2])                                      # (ntries >= 1: row 0 is a record row, the state sits in row ntries)
            state = (b[ntries, :ndim].copy(), float(b[ntries, ndim]), int(b[ntries, ndim + 1]), int(b[ntries, ndim + 2]))
            out[ichain] = (b[:n, :ndim].copy(), b[:n, ndim].copy(), b[:n, ndim + 1].astype('i8'), state)
        return out

    def run(self, min_iterations=0, max_iterations=None, check_every=300, check=None, thin_by=1, start=None):
        """Batches of ``check_every`` tries until the convergence tests of :meth:`check` pass (not before ``min_iterations``) or ``max_iterations`` tries are done
        (samplers/base.py:409-502).  Chains are saved to ``save_fn`` after every batch.  Returns the list of chains."""
        if max_iterations is None: max_iterations = sys.maxsize if (bool(check) or isinstance(check, dict)) else check_every
        self._starts(start)
        run_check = bool(check) or isinstance(check, dict)
        criteria = check if isinstance(check, dict) else {}

        def batch(ntries):
            self._run_batch(ntries, thin_by=thin_by)
            if self.save_fn is not None: self.save()
            return self.check(**criteria) if run_check else False

        _batch_iterate(batch, min_iterations=min_iterations, max_iterations=max_iterations, check_every=int(check_every))
        return self.chains

    @property
    def acceptance_rate(self):
        """Per chain: recorded states over the sum of their weights (mcmc.py:123-124)."""
        return np.array([store[2].size / store[2].sum() if store is not None and store[2].size else np.nan for store in self._store])

    def check(self, nsplits=4, burnin=0.5, stable_over=2, max_eigen_gr=0.03, max_diag_gr=None, max_geweke=None, max_geweke_pvalue=None, min_eigen_gr=None, min_diag_gr=None,
              min_geweke=None, min_geweke_pvalue=None, diagnostics=None, quiet=True):
        """Convergence tests on the weighted chains (samplers/base.py:504-690 with the chains' multiplicities as frequency weights, samples/diagnostics.py:78-84):
        Gelman-Rubin (eigenvalues and diagonal) across the chains split in ``nsplits``, Geweke; the acceptance rates are added as in mcmc.py:549-559.  Every rank
        holds every chain and gets the same answer."""
        from . import diagnostics as diag
        if diagnostics is None:
            if not isinstance(self.diagnostics, diag.Diagnostics): self.diagnostics = diag.Diagnostics(self.diagnostics)
            d = self.diagnostics
        else:
            d = diagnostics if isinstance(diagnostics, diag.Diagnostics) else diag.Diagnostics(diagnostics)
            if d is not diagnostics: self.learn_diagnostics = d
        if any(store is None for store in self._store): return False
        size = min(store[0].shape[0] for store in self._store)
        if 0 < burnin < 1: burnin = int(burnin * size + 0.5)
        burnin = int(burnin)
        nsplits = int((nsplits + self.nchains - 1) / self.nchains)
        assert nsplits * self.nchains > 1
        lensplits = (size - burnin) // nsplits
        if lensplits < 2: return False
        split = [store[0][burnin + islab * lensplits:burnin + (islab + 1) * lensplits] for islab in range(nsplits) for store in self._store]
        weights = [store[2][burnin + islab * lensplits:burnin + (islab + 1) * lensplits] for islab in range(nsplits) for store in self._store]
        kw = dict(stable_over=stable_over, quiet=quiet, log=print)
        toret = True

        def attempt(func, default=np.nan):
            try: return func()
            except (ValueError, np.linalg.LinAlgError): return default

        eigen_gr = attempt(lambda: diag.gelman_rubin(split, method='eigen', check_valid='ignore', weights=weights).max() - 1.)
        toret &= d.add_test('eigen_gr', 'max eigen Gelman-Rubin - 1', eigen_gr, limits=(min_eigen_gr, max_eigen_gr), **kw)
        diag_gr = attempt(lambda: diag.gelman_rubin(split, method='diag', weights=weights).max() - 1.)
        toret &= d.add_test('diag_gr', 'max diag Gelman-Rubin - 1', diag_gr, limits=(min_diag_gr, max_diag_gr), **kw)
        all_geweke = attempt(lambda: diag.geweke(split, first=0.1, last=0.5, weights=weights))
        toret &= d.add_test('geweke', 'max Geweke', np.max(all_geweke), limits=(min_geweke, max_geweke), **kw)
        from scipy import stats
        pvalue = attempt(lambda: stats.normaltest(all_geweke, axis=None).pvalue)
        toret &= d.add_test('geweke_pvalue', 'Geweke p-value', pvalue, limits=(min_geweke_pvalue, max_geweke_pvalue), **kw)
        d.add_test('total_acceptance_rate', 'total mean acceptance rate', float(np.nanmean(self.acceptance_rate)), **kw)
        return bool(toret)

    # ---- checkpoints -----------------------------------------------------------------------------------------------------------------------------------------
    def _chain_file(self, ichain):
        from .io import ChainFile
        state = self._state[ichain]
        attrs = {'sampler': self.name, 'tries': int(self._tries), 'counter_seed': int(self.counter_seed), 'vectorize': int(self.vectorize), 'chain_index': int(ichain),
                 'state': {'coords': np.asarray(state[0]).tolist(), 'logposterior': None if state[1] is None else float(state[1]), 'weight': int(state[2]), 'naccepted': int(state[3])},
                 'proposal_covariance': np.asarray(self.covariance).tolist()}
        return ChainFile(dict(self.chains[ichain]), params={param.name: param for param in self.varied_params}, attrs=attrs)

    def save(self, fn=None):
        """Chains in the reference's checkpoint format (``Chain.save``); the attributes carry what continuing the very same chains needs (current state and weight,
        try counter, key of the counter-based draws, proposal covariance).  One file per chain; written by rank 0."""
        if fn is None: fn = self.save_fn
        if fn is None: raise ValueError('provide a file name')
        if isinstance(fn, str): fn = [fn.replace('*', str(ichain)) for ichain in range(self.nchains)]
        if len(fn) != self.nchains: raise ValueError('provide one file name per chain')
        if self.chain_rank != 0: return
        for ichain, name in enumerate(fn):
            if self._store[ichain] is not None: self._chain_file(ichain).save(name)

    def _load_one(self, ichain, source):
        from .io import ChainFile
        chain = source if hasattr(source, 'arrays') else ChainFile.load(source)
        names = self.varied_params.names()
        coords = np.stack([np.asarray(chain.arrays[name], dtype='f8').ravel() for name in names], axis=-1)
        logp = np.asarray(chain.arrays['logposterior'], dtype='f8').ravel()
        weight = np.asarray(chain.arrays['fweight'], dtype='i8').ravel() if 'fweight' in chain.arrays else np.ones(logp.size, dtype='i8')
        self._store[ichain] = _WeightedStore(coords, logp, weight)
        attrs = chain.attrs
        state = attrs.get('state', None)
        if state is not None and attrs.get('sampler', None) == self.name:
            self._state[ichain] = (np.array(state['coords'], dtype='f8'), state['logposterior'], int(state['weight']), int(state['naccepted']))
            self._tries = int(attrs.get('tries', 0))
            if attrs.get('counter_seed', None) is not None: self.counter_seed = int(attrs['counter_seed'])
            if attrs.get('vectorize', None) is not None: self.vectorize = int(attrs['vectorize'])
            if attrs.get('proposal_covariance', None) is not None: self.covariance = np.array(attrs['proposal_covariance'], dtype='f8')
        else:
            self._state[ichain] = (coords[-1].copy(), float(logp[-1]), 1, 1)      # a chain of another sampler (or of the reference): continue from its last sample (samplers/base.py:274-280)
        self._handed = False
        if self._runner is not None:
            self._runner.close(); self._runner = None
