"""Hamiltonian Monte Carlo on batches of chains, fed by the likelihood's analytic gradient ("gradient output to feed HMC / NUTS samplers").

The reference's ``HMCSampler`` (desilike/samplers/hmc.py) wraps ``blackjax.hmc`` -- velocity-Verlet integration with a fixed number of steps, step size and inverse mass
matrix from ``blackjax.window_adaptation`` -- around ``jax.value_and_grad`` of ONE chain's log-posterior.  Here the C chains of a process are the rows of one batch: a
leapfrog step is ONE call of ``dl_eval_logposterior_grad`` (log-posterior and analytic gradient of all chains: csrc/dl_fullshape_grad.h; central differences through
``dl_eval_logposterior`` where the context is outside the analytic gradient's scope) plus a few element-wise updates on tensors that never leave the device.

Warm-up (``adaptation``): dual averaging of the common step size towards ``target_acceptance_rate`` (Nesterov 2009 / Hoffman & Gelman 2014, as Stan and blackjax do)
and the inverse mass matrix from the positions of all chains over the second half of the warm-up -- with many chains in the batch one window is enough.
No third-party arithmetic is restated: the integrator and the Metropolis test are the textbook ones (Neal 2011), checked by what they must conserve and sample."""
import numpy as np

from .samplers import BasePosteriorSampler, _batch_iterate
from .parallel import WalkerSharding


class HMCSampler(BasePosteriorSampler):
    """``HMCSampler(likelihood, chains=64, adaptation=True, covariance=None, step_size=1e-3, num_integration_steps=60, divergence_threshold=1000, seed=None)`` with the
    arguments of the reference's sampler (samplers/hmc.py:24-70); ``run(min_iterations, max_iterations, check_every, check, thin_by)`` as samplers/base.py:409-502.

    adaptation : ``True`` / dict (``niterations`` default 300, ``target_acceptance_rate`` 0.8, ``is_mass_matrix_diagonal`` True) / ``False`` (use ``step_size`` and
        ``covariance`` as they are).
    covariance : inverse mass matrix: ``None`` (the parameters' ``proposal`` squared), array [ndim, ndim] or its diagonal, ``(names, matrix)``,
        :class:`~desilike_amd.profilers.Profiles`.
    chains : number of chains advanced together (one batch per leapfrog step), distributed over the ranks of the process group.
    gradient : 'auto' (analytic where the device context provides it, else central differences), 'analytic', 'finite'."""
    name = 'hmc'

    def __init__(self, likelihood, chains=64, adaptation=True, covariance=None, step_size=1e-3, num_integration_steps=60, divergence_threshold=1000., gradient='auto', save_fn=None, **kwargs):
        super(HMCSampler, self).__init__(likelihood, **kwargs)
        self.nchains = int(chains)
        if self.nchains < 1: raise ValueError('chains must be >= 1')
        self.chain_group = self.sharding.group if self.sharding.active and self.sharding.world > 1 else None
        self.sharding = WalkerSharding(group=False)
        self.chain_rank = self.chain_group.rank if self.chain_group is not None else 0
        self.chain_world = self.chain_group.world if self.chain_group is not None else 1
        if adaptation is True: adaptation = {}
        self.adaptation = None if adaptation is False or adaptation is None else dict(adaptation)
        self.step_size, self.num_integration_steps, self.divergence_threshold = float(step_size), int(num_integration_steps), float(divergence_threshold)
        if self.num_integration_steps < 1 or not self.step_size > 0.: raise ValueError('step_size and num_integration_steps must be positive')
        if gradient not in ('auto', 'analytic', 'finite'): raise ValueError('gradient must be one of auto, analytic, finite')
        self.gradient = gradient
        self.inverse_mass_matrix = self._initial_covariance(covariance)
        self.save_fn = save_fn
        self._store = None               # [n, nchains, ndim], [n, nchains]
        self._state = None               # (positions [nchains, ndim], log-posteriors [nchains]) of all chains
        self._adapted = self.adaptation is None
        self._generator = None
        self.diagnostics = {}
        self.hyp = None
        self.naccepted, self.ndivergent, self.niterations = np.zeros(self.nchains), np.zeros(self.nchains), 0

    def _initial_covariance(self, source):
        names = self.varied_params.names()
        cov = np.diag([float(param.proposal)**2 for param in self.varied_params])
        if source is None: return cov
        if isinstance(getattr(source, 'covariance', None), tuple): source = source.covariance
        if isinstance(source, (tuple, list)) and len(source) == 2 and np.ndim(source[1]) == 2 and np.ndim(source[0]) == 1 and isinstance(source[0][0], str):
            given = ([str(name) for name in source[0]], np.asarray(source[1], dtype='f8'))
        else:
            matrix = np.asarray(source, dtype='f8')
            if matrix.ndim == 1: matrix = np.diag(matrix)
            if matrix.shape != (len(names),) * 2: raise ValueError('covariance must have shape ({0:d}, {0:d}) or ({0:d},)'.format(len(names)))
            given = (names, matrix)
        index = [names.index(name) for name in given[0] if name in names]
        sub = [i for i, name in enumerate(given[0]) if name in names]
        cov[np.ix_(index, index)] = given[1][np.ix_(sub, sub)]
        return cov

    # ---- log-posterior and gradient of a batch, on the tensors' device -----------------------------------------------------------------------------------------
    def _device(self):
        import torch
        get_context = getattr(self.likelihood, '_get_posterior_context', None)
        if get_context is None: return torch.device('cpu')
        return torch.device('cuda', self.likelihood._get_context().device)

    def _value_and_grad(self, q):
        """q [C, P] tensor -> (log-posterior [C], gradient [C, P]); rows without a finite log-posterior get -inf and a zero gradient."""
        import torch
        P = q.shape[1]
        out = None
        if q.is_cuda and self.gradient != 'finite' and not len(getattr(self.likelihood, 'solved_params', [])):
            out = self.likelihood._get_context().eval_logposterior_grad(q.contiguous())
            if out is None and self.gradient == 'analytic': raise NotImplementedError('this likelihood is outside the analytic gradient: use gradient="finite"')
        if out is None:
            # central differences, ONE batch of C (2 P + 1) rows (steps: Parameter.delta, shortened where a prior bound is closer)
            lower = torch.as_tensor(np.array([param.delta[1] for param in self.varied_params]), dtype=q.dtype, device=q.device)
            upper = torch.as_tensor(np.array([param.delta[2] for param in self.varied_params]), dtype=q.dtype, device=q.device)
            lo = torch.as_tensor(np.array([param.prior.limits[0] for param in self.varied_params]), dtype=q.dtype, device=q.device)
            hi = torch.as_tensor(np.array([param.prior.limits[1] for param in self.varied_params]), dtype=q.dtype, device=q.device)
            lower, upper = torch.clamp(torch.minimum(lower, q - lo), min=0.), torch.clamp(torch.minimum(upper, hi - q), min=0.)
            points = q[:, None, :].repeat(1, 2 * P + 1, 1)
            index = torch.arange(P, device=q.device)
            points[:, 1 + 2 * index, index] -= lower
            points[:, 2 + 2 * index, index] += upper
            flat = points.reshape(-1, P).contiguous()
            if q.is_cuda:
                values = torch.empty(flat.shape[0], dtype=q.dtype, device=q.device)
                ctx, offset = self.likelihood._get_posterior_context()
                ctx.eval_logposterior(flat, values)
                values = values + offset
            else:
                values = torch.as_tensor(self.logposterior(flat.numpy()), dtype=q.dtype)
            values = values.reshape(-1, 2 * P + 1)
            out = (values[:, 0], (values[:, 2::2] - values[:, 1::2]) / (lower + upper))
        lp, grad = out
        # rows without a finite log-posterior: -inf and a zero gradient (two element-wise launches: every launch between two gradient calls is on the critical path)
        lp = torch.nan_to_num(lp, nan=-float('inf'), posinf=-float('inf'), neginf=-float('inf'))
        grad = torch.nan_to_num(grad, nan=0., posinf=0., neginf=0.)
        return lp, grad

    # ---- one transition of all chains ----------------------------------------------------------------------------------------------------------------------------
    def _transition(self, q, lp, grad, step_size, minv, chol_m):
        """Velocity-Verlet trajectory of ``num_integration_steps`` steps and the Metropolis test (Neal 2011, section 5.3.2): returns the new (q, lp, grad), the acceptance
        probabilities and the divergence flags."""
        import torch
        z = torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=self._generator)
        p = torch.linalg.solve_triangular(chol_m.T, z.T, upper=True).T if chol_m.ndim == 2 else z / chol_m     # momentum ~ N(0, M), M = inverse of `minv` = L L^T
        kinetic = lambda p: 0.5 * ((p @ minv) * p).sum(dim=1) if minv.ndim == 2 else 0.5 * (p * p * minv).sum(dim=1)
        h0 = -lp + kinetic(p)
        # velocity Verlet; the half kicks of consecutive steps are merged: kick(eps / 2), [drift, gradient, kick(eps)] x (L - 1), drift, gradient, kick(eps / 2)
        p = torch.add(p, grad, alpha=0.5 * step_size)
        qn, lpn, gn = q, lp, grad
        for istep in range(self.num_integration_steps):
            qn = torch.addcmul(qn, p, minv, value=step_size) if minv.ndim == 1 else torch.addmm(qn, p, minv, alpha=step_size)
            lpn, gn = self._value_and_grad(qn)
            p = torch.add(p, gn, alpha=step_size if istep + 1 < self.num_integration_steps else 0.5 * step_size)
        h1 = -lpn + kinetic(p)
        delta = h0 - h1
        delta = torch.where(torch.isnan(delta), torch.full_like(delta, -float('inf')), delta)
        divergent = ~(delta.abs() < self.divergence_threshold)
        prob = torch.clamp(torch.exp(torch.clamp(delta, max=0.)), max=1.)
        prob = torch.where(divergent & (delta < 0.), torch.zeros_like(prob), prob)
        accept = torch.rand(q.shape[0], dtype=q.dtype, device=q.device, generator=self._generator) < prob
        q = torch.where(accept[:, None], qn, q); grad = torch.where(accept[:, None], gn, grad); lp = torch.where(accept, lpn, lp)
        return q, lp, grad, prob, accept, divergent

    def _mass(self, device):
        import torch
        minv = np.asarray(self.inverse_mass_matrix, dtype='f8')
        if np.allclose(np.diag(np.diag(minv)), minv):            # samplers/hmc.py:9-15: keep a diagonal matrix as its diagonal
            d = torch.as_tensor(np.diag(minv).copy(), dtype=torch.float64, device=device)
            return d, torch.sqrt(d)                              # (momentum = z / sqrt(minv))
        L = np.linalg.cholesky(minv)
        return torch.as_tensor(minv, dtype=torch.float64, device=device), torch.as_tensor(L, dtype=torch.float64, device=device)

    def _warmup(self, q, lp, grad):
        """Step size by dual averaging, inverse mass matrix from the chains' positions over the second half (all local chains pooled)."""
        import torch
        a = self.adaptation
        niterations, target = int(a.get('niterations', 300)), float(a.get('target_acceptance_rate', 0.8))
        diagonal = bool(a.get('is_mass_matrix_diagonal', True))
        step_size = float(a.get('initial_step_size', self.step_size))
        mu, gamma, t0, kappa = np.log(10. * step_size), 0.05, 10., 0.75
        window = (niterations // 2, niterations - max(niterations // 6, 10))     # positions collected in [start, stop): then the mass matrix is set and the step size re-adapted
        hbar, logbar, count, collected = 0., np.log(step_size), 0, []
        minv, chol = self._mass(q.device)
        for it in range(niterations):
            q, lp, grad, prob, accept, divergent = self._transition(q, lp, grad, step_size, minv, chol)
            count += 1
            hbar = (1. - 1. / (count + t0)) * hbar + (target - float(prob.mean())) / (count + t0)
            logstep = mu - np.sqrt(count) / gamma * hbar
            eta = count**(-kappa)
            logbar = eta * logstep + (1. - eta) * logbar
            step_size = float(np.exp(logstep))
            if window[0] <= it < window[1]: collected.append(q.clone())
            if it == window[1] - 1 and collected:
                x = torch.cat(collected).cpu().numpy()
                if x.shape[0] > 2 * x.shape[1]:
                    cov = np.atleast_2d(np.cov(x, rowvar=False, ddof=1))
                    n, d = x.shape
                    cov = (n / (n + 5.)) * cov + 1e-3 * (5. / (n + 5.)) * np.eye(d)       # Stan's regularisation of the estimate
                    self.inverse_mass_matrix = np.diag(np.diag(cov)) if diagonal else cov
                    minv, chol = self._mass(q.device)
                step_size = float(np.exp(logbar))
                mu, hbar, logbar, count = np.log(10. * step_size), 0., np.log(step_size), 0
        self.step_size = float(np.exp(logbar))
        self.hyp = {'step_size': self.step_size, 'inverse_mass_matrix': np.asarray(self.inverse_mass_matrix).copy()}
        self._adapted = True
        return q, lp, grad

    # ---- chains ------------------------------------------------------------------------------------------------------------------------------------------------------
    def local_chains(self):
        return [ichain for ichain in range(self.nchains) if ichain % self.chain_world == self.chain_rank]

    @property
    def chains(self):
        """Per chain: dict name -> [n] (incl. 'logposterior'), or None before the first iteration."""
        if self._store is None: return [None] * self.nchains
        coords, logp = self._store
        out = []
        for c in range(self.nchains):
            chain = {param.name: coords[:, c, iparam] for iparam, param in enumerate(self.varied_params)}
            chain['logposterior'] = logp[:, c]
            out.append(chain)
        return out

    def _run_batch(self, niterations, thin_by=1):
        import torch
        device = self._device()
        local = self.local_chains()
        if self._generator is None:
            self._generator = torch.Generator(device=device)
            self._generator.manual_seed(int(self.rng.randint(0, 2**31 - 1)) + 7919 * self.chain_rank)
        q = torch.as_tensor(self._state[0][local], dtype=torch.float64, device=device).contiguous()
        lp, grad = self._value_and_grad(q)
        if not bool(torch.isfinite(lp).all()): raise ValueError('the log-posterior of a starting position is not finite')
        if not self._adapted:
            # a rank without chains (fewer chains than ranks) has nothing to adapt on: it would average NaNs into everybody's step size
            if len(local): q, lp, grad = self._warmup(q, lp, grad)
            if self.chain_group is not None:      # every rank adapted on its own chains: mean step size and mass matrix, weighted by the number of chains of the rank
                shape = np.asarray(self.inverse_mass_matrix).shape
                packed = np.concatenate([[float(len(local))], [self.step_size if len(local) else 0.], np.asarray(self.inverse_mass_matrix, dtype='f8').ravel() if len(local) else np.zeros(int(np.prod(shape)))])
                packed = np.asarray(self.chain_group.allgather(packed)).reshape(self.chain_world, -1)
                weights = packed[:, 0] / packed[:, 0].sum()
                mean = (weights[:, None] * packed[:, 1:]).sum(axis=0)
                self.step_size, self.inverse_mass_matrix = float(mean[0]), mean[1:].reshape(shape)
                # the hyper-parameters the run uses, on every rank (also one without chains): what save() writes into the chain attributes
                self.hyp = {'step_size': self.step_size, 'inverse_mass_matrix': np.asarray(self.inverse_mass_matrix).copy()}
                self._adapted = True
        minv, chol = self._mass(device)
        nrec = niterations // thin_by
        coords = torch.empty((nrec, len(local), q.shape[1]), dtype=torch.float64, device=device)
        logp = torch.empty((nrec, len(local)), dtype=torch.float64, device=device)
        nacc, ndiv = torch.zeros(len(local), dtype=torch.float64, device=device), torch.zeros(len(local), dtype=torch.float64, device=device)
        for it in range(nrec * thin_by):
            q, lp, grad, prob, accept, divergent = self._transition(q, lp, grad, self.step_size, minv, chol)
            nacc += accept.to(nacc.dtype); ndiv += divergent.to(ndiv.dtype)
            if (it + 1) % thin_by == 0: coords[it // thin_by], logp[it // thin_by] = q, lp
        coords, logp, nacc, ndiv = coords.cpu().numpy(), logp.cpu().numpy(), nacc.cpu().numpy(), ndiv.cpu().numpy()
        ndim = coords.shape[-1]
        if self.chain_group is not None:
            nmax = (self.nchains + self.chain_world - 1) // self.chain_world
            block = np.zeros((nmax, nrec + 1, ndim + 1))
            for slot in range(len(local)):
                block[slot, :nrec, :ndim], block[slot, :nrec, ndim], block[slot, nrec, 0], block[slot, nrec, 1] = coords[:, slot], logp[:, slot], nacc[slot], ndiv[slot]
            gathered = np.asarray(self.chain_group.allgather(block)).reshape(self.chain_world, nmax, nrec + 1, ndim + 1)
            coords, logp, nacc, ndiv = np.empty((nrec, self.nchains, ndim)), np.empty((nrec, self.nchains)), np.empty(self.nchains), np.empty(self.nchains)
            for c in range(self.nchains):
                b = gathered[c % self.chain_world, c // self.chain_world]
                coords[:, c], logp[:, c], nacc[c], ndiv[c] = b[:nrec, :ndim], b[:nrec, ndim], b[nrec, 0], b[nrec, 1]
        self.naccepted += nacc; self.ndivergent += ndiv; self.niterations += nrec * thin_by
        if nrec:
            self._state = (coords[-1].copy(), logp[-1].copy())
            self._store = (coords, logp) if self._store is None else (np.concatenate([self._store[0], coords]), np.concatenate([self._store[1], logp]))

    def run(self, min_iterations=0, max_iterations=None, check_every=300, check=None, thin_by=1, start=None):
        """Batches of ``check_every`` transitions of every chain until :meth:`check` passes or ``max_iterations``.  Returns the list of chains."""
        run_check = bool(check) or isinstance(check, dict)
        if max_iterations is None: max_iterations = np.iinfo('i8').max if run_check else check_every
        if start is not None:
            start = np.asarray(start, dtype='f8').reshape(self.nchains, len(self.varied_params))
            self._state = (start, None)
        elif self._state is None:
            self._state = self._get_start(self.nchains)
        criteria = check if isinstance(check, dict) else {}

        def batch(niterations):
            self._run_batch(niterations, thin_by=thin_by)
            if self.save_fn is not None: self.save()
            return self.check(**criteria) if run_check else False

        _batch_iterate(batch, min_iterations=min_iterations, max_iterations=max_iterations, check_every=int(check_every))
        return self.chains

    @property
    def acceptance_rate(self):
        return self.naccepted / max(self.niterations, 1)

    def check(self, nsplits=4, burnin=0.5, stable_over=2, max_eigen_gr=0.03, max_diag_gr=None, min_eigen_gr=None, min_diag_gr=None, quiet=True):
        """Gelman-Rubin (eigenvalues and diagonal) across the chains, each split in ``nsplits`` (samplers/base.py:504-600)."""
        from . import diagnostics as diag
        if not isinstance(self.diagnostics, diag.Diagnostics): self.diagnostics = diag.Diagnostics(self.diagnostics)
        d = self.diagnostics
        if self._store is None: return False
        coords = self._store[0]
        size = coords.shape[0]
        if 0 < burnin < 1: burnin = int(burnin * size + 0.5)
        nsplits = max(int((nsplits + self.nchains - 1) / self.nchains), 1)
        if nsplits * self.nchains < 2: return False
        lensplits = (size - int(burnin)) // nsplits
        if lensplits < 2: return False
        split = [coords[int(burnin) + islab * lensplits:int(burnin) + (islab + 1) * lensplits, c] for islab in range(nsplits) for c in range(self.nchains)]
        kw = dict(stable_over=stable_over, quiet=quiet, log=print)
        toret = True

        def attempt(func):
            try: return func()
            except (ValueError, np.linalg.LinAlgError): return np.nan

        toret &= d.add_test('eigen_gr', 'max eigen Gelman-Rubin - 1', attempt(lambda: diag.gelman_rubin(split, method='eigen', check_valid='ignore').max() - 1.), limits=(min_eigen_gr, max_eigen_gr), **kw)
        toret &= d.add_test('diag_gr', 'max diag Gelman-Rubin - 1', attempt(lambda: diag.gelman_rubin(split, method='diag').max() - 1.), limits=(min_diag_gr, max_diag_gr), **kw)
        d.add_test('acceptance_rate', 'mean acceptance rate', float(np.mean(self.acceptance_rate)), **kw)
        return bool(toret)

    def save(self, fn=None):
        """One file per chain in the reference's checkpoint format (``Chain.save``); the hyper-parameters of the warm-up travel in the attributes ('hyp', samplers/hmc.py:196)."""
        from .io import ChainFile
        if fn is None: fn = self.save_fn
        if fn is None: raise ValueError('provide a file name')
        if isinstance(fn, str): fn = [fn.replace('*', str(ichain)) for ichain in range(self.nchains)]
        if len(fn) != self.nchains: raise ValueError('provide one file name per chain')
        if self.chain_rank != 0 or self._store is None: return
        hyp = None if self.hyp is None else {'step_size': self.hyp['step_size'], 'inverse_mass_matrix': np.asarray(self.hyp['inverse_mass_matrix']).tolist()}
        for chain, name in zip(self.chains, fn):
            ChainFile(dict(chain), params={param.name: param for param in self.varied_params}, attrs={'sampler': self.name, 'hyp': hyp}).save(name)
