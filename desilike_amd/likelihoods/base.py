"""Gaussian likelihoods (reference: desilike/likelihoods/base.py -- ``BaseLikelihood`` 203-462,
``ObservablesGaussianLikelihood`` 504-712, ``SumLikelihood`` 715-732).

Host side: collects the observables' constants and parameters into one likelihood ``spec``, builds the
precision matrix exactly like the reference (block-wise inverse 617-619, scale_covariance 606-621, Hartlap 623-629,
Percival 633-656) and hands everything to ``dl_create``.  Evaluation (``__call__``, ``vmap``) is one
``dl_eval_batch`` call on the GPU; there is no CPU path.
"""
import numpy as np

from ..base import BaseCalculator, PipelineError
from ..parameter import Parameter, ParameterCollection, Samples
from .. import utils


class BaseLikelihood(BaseCalculator):
    """Base likelihood: ``loglikelihood`` / ``logprior`` derived parameters, ``__call__`` surface (likelihoods/base.py:203-245)."""
    name = None
    solved_default = '.marg'
    _attrs = ['loglikelihood', 'logprior']

    def _set_derived_params(self):
        self.name = self.init.get('name', self.name)
        for name in self._attrs:
            setattr(self, '_param_{}'.format(name), Parameter(basename=name, namespace=self.name or '', derived=True))

    # ---- parameters -------------------------------------------------------------------------------
    @property
    def all_params(self):
        self.initialize()
        return self._all_params

    @all_params.setter
    def all_params(self, config):
        """``likelihood.all_params = {'sn0': {'derived': '.marg'}} | {'*': {...}} | ParameterCollection | 'params.yaml'`` (base.py:1307-1310): update the pipeline's
        parameters by name, patterns and meta entries (:meth:`ParameterCollection.update_config`); the compiled contexts follow (:meth:`_check_params`)."""
        self.initialize()
        self._all_params.update_config(config)

    @property
    def varied_params(self):
        """Sampled parameters: varied, not derived, not solved (base.py:1273-1281)."""
        return ParameterCollection([param for param in self.all_params if param.varied and not param.solved and param.derived is False])

    @property
    def dependent_params(self):
        """Parameters computed from others by an expression, ``derived='{a} * {b}'`` (parameter.py:758-807, base.py:533-545): not sampled, reported as derived."""
        return ParameterCollection([param for param in self.all_params if param.depends])

    @property
    def solved_params(self):
        """Parameters solved analytically at every point ('.marg', '.best', '.auto'): likelihoods/base.py:262-271."""
        return ParameterCollection([param for param in self.all_params if param.solved and not param.derived.startswith('.prec')])

    @property
    def prec_params(self):
        """Parameters marginalised ONCE into the precision matrix ('.prec'): likelihoods/base.py:262-267."""
        return ParameterCollection([param for param in self.all_params if param.solved and param.derived.startswith('.prec')])

    def _set_speed(self, niterations=10, override=False, seed=42, batch=1024):
        """Measure and set the calculators' speed (base.py:695-735: ``calculator.runtime_info.speed`` = evaluations per second of the calculator's own part, from
        ``niterations`` evaluations at parameters drawn from their ``ref`` distributions).  Here an evaluation is a batch of ``batch`` points and the parts are the
        device kernels, timed by the events attached to their dispatch packets (``dl_profile_*``): the theory kernel is booked on the theory calculators (template
        included), the window / chi2 GEMM on the window calculators, the finalize (priors, status) on the likelihood.  Returns {calculator: speed}."""
        import torch
        self.initialize()
        ctx = self._get_context()
        rng = np.random.RandomState(seed)
        device = torch.device('cuda', ctx.device)
        groups = {'theory': [obs.wmatrix.theory for obs in self.observables], 'window_gemm': [obs.wmatrix for obs in self.observables], 'finalize': [self]}
        for calculators in groups.values():
            for calculator in calculators: calculator.runtime_info.monitor.reset()
        out = torch.empty(batch, dtype=torch.float64, device=device)
        for _ in range(int(niterations)):
            theta = np.column_stack([param.ref.sample(size=batch, random_state=rng) if param.ref.is_proper() else np.full(batch, param.value) for param in self.varied_params])
            th = torch.as_tensor(theta, dtype=torch.float64, device=device).contiguous()
            ctx.eval_logposterior(th, out)             # (untimed: clocks, caches)
            ctx.profile_enable(1)
            ctx.eval_logposterior(th, out)
            torch.cuda.synchronize(device)
            ms = ctx.profile_read()
            ctx.profile_enable(0)
            for name, calculators in groups.items():
                for calculator in calculators:
                    calculator.runtime_info.monitor.add(1e-3 * ms[name] / len(calculators), count=batch)
        speeds = {}
        for calculators in groups.values():
            for calculator in calculators:
                monitor = calculator.runtime_info.monitor
                if calculator.runtime_info.speed is None or override:
                    total = monitor.get('time', average=False)
                    calculator.runtime_info.speed = monitor.counter / total if total > 0. else 1e6       # (base.py:724-727)
                speeds[calculator] = calculator.runtime_info.speed
        return speeds

    # ---- evaluation -------------------------------------------------------------------------------
    def __call__(self, *args, return_derived=False, **kwargs):
        """``likelihood(**params)`` or ``likelihood(dict)`` -> loglikelihood + logprior (base.py:1194-1196, likelihoods/base.py:242-245)."""
        params = {}
        for arg in args:
            params.update(arg)
        params.update(kwargs)
        (logposterior, derived), errs = self._evaluate_dict({name: np.asarray(value, dtype='f8') for name, value in params.items()}, (), errors='return',
                                                            return_derived=True)
        self.loglikelihood = float(derived[self._param_loglikelihood][()])
        self.logprior = float(derived[self._param_logprior][()])
        toret = self.loglikelihood + self.logprior
        if return_derived:
            return toret, derived
        return toret

    def _evaluate_dict(self, flat, shape, errors='raise', return_derived=False):
        """Batched evaluation for :func:`desilike_amd.base.vmap`: dict name -> array[B] in, logposterior[shape] out."""
        raise NotImplementedError

    def __add__(self, other):
        return SumLikelihood(likelihoods=[self, other])

    def __radd__(self, other):
        if other == 0:
            return SumLikelihood(likelihoods=[self])
        return self.__add__(other)


class BaseGaussianLikelihood(BaseLikelihood):
    """Gaussian likelihood of a flat theory vector (likelihoods/base.py:465-501); concrete classes define the observables."""


def _collect_varied(observables):
    all_params = ParameterCollection()
    for obs in observables:
        for param in list(obs.wmatrix.theory._all_params()) + list(getattr(obs.wmatrix, '_extra_params', lambda: [])()):
            if param.name in all_params:
                if all_params[param.name] != param:
                    # same name used with different settings by two calculators: the first definition wins, like one shared pipeline parameter
                    continue
            else:
                all_params.set(param)
    return all_params


class ObservablesGaussianLikelihood(BaseGaussianLikelihood):
    """
    Gaussian likelihood of observables (likelihoods/base.py:504-712).

    Parameters
    ----------
    observables : list, BaseCalculator
    covariance : array, default=None
        Covariance (2D) of the concatenated observables; if ``None``, block-diagonal from each observable's own.
    scale_covariance : float, default=1.
    correct_covariance : str, dict, default='hartlap-percival2014'
        Applied only when the number of observations ``nobs`` is known (``{'nobs': nobs, 'correction': ...}``).
    precision : array, default=None
        Precision matrix (2D) or its diagonal (1D), used instead of the inverse covariance.
    device : int, default=None
        GPU ordinal (default: ``LOCAL_RANK`` env variable, else 0).
    """

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        self._set_derived_params()
        observables = init['observables']
        if not isinstance(observables, (list, tuple)):
            observables = [observables]
        self.observables = list(observables)
        for obs in self.observables:
            self._require(obs)
            obs.initialize()
        covariance, precision = init.get('covariance', None), init.get('precision', None)
        scale_covariance = init.get('scale_covariance', 1.)
        correct_covariance = init.get('correct_covariance', 'hartlap-percival2014')
        self.nobs = init.get('nobs', None)
        if isinstance(correct_covariance, dict):
            self.nobs = correct_covariance.get('nobs', self.nobs)
            correct_covariance = correct_covariance['correction']
        self.correct_covariance = correct_covariance
        sizes = [obs.wmatrix.size for obs in self.observables]
        size = sum(sizes)
        if covariance is not None and not isinstance(covariance, np.ndarray):
            from ..observables.galaxy_clustering import _containers
            if _containers.is_matrix_container(covariance):   # an lsstypes-like CovarianceMatrix (duck-typed: .value(), .observable), likelihoods/base.py:594-603
                covariance = _containers.read_covariance(covariance, self.observables)
        if covariance is None and precision is None:
            if all(getattr(obs, 'covariance', None) is not None for obs in self.observables):  # likelihoods/base.py:552-566
                covariance = np.zeros((size, size), dtype='f8')
                start = 0
                for obs, n in zip(self.observables, sizes):
                    covariance[start:start + n, start:start + n] = obs.covariance
                    start += n
                if self.nobs is None:
                    nobs = [getattr(obs, 'nobs', None) for obs in self.observables]
                    if all(nobs): self.nobs = int(np.mean(nobs))
            else:
                raise ValueError('Observables must have their own covariance if global covariance or precision matrix not provided')

        def check_matrix(matrix, name, allow_1d=False):
            if matrix is None:
                return None
            matrix = np.array(matrix, dtype='f8')
            if allow_1d and matrix.ndim == 1:
                if matrix.size != size: raise ValueError('{} diagonal must have size {:d}'.format(name, size))
                return matrix
            matrix = np.atleast_2d(matrix)
            if matrix.shape != (size, size):
                raise ValueError('based on provided observables, {} expected to be a matrix of shape ({:d}, {:d}), but found {}'.format(name, size, size, matrix.shape))
            return matrix

        self.precision = check_matrix(precision, 'precision', allow_1d=True)
        self.covariance = check_matrix(covariance, 'covariance')
        if self.covariance is not None:
            self.covariance = self.covariance * scale_covariance
            slices, start = [], 0
            for obs, n in zip(self.observables, sizes):
                slices.append(slice(start, start + n))
                obs.covariance = self.covariance[slices[-1], slices[-1]]
                start += n
            if self.precision is None:
                # block-inversion, as the reference (likelihoods/base.py:617-619, utils.py:561-599)
                self.precision = utils.blockinv([[self.covariance[sl1, sl2] for sl2 in slices] for sl1 in slices])
        else:
            self.precision = self.precision / scale_covariance
        if self.nobs is not None and 'hartlap' in self.correct_covariance:  # likelihoods/base.py:623-629
            nbins = self.precision.shape[0]
            self.hartlap2007_factor = (self.nobs - nbins - 2.) / (self.nobs - 1.)
            self.precision = self.precision * self.hartlap2007_factor
        self.precision_hartlap2007 = self.precision.copy()
        self._all_params = _collect_varied(self.observables)
        if self.nobs is not None and 'percival' in self.correct_covariance:  # likelihoods/base.py:633-656
            nbins = self.precision_hartlap2007.shape[0]
            A = 2. / (self.nobs - nbins - 1.) / (self.nobs - nbins - 4.)
            B = (self.nobs - nbins - 2.) / (self.nobs - nbins - 1.) / (self.nobs - nbins - 4.)
            nparams = len([param for param in self._all_params if param.varied and param.derived is False or param.solved])
            self.percival2014_factor = (1 + B * (nbins - nparams)) / (1 + A + B * (nparams + 1))
            self.precision = self.precision_hartlap2007 / self.percival2014_factor
        self._contexts, self._context_specs = {}, {}
        self._flatdata = None
        self._precision_input = self.precision
        self._initialized = True
        self._generate_data()
        return self

    # ---- compile ------------------------------------------------------------------------------------
    @property
    def device(self):
        import os
        device = self.init.get('device', None)
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', 0))
        return int(device)

    def _spec(self, fixed_values, flatdata_list, precision, drop_solved=False, vary_solved=False):
        """Nested likelihood spec flattened by ``_lib.fill_config`` into the C-ABI config keys (include/desilike_amd.h).
        ``drop_solved``: analytically solved parameters are treated as fixed (at ``fixed_values`` or their default value);
        ``vary_solved``: they (and '.prec' parameters) become sampled parameters, appended to the theta columns (Fisher, fisher.py:688-695)."""
        varied = self.varied_params
        if vary_solved:
            varied = ParameterCollection(list(varied) + [param for param in self.all_params if param.solved])
            drop_solved = True
        # parameters derived by an expression are extra columns of the device theta, filled by Context.expand (:meth:`_expand_theta`); no prior of their own
        # (the reference's logprior runs over varied_params only: likelihoods/base.py:231-236)
        dependents = self.dependent_params
        if vary_solved and len(dependents):
            raise NotImplementedError('Fisher algebra with parameters derived by an expression ({}) is not implemented'.format(dependents.names()))
        names = varied.names() + dependents.names()
        solved = ParameterCollection() if drop_solved else self.solved_params
        solved_names = solved.names()
        observables = []
        for obs, flatdata in zip(self.observables, flatdata_list):
            theory = obs.wmatrix.theory
            spec = obs._observable_spec(flatdata=flatdata)

            def resolve(pname, default):
                if pname in names:
                    return (names.index(pname), default)
                if pname in fixed_values:
                    return (-1, float(fixed_values[pname]))
                if pname in self._all_params:
                    return (-1, float(self._all_params[pname].value))
                return (-1, default)

            defaults = dict(qpar=1., qper=1., qiso=1., qap=1., df=1., dm=0., dn=0., sigmapar=0., sigmaper=0., b1X=1., b1Y=1., sn0=0., dbeta=1., sigmas=0., dres=1., sigmav=0., b2=0., bs=0., b3=0., fnl_loc=0., pX=1., pY=1., bphiX=1., bphiY=1., sigmasY=0., m=0.6, n=0.9, qto=1., dpto=1., bv=1., sigmau=0.)
            inputs = {}
            imap = dict(theory._input_map())
            window_pass = getattr(obs.wmatrix, '_pass_params', lambda: [])()   # systematic templates: pass-through columns appended by the window
            if window_pass: imap['pass'] = list(imap.get('pass', [])) + list(window_pass)
            for iname, pname in imap.items():
                if iname == 'ct':
                    res = [[resolve(pn, 0.) for pn in pair] for pair in pname]
                    if res: inputs['ct'] = ([[r[0] for r in pair] for pair in res], [[r[1] for r in pair] for pair in res])
                elif iname == 'band':   # band template: amplitudes relative to the fiducial (default 1)
                    res = [resolve(pn, 1.) for pn in pname]
                    inputs[iname] = ([r[0] for r in res], [r[1] for r in res])
                elif iname in ('sn', 'pass', 'x', 'ml'):
                    res = [resolve(pn, 0.) for pn in pname]
                    if res: inputs[iname] = ([r[0] for r in res], [r[1] for r in res])
                elif iname == 'vp':   # velocileptors 'pars': the reference's defaults are 0, except b1 (full_shape.py:1290-1293)
                    res = [resolve(pn, 1. if ip == 0 and pn == 'b1' else 0.) for ip, pn in enumerate(pname)]
                    inputs[iname] = ([r[0] for r in res], [r[1] for r in res])
                else:
                    inputs[iname] = resolve(pname, defaults[iname])
            spec['inputs'] = inputs
            if solved_names:
                # analytically solved parameters must enter the theory linearly: shot-noise like and counter terms (full_shape.py:545-550, 628-634)

                def sindex(pname):
                    return solved_names.index(pname) if pname in solved_names else -1

                marg = {}
                if 'sn0' in imap: marg['sn0'] = [sindex(imap['sn0'])]
                if 'sn' in imap and imap['sn']: marg['sn'] = [sindex(pn) for pn in imap['sn']]
                if 'pass' in imap and imap['pass']: marg['pass'] = [sindex(pn) for pn in imap['pass']]
                if 'vp' in imap: marg['vp'] = [sindex(pn) for pn in imap['vp']]
                if 'ct' in imap and imap['ct']: marg['ct'] = [[sindex(pn) for pn in pair] for pair in imap['ct']]
                for iname, pname in imap.items():
                    if iname in ('x', 'ml') and any(pn in solved_names for pn in pname) or iname not in ('sn0', 'sn', 'ct', 'pass', 'vp', 'x', 'ml', 'band') and pname in solved_names:
                        raise PipelineError('parameter {} cannot be solved analytically: the theory is not linear in it'.format(pname))
                spec['marg'] = marg
            observables.append(spec)
        from ..parameter import ParameterPrior
        # a varied dependent keeps its own prior, like in pipeline.params.prior (parameter.py:1889-1897)
        priors = np.array([param.prior.spec() for param in varied] + [(param.prior if param.varied else ParameterPrior()).spec() for param in dependents], dtype='f8').reshape(len(names), 5)
        spec = dict(n_params=np.array([len(names)], dtype='i4'), priors=priors, precision=precision, observables=observables)
        if len(dependents):
            spec['_expand'] = (self._make_expand(varied, dependents, fixed_values), len(varied))
        if solved_names:
            kind, mprior, x0 = [], [], []
            for param in solved:
                derived = param.derived
                if derived.startswith('.auto'):
                    derived = derived.replace('.auto', self.solved_default)                     # likelihoods/base.py:336-337
                kind.append(1 if derived.startswith('.marg') else 0)
                loc, scale = (param.prior.loc, param.prior.scale) if param.prior.dist == 'norm' else (0., np.inf)   # likelihoods/base.py:180-181
                mprior.append([loc, scale**(-2)])
                x0.append(float(fixed_values.get(param.name, param.value)))                  # likelihoods/base.py:355
            spec['marg'] = dict(kind=np.array(kind, dtype='i4'), prior=np.array(mprior, dtype='f8'), x0=np.array(x0, dtype='f8'))
        return spec

    def _make_expand(self, varied, dependents, fixed_values):
        """``theta [B, len(varied)] -> [B, len(varied) + len(dependents)]``: the columns of parameters derived by an expression, evaluated from the sampled columns
        and the values of fixed parameters (parameter.py:795-807; the pipeline does this before every calculation: base.py:536-539).  Works on numpy arrays
        and torch tensors alike; dependents may depend on other dependents (resolved in order, cycles are an error)."""
        names = varied.names()
        constants = {param.name: float(fixed_values.get(param.name, param.value)) for param in self._all_params if param.name not in names and not param.depends}
        order, done, pending = [], set(names) | set(constants), list(dependents)
        while pending:
            ready = [param for param in pending if all(name in done for name in param.depends)]
            if not ready:
                unknown = sorted(set(name for param in pending for name in param.depends if name not in self._all_params))
                if unknown:
                    raise PipelineError('parameters {} are derived from {}, which are not parameters of the pipeline: {}'.format([p.name for p in pending], unknown, self._all_params.names()))
                raise PipelineError('parameters {} are derived from each other'.format([param.name for param in pending]))
            for param in ready:
                order.append(param); done.add(param.name); pending.remove(param)
        columns = {param.name: len(names) + i for i, param in enumerate(dependents)}

        def expand(theta):
            values = dict(constants)
            for i, name in enumerate(names): values[name] = theta[:, i]
            extra = {}
            for param in order:
                values[param.name] = extra[columns[param.name]] = param.eval(**values)
            if isinstance(theta, np.ndarray):
                out = np.empty((theta.shape[0], len(names) + len(columns)), dtype='f8')
                out[:, :len(names)] = theta
                for col, value in extra.items(): out[:, col] = value
                return out
            import torch
            out = torch.empty((theta.shape[0], len(names) + len(columns)), dtype=theta.dtype, device=theta.device)
            out[:, :len(names)] = theta
            for col, value in extra.items(): out[:, col] = value
            return out

        return expand

    def _check_params(self):
        """Drop the compiled contexts if ``all_params`` changed since they were built (``likelihood.all_params['b1'].update(fixed=True)``, a new parameter ...):
        the reference re-reads its parameters at every call (base.py:533-539).  One integer comparison when nothing changed."""
        from ..parameter import generation
        current = generation()
        if current != getattr(self, '_params_generation', None):
            signature = tuple((param.name, repr(param.__getstate__())) for param in self._all_params)
            if signature != getattr(self, '_params_signature', signature):
                self._contexts, self._context_specs = {}, {}   # (dropped, not closed: a device-resident ensemble may still hold one; unreferenced contexts free their device memory themselves)
            self._params_signature, self._params_generation = signature, generation()

    MAX_CONTEXTS = 32

    def _evict(self):
        """Keep the cache of compiled contexts bounded: the oldest entries are DROPPED, never closed -- a device-resident ensemble / Metropolis-Hastings runner (one replica
        per chain or stream) or a caller of ``_get_context`` may still hold one; runners keep a reference to their context, and an unreferenced ``Context`` frees its device
        memory in ``__del__``.  The spec of a configuration goes with its last context."""
        while len(self._contexts) > self.MAX_CONTEXTS:
            self._contexts.pop(next(iter(self._contexts)))
        base = {key[2:] if key and key[0] == 'replica' else key for key in self._contexts}
        for key in [key for key in self._context_specs if key not in base]:
            del self._context_specs[key]

    def _replica(self, key, replica):
        """Context ``replica`` > 0 of the compiled configuration ``key``: same constants, its own device workspaces -- callers that keep several evaluations in flight
        on different HIP streams (chains of a chain-parallel sampler, pipelined batches) need one context per stream (calls on ONE context are serialised by contract:
        include/desilike_amd.h)."""
        from .._lib import Context
        rkey = ('replica', int(replica)) + key
        if rkey not in self._contexts:
            spec = self._context_specs[key]
            self._evict()
            self._context_specs[key] = spec
            self._contexts[rkey] = Context(spec, device=self.device)
        return self._contexts[rkey]

    def _get_context(self, fixed_values=None, replica=0):
        from .._lib import Context
        self.initialize()
        self._check_params()
        fixed_values = dict(fixed_values or {})
        key = tuple(sorted(fixed_values.items()))
        if key not in self._contexts:
            self._evict()
            flatdata, precision = self._flatdata_list(), self._precision_input
            if len(self.prec_params):
                flatdata, precision = self._marginalize_precision(fixed_values, flatdata, precision)
            self._context_specs[key] = self._spec(fixed_values, flatdata, precision)
            self._contexts[key] = Context(self._context_specs[key], device=self.device)
            # what the context holds, as the reference exposes it (likelihoods/base.py:308-309)
            self.precision, self.flatdata = precision, np.concatenate(flatdata)
        return self._contexts[key] if not replica else self._replica(key, replica)

    def _solved_are_constant(self):
        """True if the derivative of the theory vector w.r.t. every analytically solved parameter is the same at all points: shot-noise like terms and
        pass-through columns (broadband terms, systematic templates); counter terms and velocileptors parameters multiply point-dependent spectra."""
        solved_names = self.solved_params.names()
        if not solved_names:
            return False
        constant = set()
        for obs in self.observables:
            imap = obs.wmatrix.theory._input_map()
            if 'sn0' in imap: constant.add(imap['sn0'])
            constant.update(imap.get('sn', []))
            constant.update(imap.get('pass', []))
            constant.update(getattr(obs.wmatrix, '_pass_params', lambda: [])())
            if imap.get('vp', None):   # velocileptors: every solved parameter multiplies emulated tables
                constant.difference_update(imap['vp'])
        return all(name in constant for name in solved_names)

    def _get_posterior_context(self, fixed_values=None, replica=0):
        r"""Context and constant offset such that ``logposterior = ctx.eval_logposterior(theta) + offset`` (what samplers consume).

        When every analytically solved parameter has a point-independent derivative row T (:meth:`_solved_are_constant`), solving / marginalising them at each point
        (likelihoods/base.py:314-413) is the same as a plain Gaussian likelihood with the marginalised precision -- the one-off transformation the reference offers as
        '.prec' (257-312), here applied to '.marg' / '.best' parameters for the SUM loglikelihood + logprior:

            P' = L Q L^T,  Q = 1 - T~^T (T~ T~^T + D)^{-1} T~,  T~ = T L,  P = L L^T,  D = diag(1 / scale^2),
            data' = data + (x0 - loc) T  (Gaussian priors),   offset = -1/2 logdet (T~ T~^T + D)[marg, marg]

        so the marginalised fit runs through the chi2 GEMM at the cost of the non-marginalised one (no residual rows, no per-point Cholesky).  Q is singular for flat
        priors: the device takes the factor F = L V sqrt(lambda) of P' = F F^T (eigen-decomposition of Q) instead of a Cholesky factor.  Separate ``loglikelihood`` /
        ``logprior`` / solved values still come from the per-point path (:meth:`_get_context`)."""
        self.initialize()
        self._check_params()
        fixed_values = dict(fixed_values or {})
        if not len(self.solved_params):
            return self._get_context(fixed_values, replica=replica), 0.
        if not self._solved_are_constant():
            return self._get_context(fixed_values, replica=replica), 0.
        key = ('posterior',) + tuple(sorted(fixed_values.items()))
        if key not in self._contexts:
            from .._lib import Context
            flatdata_list, precision = self._flatdata_list(), self._precision_input
            if len(self.prec_params):
                flatdata_list, precision = self._marginalize_precision(fixed_values, flatdata_list, precision)
            solved = self.solved_params
            varied = self.varied_params
            theta = np.array([[float(fixed_values.get(param.name, param.value)) for param in varied]], dtype='f8')
            x0 = np.array([float(fixed_values.get(param.name, param.value)) for param in solved], dtype='f8')
            base = dict(fixed_values)
            base.update({param.name: value for param, value in zip(solved, x0)})

            def flattheory(fixed):
                ctx = Context(self._spec(fixed, flatdata_list, precision, drop_solved=True), device=self.device)
                flat = ctx.eval_batch_host(theta, return_flattheory=True)[3][0]
                ctx.close()
                return flat

            flat0 = flattheory(base)
            T = np.array([flattheory({**base, param.name: value + 1.}) - flat0 for param, value in zip(solved, x0)])        # [n_s, n]
            full = np.diag(precision) if precision.ndim == 1 else 0.5 * (precision + precision.T)
            L = np.linalg.cholesky(full)
            Tt = T.dot(L)
            loc = np.array([param.prior.loc if param.prior.dist == 'norm' else 0. for param in solved])
            prec = np.array([param.prior.scale**(-2) if param.prior.dist == 'norm' else 0. for param in solved])                 # likelihoods/base.py:180-183
            A = Tt.dot(Tt.T) + np.diag(prec)
            Q = np.eye(full.shape[0]) - Tt.T.dot(np.linalg.solve(A, Tt))
            lam, V = np.linalg.eigh(0.5 * (Q + Q.T))
            factor = L.dot(V * np.sqrt(np.clip(lam, 0., None)))
            shift = np.where(prec > 0., x0 - loc, 0.).dot(T)
            new_flatdata, start = [], 0
            for flatdata in flatdata_list:
                new_flatdata.append(flatdata + shift[start:start + flatdata.size])
                start += flatdata.size
            marg = np.array([(param.derived.replace('.auto', self.solved_default) if param.derived.startswith('.auto') else param.derived).startswith('.marg') for param in solved])
            offset = -0.5 * np.linalg.slogdet(A[np.ix_(marg, marg)])[1] if marg.any() else 0.                                     # likelihoods/base.py:394-404
            spec = self._spec(base, new_flatdata, factor.dot(factor.T), drop_solved=True)
            spec['precision_factor'] = factor
            self._evict()
            self._context_specs[key] = spec
            self._contexts[key] = Context(spec, device=self.device)
            self._posterior_offsets = getattr(self, '_posterior_offsets', {})
            self._posterior_offsets[key] = float(offset)
        return (self._contexts[key] if not replica else self._replica(key, replica)), self._posterior_offsets[key]

    def _marginalize_precision(self, fixed_values, flatdata_list, precision):
        r"""'.prec' parameters (likelihoods/base.py:257-312): linear parameters marginalised once, at the current values of the others, into

            P <- P - P T^T (T P T^T + diag(1 / scale^2))^{-1} T P ,     flatdata <- flatdata - \sum_i loc_i T_i ,

        T_i = d(flattheory) / d(p_i) (constant along p_i: the theory is linear in it).  The reference takes T from automatic differentiation; here the two
        evaluations flattheory(p_i = 1) - flattheory(p_i = 0) of the device theory give it exactly (to rounding).  Afterwards the parameter stays at its
        default value, like any fixed parameter.  A 1-D (diagonal) input precision becomes a full matrix (the reference's line 308 broadcasts the 1-D array
        against the 2-D correction instead: not reproduced)."""
        from .._lib import Context
        prec_params = self.prec_params
        varied = self.varied_params
        theta = np.array([[float(fixed_values.get(param.name, param.value)) for param in varied]], dtype='f8')   # pipeline.input_values (likelihoods/base.py:288)
        base = dict(fixed_values)
        for param in prec_params: base[param.name] = 0.                                                         # likelihoods/base.py:290

        def flattheory(fixed):
            ctx = Context(self._spec(fixed, flatdata_list, precision), device=self.device)
            flat = ctx.eval_batch_host(theta, return_flattheory=True)[3][0]
            ctx.close()
            return flat

        flat0 = flattheory(base)
        T = np.array([flattheory({**base, param.name: 1.}) - flat0 for param in prec_params])                    # [n_prec, n]
        if not np.isfinite(T).all():
            raise PipelineError("'.prec': non-finite theory at the current parameter values")
        full = np.diag(precision) if precision.ndim == 1 else precision
        derivp = T.dot(full)
        prior_hessian = np.array([-param.prior.scale**(-2) if param.prior.dist == 'norm' else 0. for param in prec_params])   # likelihoods/base.py:180-183
        posterior_hessian = -derivp.dot(T.T) + np.diag(prior_hessian)
        new_precision = full - derivp.T.dot(np.linalg.solve(-posterior_hessian, derivp))                        # likelihoods/base.py:308
        loc = np.array([getattr(param.prior, 'loc', 0.) if param.prior.dist == 'norm' else 0. for param in prec_params])
        shift = loc.dot(T)                                                                                      # flatdiff(loc) - flatdiff(0), likelihoods/base.py:309
        new_flatdata, start = [], 0
        for flatdata in flatdata_list:
            new_flatdata.append(flatdata - shift[start:start + flatdata.size])
            start += flatdata.size
        self.prec_derivatives = T
        return new_flatdata, new_precision

    def _flatdata_list(self):
        self.initialize()
        return [obs.flatdata for obs in self.observables]

    def _generate_data(self):
        """Observables given ``data=dict(params)`` take the theory evaluated at these parameters as data (power_spectrum.py:86-88)."""
        if not any(obs.flatdata is None for obs in self.observables):
            self.flatdata = np.concatenate(self._flatdata_list())
            return
        from .._lib import Context
        varied = self.varied_params
        sizes = [obs.wmatrix.size for obs in self.observables]
        zeros = [np.zeros(n) if obs.flatdata is None else obs.flatdata for obs, n in zip(self.observables, sizes)]
        for iobs, obs in enumerate(self.observables):
            if obs.flatdata is not None: continue
            data_params = obs._data_params
            fixed = {name: value for name, value in data_params.items() if name not in varied}
            spec = self._spec(fixed, zeros, np.ones(sum(sizes)))
            for ospec in spec['observables']:   # generated data = window output, BEFORE any observable transform (power_spectrum.py:95-97)
                ospec['transform'] = np.array([0], dtype='i4')
            ctx = Context(spec, device=self.device)
            theta = np.array([[data_params.get(param.name, param.value) for param in varied]], dtype='f8')
            flat = ctx.eval_batch_host(theta, return_flattheory=True)[3][0]
            ctx.close()
            start = sum(sizes[:iobs])
            obs.flatdata = flat[start:start + sizes[iobs]].copy()
        self.flatdata = np.concatenate(self._flatdata_list())

    # ---- evaluation -------------------------------------------------------------------------------------
    def _split_params(self, flat):
        """Separate the values of varied parameters (theta columns) from overrides of fixed parameters."""
        varied = self.varied_params
        unknown = [name for name in flat if name not in self._all_params]
        if unknown:
            raise PipelineError('Input parameter {} is not one of parameters: {}'.format(unknown[0], self._all_params.names()))
        size = max([np.size(value) for value in flat.values()] + [1])
        theta = np.empty((size, len(varied)), dtype='f8')
        for i, param in enumerate(varied):
            theta[:, i] = flat.get(param.name, param.value)
        fixed = {}
        for name, value in flat.items():
            if name not in varied:
                value = np.unique(value)
                if value.size != 1:
                    raise PipelineError('parameter {} is fixed: it takes one value per batch'.format(name))
                if float(value[0]) != self._all_params[name].value:
                    fixed[name] = float(value[0])
        return theta, fixed

    def _evaluate_dict(self, flat, shape, errors='raise', return_derived=False, return_flattheory=False):
        self.initialize()
        theta, fixed = self._split_params(flat)
        ctx = self._get_context(fixed)
        hessian = None
        if return_derived and ctx.n_solved and not return_flattheory:
            # derived outputs of a marginalised fit: solution and likelihood Hessian w.r.t. the solved parameters (likelihoods/base.py:361-368, 388-390)
            out = ctx.eval_batch_derived_host(theta)
            hessian = out[4]
            out = out[:4]
        else:
            out = ctx.eval_batch_host(theta, return_flattheory=return_flattheory, return_solved=ctx.n_solved > 0)
        loglike, logprior, status = out[:3]
        self._last_point = (theta, dict(fixed), shape)    # state of the last call: ``flattheory`` / ``flatdiff`` / observables' theory vectors are produced on demand
        self._flattheory = out[3].reshape(shape + (ctx.n_data,)) if return_flattheory else None
        errs = {}
        bad = np.flatnonzero(status >= 2)
        if bad.size:
            if errors == 'raise':
                raise PipelineError('non-finite loglikelihood / NaN input for points {}'.format(bad.tolist()))
            errs = {int(i): (FloatingPointError('non-finite likelihood (status {:d})'.format(int(status[i]))), '') for i in bad}
        logposterior = (loglike + logprior).reshape(shape)
        if return_derived:
            derived = Samples()
            derived[self._param_loglikelihood] = loglike.reshape(shape)
            derived[self._param_logprior] = logprior.reshape(shape)
            if ctx.expand is not None:   # parameters derived by an expression are reported with the derived parameters (base.py:541-545)
                for param, column in zip(self.dependent_params, ctx.expand(theta)[:, theta.shape[1]:].T):
                    derived[param] = column.reshape(shape)
            if ctx.n_solved:   # solution of the analytically solved parameters (likelihoods/base.py:361-368)
                for param, column in zip(self.solved_params, out[-1].T):
                    derived[param] = column.reshape(shape)
                if hessian is not None:
                    # the reference stores these as ParameterArray(loglikelihood, derivs=[(), (p1, p2), ...]) (likelihoods/base.py:388-390, 409-411); here: one
                    # array per pair, keyed (name, (p1, p2)) -> 'name.p1.p2'; the prior's Hessian is the constant -1 / scale^2 on the diagonal
                    names = self.solved_params.names()
                    for i1, p1 in enumerate(names):
                        for i2, p2 in enumerate(names[i1:], start=i1):
                            derived['{}.{}.{}'.format(self._param_loglikelihood, p1, p2)] = hessian[:, i1, i2].reshape(shape)
                    for param in self.solved_params:
                        scale = getattr(param.prior, 'scale', np.inf) if param.prior.dist == 'norm' else np.inf
                        derived['{}.{}.{}'.format(self._param_logprior, param.name, param.name)] = np.full(shape, -1. / scale**2 if np.isfinite(scale) else 0.)
            return (logposterior, derived), errs
        return logposterior, errs

    @property
    def flattheory(self):
        """Theory vector of the LAST call (likelihoods/base.py:658-664), at the default values of analytically solved parameters: evaluated on demand (the
        hot path does not write it unless asked)."""
        if getattr(self, '_flattheory', None) is None:
            if getattr(self, '_last_point', None) is None:
                raise AttributeError('flattheory is available after the likelihood has been called')
            theta, fixed, shape = self._last_point
            ctx = self._get_context(fixed)
            self._flattheory = ctx.eval_batch_host(theta, return_flattheory=True)[3].reshape(shape + (ctx.n_data,))
        return self._flattheory

    @flattheory.setter
    def flattheory(self, value):
        self._flattheory = value

    @property
    def flatdiff(self):
        """``flattheory - flatdata`` of the last call (likelihoods/base.py:659)."""
        return self.flattheory - self.flatdata

    @property
    def catch_errors(self):
        """Exception classes the samplers turn into -inf (likelihoods/base.py:247-255): none here, per-point failures are status codes of the C ABI."""
        return ()

    def observable_flattheory(self, iobs=0):
        """Slice of ``flattheory`` of observable ``iobs`` (what ``observable.flattheory`` holds in the reference, power_spectrum.py:400-404)."""
        sizes = [obs.wmatrix.size for obs in self.observables]
        start = sum(sizes[:iobs])
        return self.flattheory[..., start:start + sizes[iobs]]

    def theory_power(self, iobs=0):
        """Theory multipoles ``power [n_ell, n_k]`` of observable ``iobs`` at the last call (the ``power`` state of the reference's theory calculators,
        full_shape.py:502-510), through ``dl_eval_theory``."""
        if getattr(self, '_last_point', None) is None:
            raise AttributeError('theory_power is available after the likelihood has been called')
        theta, fixed, shape = self._last_point
        power = self._get_context(fixed).eval_theory_host(theta, iobs=iobs)
        return power.reshape(shape + power.shape[1:])

    def evaluate_batch(self, theta, loglike=None, logprior=None, status=None, flattheory=None, stream=None):
        """Fast path: ``theta`` is a float64 torch tensor [B, n_varied] resident on this likelihood's GPU
        (columns ordered as ``varied_params``); outputs are written asynchronously to the given tensors (allocated if ``None``).
        Returns (loglike, logprior, status)."""
        import torch
        ctx = self._get_context()
        B = theta.shape[0]
        if loglike is None: loglike = torch.empty(B, dtype=torch.float64, device=theta.device)
        if logprior is None: logprior = torch.empty(B, dtype=torch.float64, device=theta.device)
        if status is None: status = torch.empty(B, dtype=torch.int32, device=theta.device)
        ctx.eval_batch(theta, loglike=loglike, logprior=logprior, flattheory=flattheory, status=status, stream=stream)
        return loglike, logprior, status

    def evaluate_logposterior(self, theta, logposterior=None, status=None, stream=None):
        """Fast path for samplers resident on the GPU: ``theta`` float64 torch tensor [B, n_varied] on this likelihood's device -> ``logposterior [B]``
        (loglikelihood + logprior with the samplers' -inf conventions, samplers/base.py:144-200), asynchronous on ``stream``.  Linear parameters with constant
        derivative rows are marginalised once into the precision (:meth:`_get_posterior_context`)."""
        import torch
        ctx, offset = self._get_posterior_context()
        if logposterior is None: logposterior = torch.empty(theta.shape[0], dtype=torch.float64, device=theta.device)
        ctx.eval_logposterior(theta, logposterior, status=status, stream=stream)
        if offset != 0.:
            if stream is None:
                logposterior += offset
            else:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=theta.device)):
                    logposterior += offset
        return logposterior

    @property
    def size(self):
        return len(self.flatdata)

    @property
    def nvaried(self):
        return len(self.varied_params) + len(self.all_params.select(solved=True))

    @property
    def ndof(self):
        return self.size - self.nvaried


class SumLikelihood(BaseLikelihood):
    """Sum of independent likelihoods (likelihoods/base.py:715-732): Gaussian members are fused into ONE device context
    with a block-diagonal precision, so a batch is still a single GPU evaluation."""

    def initialize(self):
        if self._initialized:
            return self
        self._set_derived_params()
        likelihoods = self.init['likelihoods']
        if not isinstance(likelihoods, (list, tuple)): likelihoods = [likelihoods]
        flat = []
        for like in likelihoods:
            like.initialize()
            flat += like.likelihoods if isinstance(like, SumLikelihood) else [like]
        self.likelihoods = flat
        if not all(isinstance(like, ObservablesGaussianLikelihood) for like in flat):
            raise NotImplementedError('only sums of ObservablesGaussianLikelihood are supported on the GPU path')
        observables, blocks = [], []
        for like in flat:
            self._require(like)
            observables += like.observables
            prec = like.precision
            blocks.append(np.diag(prec) if prec.ndim == 1 else prec)
        from scipy import linalg
        self._fused = ObservablesGaussianLikelihood(observables=observables, precision=linalg.block_diag(*blocks), device=self.init.get('device', None),
                                                    correct_covariance='none', name=self.init.get('name', None))
        self._fused.initialize()
        self._all_params = self._fused._all_params
        self.flatdata = self._fused.flatdata
        self._initialized = True
        return self

    def _evaluate_dict(self, flat, shape, **kwargs):
        self.initialize()
        return self._fused._evaluate_dict(flat, shape, **kwargs)

    def evaluate_batch(self, *args, **kwargs):
        self.initialize()
        return self._fused.evaluate_batch(*args, **kwargs)

    def _get_context(self, fixed_values=None, replica=0):
        self.initialize()
        return self._fused._get_context(fixed_values, replica=replica)

    def _get_posterior_context(self, fixed_values=None, replica=0):
        self.initialize()
        return self._fused._get_posterior_context(fixed_values, replica=replica)

    def evaluate_logposterior(self, *args, **kwargs):
        self.initialize()
        return self._fused.evaluate_logposterior(*args, **kwargs)

    @property
    def size(self):
        return self._fused.size
