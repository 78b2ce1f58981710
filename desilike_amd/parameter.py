"""Parameters, priors and parameter collections: the host-side mirror of desilike's parameter API
(reference: desilike/parameter.py -- ``Parameter`` 654-1023, ``ParameterCollection`` 1657-1897,
``ParameterPrior`` 1908-2120) restricted to what the likelihood hot path and its callers need.

Only the *container* lives here; the prior arithmetic for batches runs on the GPU (``dl_finalize_kernel``),
``ParameterPrior.logpdf`` below is the scalar host version used for solved parameters and for sampling starts.
"""
import copy

import numpy as np

namespace_delimiter = '.'
ALLOWED_SOLVED = ['.best', '.marg', '.auto', '.best_not_derived', '.marg_not_derived', '.auto_not_derived', '.prec']


class ParameterError(Exception):
    pass


# prior kinds of the device table (csrc/dl_prior.h): uniform, norm, then scipy.stats location-scale families whose density is finite and non-zero at loc
PRIOR_KINDS = ['uniform', 'norm', 'expon', 'laplace', 'cauchy', 'logistic', 'halfnorm', 'halfcauchy', 'gumbel_r', 'gumbel_l']


class ParameterPrior(object):
    """1D prior (parameter.py:1908-2017): 'uniform' (possibly improper) or 'norm', optionally truncated to ``limits``; or one of the location-scale families of
    scipy.stats in :data:`PRIOR_KINDS` with explicit ``loc`` (and ``scale``, default 1), evaluated like the reference as ``rv.logpdf(x) - rv.logpdf(loc)`` (2012-2016).
    Not reproduced: limits on those families (the reference maps them onto scipy's ``trunc<dist>`` with the arguments of ``truncnorm``, 1958-1963) and a missing
    ``loc`` (the reference then removes ``logpdf(mean(limits)) = logpdf(nan)``)."""

    def __init__(self, dist='uniform', limits=None, **kwargs):
        if isinstance(dist, ParameterPrior):
            self.__dict__.update(copy.deepcopy(dist.__dict__))
            return
        limits = list(limits) if limits is not None else [-np.inf, np.inf]
        if limits[0] is None: limits[0] = -np.inf
        if limits[1] is None: limits[1] = np.inf
        if limits[1] <= limits[0]:
            raise ParameterError('ParameterPrior range {} has min greater than max'.format(limits))
        self.limits = (float(limits[0]), float(limits[1]))
        self.dist = str(dist)
        if self.dist.startswith('trunc'): self.dist = self.dist[5:]
        if self.dist not in PRIOR_KINDS:
            raise ParameterError('prior distribution must be one of {}, found {}'.format(PRIOR_KINDS, self.dist))
        self.attrs = {name: float(value) for name, value in kwargs.items()}
        if self.dist == 'norm':
            self.attrs.setdefault('loc', 0.)
            self.attrs.setdefault('scale', 1.)
        elif self.dist != 'uniform':
            if self.is_limited():
                raise ParameterError('limits are supported for "uniform" and "norm" priors only, found {} with limits {}'.format(self.dist, self.limits))
            if 'loc' not in self.attrs:
                raise ParameterError('prior {} needs an explicit loc: its log-density is reported relative to the value at loc (parameter.py:2012-2016)'.format(self.dist))
            self.attrs.setdefault('scale', 1.)
            unknown = [name for name in self.attrs if name not in ('loc', 'scale')]
            if unknown:
                raise ParameterError('prior {} takes loc and scale only, found {}'.format(self.dist, unknown))
        if self.dist != 'uniform' and not self.attrs['scale'] > 0.:
            raise ParameterError('the scale of a prior must be positive')

    def copy(self):
        return ParameterPrior(self)

    @property
    def loc(self):
        if self.dist != 'uniform': return self.attrs['loc']
        raise AttributeError('uniform distribution has no loc')

    @property
    def scale(self):
        if self.dist != 'uniform': return self.attrs['scale']
        raise AttributeError('uniform distribution has no scale')

    def _rv(self):
        from scipy import stats
        return getattr(stats, self.dist)(loc=self.attrs['loc'], scale=self.attrs['scale'])

    def is_proper(self):
        return self.dist != 'uniform' or not np.isinf(self.limits).any()

    def is_limited(self):
        return not np.isinf(self.limits).all()

    def isin(self, x):
        x = np.asarray(x)
        return (self.limits[0] < x) & (x < self.limits[1])

    def logpdf(self, x, remove_zerolag=True):
        """Log-density with its maximum removed (parameter.py:1994-2017); limits are closed."""
        x = np.asarray(x, dtype='f8')
        isin = (self.limits[0] <= x) & (x <= self.limits[1])
        if self.dist == 'uniform':
            toret = np.where(isin, 0., -np.inf)
            if not remove_zerolag and self.is_proper():
                toret = toret - np.log(self.limits[1] - self.limits[0])
            return toret
        loc, scale = self.attrs['loc'], self.attrs['scale']
        if self.dist != 'norm':   # parameter.py:2012-2016
            rv = self._rv()
            with np.errstate(divide='ignore'):
                toret = rv.logpdf(x)
                if remove_zerolag: toret = toret - rv.logpdf(loc)
            return np.where(isin, toret, -np.inf)
        toret = np.where(isin, -0.5 * (x - loc)**2 / scale**2, -np.inf)
        if not remove_zerolag:
            from scipy import special
            norm = 0.5 * (special.erf((self.limits[1] - loc) / scale / 2**0.5) - special.erf((self.limits[0] - loc) / scale / 2**0.5))
            toret = toret - np.log(scale * np.sqrt(2. * np.pi) * norm)
        return toret

    __call__ = logpdf

    def center(self):
        if self.dist != 'uniform':
            return self.attrs['loc']
        if self.is_limited():
            return float(np.mean([lim for lim in self.limits if not np.isinf(lim)]))
        return 0.

    def std(self):
        if self.dist == 'norm':
            return self.attrs['scale']
        if self.dist != 'uniform':
            return float(self._rv().std())
        if not self.is_proper():
            raise AttributeError('improper uniform distribution has no std')
        return (self.limits[1] - self.limits[0]) / 12.**0.5

    def sample(self, size=None, random_state=None):
        """Draw from the (truncated) distribution, as ``Parameter.ref.sample`` is used by samplers (samplers/base.py:222-230)."""
        if not self.is_proper():
            raise ParameterError('Cannot sample from improper prior')
        rng = random_state if isinstance(random_state, (np.random.RandomState, np.random.Generator)) else np.random.RandomState(random_state)
        if self.dist == 'uniform':
            return rng.uniform(self.limits[0], self.limits[1], size=size)
        loc, scale = self.attrs['loc'], self.attrs['scale']
        if self.dist != 'norm':
            return self._rv().rvs(size=size, random_state=rng)
        if not self.is_limited():
            return loc + scale * rng.standard_normal(size=size)
        from scipy import stats
        a, b = ((lim - loc) / scale for lim in self.limits)
        return stats.truncnorm(a, b, loc=loc, scale=scale).rvs(size=size, random_state=rng)

    def spec(self):
        """Row (kind, lo, hi, loc, scale) of the C-ABI ``priors`` table (include/desilike_amd.h)."""
        if self.dist != 'uniform':
            return [float(PRIOR_KINDS.index(self.dist)), self.limits[0], self.limits[1], self.attrs['loc'], self.attrs['scale']]
        return [0., self.limits[0], self.limits[1], 0., 1.]

    def __getstate__(self):
        return {'dist': self.dist, 'limits': self.limits, **self.attrs}

    def __setstate__(self, state):
        self.__init__(**state)

    def __repr__(self):
        base = self.dist
        if self.is_limited():
            base = '{}[{}, {}]'.format(base, *self.limits)
        return '{}({})'.format(base, self.attrs)

    def __eq__(self, other):
        return type(other) == type(self) and (self.dist, self.limits, self.attrs) == (other.dist, other.limits, other.attrs)


class Parameter(object):
    """One parameter (parameter.py:654-1023): ``basename``, ``namespace``, ``value``, ``fixed``, ``derived`` ('.marg' etc. to solve), ``prior``, ``ref``."""

    _attrs = ['basename', 'namespace', 'value', 'fixed', 'derived', 'prior', 'ref', 'proposal', 'delta', 'latex']

    def __init__(self, basename, namespace='', value=None, fixed=None, derived=False, prior=None, ref=None, proposal=None, delta=None, latex=None):
        if isinstance(basename, Parameter):   # (a copy changes nothing anybody compiled)
            self.__dict__.update(copy.deepcopy(basename.__dict__))
            return
        _generation[0] += 1
        if isinstance(basename, dict):
            state = dict(basename)
            if 'name' in state: state['basename'] = state.pop('name')
            self.__init__(**state)
            return
        names = str(basename).split(namespace_delimiter)
        self._basename = names[-1]
        parts = [str(namespace)] if namespace else []
        parts += names[:-1]
        self._namespace = namespace_delimiter.join(part for part in parts if part)
        self._value = float(value) if value is not None else None
        self._prior = prior if isinstance(prior, ParameterPrior) else ParameterPrior(**(prior or {}))
        self._ref = (ref if isinstance(ref, ParameterPrior) else ParameterPrior(**ref)) if ref is not None else self._prior.copy()
        self._proposal, self._latex = proposal, latex
        if delta is not None and np.ndim(delta) == 0:
            delta = (delta,) * 2
        self._delta = None if delta is None else tuple(delta)
        if isinstance(derived, str):
            if derived in ALLOWED_SOLVED:
                if self._prior.dist not in ('norm', 'uniform') or self._prior.is_limited():      # parameter.py:762-764
                    raise ParameterError('Prior must be "norm" or "uniform" with no limits to use analytic marginalisation for {}'.format(self._basename))
            elif not self._placeholders(derived):
                raise ParameterError('derived must be one of {} or an expression of other parameters in braces, e.g. "{{a}} + {{b}}" (parameter.py:758-776); found {!r}'.format(ALLOWED_SOLVED, derived))
            self._derived = derived
        else:
            self._derived = bool(derived)
        if fixed is None:
            fixed = prior is None and ref is None and not self.depends     # parameter.py:778-779
        self._fixed = bool(fixed)

    basename = property(lambda self: self._basename)
    namespace = property(lambda self: self._namespace)
    prior = property(lambda self: self._prior)
    ref = property(lambda self: self._ref)
    derived = property(lambda self: self._derived)
    fixed = property(lambda self: self._fixed)
    varied = property(lambda self: not self._fixed)
    limits = property(lambda self: self._prior.limits)

    @property
    def name(self):
        return namespace_delimiter.join([self._namespace, self._basename]) if self._namespace else self._basename

    @property
    def value(self):
        return self._value if self._value is not None else self._ref.center()

    @property
    def proposal(self):
        return self._proposal if self._proposal is not None else self._ref.std()

    @property
    def delta(self):
        delta = self._delta
        if delta is None:
            delta = (1e-1 * self.proposal,) * 2
        if len(delta) == 2:
            delta = (self.value,) + tuple(delta)
        return delta

    @staticmethod
    def _placeholders(expression):
        import re
        return re.findall(r'\{(.*?)\}', expression)

    @property
    def depends(self):
        """Names of the parameters this one is computed from (``derived='{a} + {b}'``: parameter.py:758-776), in order of appearance; empty otherwise."""
        if isinstance(self._derived, str) and self._derived not in ALLOWED_SOLVED:
            names = []
            for name in self._placeholders(self._derived):
                if name not in names: names.append(name)
            return names
        return []

    def eval(self, **values):
        """Value given the values of all parameters (parameter.py:795-807): the expression with each ``{name}`` replaced by ``values[name]`` (scalars, numpy arrays or
        torch tensors: arithmetic operators work on all of them; ``np`` / ``jnp`` name numpy for host arrays), or ``values[self.name]`` for an ordinary parameter."""
        depends = self.depends
        if not depends:
            return values[self.name]
        missing = [name for name in depends if name not in values]
        if missing:
            raise ParameterError('Parameter {} is to be derived from parameters {}, as {}, but {} are not provided'.format(self.name, depends, self._derived, missing))
        expression, local = self._derived, {}
        for i, name in enumerate(depends):
            key = '_dl_arg{:d}_'.format(i)
            expression = expression.replace('{' + name + '}', key)
            local[key] = values[name]
        xp = np
        if any(type(value).__module__.startswith('torch') for value in local.values()):
            import torch as xp      # device tensors: sqrt / exp / log ... have the same names
        return eval(expression, {'__builtins__': {}, 'np': xp, 'jnp': xp, 'abs': abs, 'min': min, 'max': max}, local)

    @property
    def solved(self):
        return (not self._fixed) and self._derived in ALLOWED_SOLVED

    @property
    def input(self):
        return self._derived is False or isinstance(self._derived, str)

    def latex(self, **kwargs):
        return self._latex if self._latex is not None else self.name

    def __getstate__(self):
        state = {key: getattr(self, '_' + key) for key in self._attrs}
        state['prior'], state['ref'] = self._prior.__getstate__(), self._ref.__getstate__()
        return state

    def __setstate__(self, state):
        self.__init__(**state)

    def update(self, *args, **kwargs):
        state = self.__getstate__()
        if len(args) == 1 and isinstance(args[0], Parameter):
            state.update(args[0].__getstate__())
        elif args:
            raise ValueError('Unrecognized arguments {}'.format(args))
        if 'name' in kwargs:
            kwargs['basename'], kwargs['namespace'] = kwargs.pop('name'), ''
        if 'prior' in kwargs and 'ref' not in kwargs and self._ref == self._prior:
            state.pop('ref')  # ref defaulted to the prior: keep following it
        state.update(kwargs)
        self.__init__(**state)

    def clone(self, *args, **kwargs):
        new = self.copy()
        new.update(*args, **kwargs)
        return new

    def copy(self):
        return Parameter(self)

    def __repr__(self):
        return 'Parameter({}, {})'.format(self.name, 'fixed' if self._fixed else 'varied')

    def __str__(self):
        return self.name

    def __eq__(self, other):
        return type(other) == type(self) and self.__getstate__() == other.__getstate__()

    def __hash__(self):
        return hash(self.name)


_generation = [0]   # bumped by every Parameter construction / update and every change of a collection: compiled contexts check it before trusting their cache


def generation():
    return _generation[0]


class ParameterCollection(object):
    """Ordered name -> :class:`Parameter` collection (parameter.py:1657-1897)."""

    def __init__(self, data=None):
        self.data = []
        if data is None:
            return
        if isinstance(data, ParameterCollection):
            self.data = [param.copy() for param in data.data]
            return
        if isinstance(data, dict):
            for name, conf in data.items():
                if isinstance(conf, Parameter):
                    self.set(conf)
                else:
                    self.set(Parameter(basename=name, **(conf or {})))
            return
        for item in data:
            self._set(item if isinstance(item, Parameter) else Parameter(item))

    def _index(self, name):
        name = str(name)
        for i, param in enumerate(self.data):
            if param.name == name:
                return i
        return None

    def _set(self, param):
        i = self._index(param.name)
        if i is None: self.data.append(param)
        else: self.data[i] = param

    def set(self, param):
        """Add / replace a parameter.  Bumps the global generation: compiled contexts re-check their parameters.  (Building a collection from existing parameters
        -- the ``varied_params`` / ``solved_params`` views made at every call -- goes through :meth:`_set` and does not.)"""
        _generation[0] += 1
        self._set(param)

    def get(self, name, *default):
        i = self._index(name)
        if i is None:
            if default: return default[0]
            raise KeyError('Parameter {} not found'.format(name))
        return self.data[i]

    def pop(self, name, *default):
        i = self._index(name)
        if i is None:
            if default: return default[0]
            raise KeyError('Parameter {} not found'.format(name))
        _generation[0] += 1
        return self.data.pop(i)

    def __getitem__(self, name):
        if isinstance(name, (int, np.integer)):
            return self.data[name]
        return self.get(name)

    def __setitem__(self, name, item):
        if isinstance(item, dict):
            item = Parameter(basename=name, **item)
        if str(name) != item.name:
            raise KeyError('Parameter {} must be indexed by name (incorrect {})'.format(item.name, name))
        self.set(item)

    def __delitem__(self, name):
        self.pop(str(name))

    def __contains__(self, name):
        return self._index(str(name)) is not None

    def __iter__(self):
        return iter(self.data)

    def __len__(self):
        return len(self.data)

    def names(self, **kwargs):
        return [param.name for param in (self.select(**kwargs) if kwargs else self)]

    def basenames(self, **kwargs):
        return [param.basename for param in (self.select(**kwargs) if kwargs else self)]

    def select(self, **kwargs):
        """Select by attribute, e.g. ``select(varied=True, solved=False)``, ``select(basename=['b1', 'sn0'])`` (fnmatch patterns allowed)."""
        import fnmatch
        toret = ParameterCollection()
        for param in self.data:
            keep = True
            for key, value in kwargs.items():
                if key in ('name', 'basename'):
                    patterns = [value] if isinstance(value, str) else list(value)
                    keep &= any(fnmatch.fnmatchcase(getattr(param, key), str(pattern)) for pattern in patterns)
                else:
                    keep &= getattr(param, key) == value
            if keep:
                toret.data.append(param)
        return toret

    def update(self, other=None, **kwargs):
        if other is not None:
            for param in (other if not isinstance(other, dict) else ParameterCollection(other)):
                self.set(param)
        for name, conf in kwargs.items():
            self[name].update(**conf)

    def update_config(self, config, accept_new=None):
        """Apply a parameter configuration to the parameters of this collection IN PLACE, with the semantics of ``calculator.all_params = config`` /
        ``pipeline.params = config`` (base.py:1307-1310, 436-470; parameter.py:1472-1547, 1588-1620):

        - ``config``: dict ``name -> settings`` (``dict(value=..., fixed=..., prior=..., ref=..., delta=..., derived=..., latex=...)``, a number = the value, or a
          :class:`Parameter`), a :class:`ParameterCollection`, or the name of a YAML file holding such a dict;
        - names with ``*`` are patterns applied to every matching parameter (``{'*': {'prior': ...}}``, ``{'*mega_m': ...}``);
        - meta entries ``.fixed`` / ``.varied`` / ``.derived`` (name, pattern, list of them, or ``dict name -> bool``) and ``.delete``;
        - order: deletions, meta entries, patterns, then the exact names (an explicit entry wins over a pattern of the same configuration);
        - an exact name that is not in the collection is a NEW parameter if other parameters are derived from it (``derived='{b}**2'``; or if ``accept_new(name)``
          says so), otherwise an error ("Cannot attribute parameter ... to any calculator", base.py:463-464).
        Returns the names of the parameters that were added."""
        import fnmatch
        import numbers
        if isinstance(config, str):
            import yaml
            with open(config, 'r') as file:
                config = yaml.safe_load(file) or {}
        if isinstance(config, ParameterCollection):
            config = {param.name: param for param in config}
        config = dict(config)

        def matching(pattern):
            return [param for param in self.data if fnmatch.fnmatchcase(param.name, str(pattern))]

        def meta(key):
            value = config.pop(key, {})
            if isinstance(value, dict): return dict(value)
            if isinstance(value, (list, tuple)): return {name: True for name in value}
            return {value: True}

        fixed = {name: bool(value) for name, value in meta('.fixed').items()}
        fixed.update({name: not bool(value) for name, value in meta('.varied').items()})
        derived, delete = meta('.derived'), meta('.delete')
        if config.pop('.namespace', None):
            raise ParameterError('.namespace entries are not supported: name the parameters in full (namespace.basename)')
        for name, flag in delete.items():
            if flag:
                for param in matching(name): self.pop(param.name)
        for name, flag in fixed.items():
            for param in matching(name): param.update(fixed=flag)
        for name, flag in derived.items():
            for param in matching(name): param.update(derived=bool(flag))

        def settings(conf):
            if isinstance(conf, Parameter):
                state = conf.__getstate__()
                return {key: value for key, value in state.items() if key not in ('basename', 'namespace')}
            if isinstance(conf, numbers.Number): return {'value': conf}
            conf = dict(conf or {})
            for key in ('name', 'basename', 'namespace'): conf.pop(key, None)
            return conf

        # names other parameters are (or, with this configuration, will be) derived from
        referenced = set(name for param in self.data for name in param.depends)
        for conf in config.values():
            expression = conf.derived if isinstance(conf, Parameter) else (conf.get('derived', None) if isinstance(conf, dict) else None)
            if isinstance(expression, str): referenced.update(Parameter._placeholders(expression))
        if accept_new is None:
            accept_new = referenced.__contains__
        exact = {}
        for name, conf in config.items():
            if '*' in str(name):
                for param in matching(name): param.update(**settings(conf))
            else:
                exact[str(name)] = conf
        added = []
        for name, conf in exact.items():
            if name in self:
                self[name].update(**settings(conf))
            elif accept_new(name):
                self.set(conf.copy() if isinstance(conf, Parameter) else Parameter(basename=name, **settings(conf)))
                added.append(name)
            else:
                raise ParameterError('Cannot attribute parameter {} to any calculator: parameters are {}'.format(name, self.names()))
        return added

    def clear(self):
        self.data = []

    def copy(self):
        return ParameterCollection(self)

    def deepcopy(self):
        return ParameterCollection(self)

    def __add__(self, other):
        new = self.copy()
        new.update(ParameterCollection(other))
        return new

    def __radd__(self, other):
        if other in (0, None): return self.copy()
        return ParameterCollection(other) + self

    def eval(self, **params):
        """Values of all parameters that can be computed from ``params``, e.g. ``{'a': 2., 'b': 3., 'c': 5.}`` if ``c.derived`` is '{a} + {b}' (parameter.py:1872-1887)."""
        toret = {}
        pending = list(self.data)
        values = dict(params)
        while pending:   # parameters derived from derived ones: as many passes as levels
            left = []
            for param in pending:
                try:
                    values[param.name] = toret[param.name] = param.eval(**values)
                except (ParameterError, KeyError):
                    left.append(param)
            if len(left) == len(pending): break
            pending = left
        return {param.name: toret[param.name] for param in self.data if param.name in toret}

    def prior(self, **params):
        """Total log-prior of varied, non-solved parameters, including those derived from others by an expression (parameter.py:1889-1897)."""
        values = self.eval(**params)
        toret = 0.
        for param in self.data:
            if param.varied and not param.solved and (param.depends or param.derived is False) and param.name in values:
                toret = toret + param.prior(values[param.name])
        return toret

    def __repr__(self):
        return 'ParameterCollection({})'.format(self.names())


class Samples(dict):
    """Minimal dict-of-arrays container standing in for ``desilike.parameter.Samples`` (parameter.py:2127): name -> array[B]."""

    def __init__(self, data=None, params=None):
        super(Samples, self).__init__()
        if data is not None and params is not None:
            for param, column in zip(params, data):
                self[str(param)] = np.asarray(column)
        elif data is not None:
            self.update(data)

    def __getitem__(self, name):
        return super(Samples, self).__getitem__(str(name))

    def __setitem__(self, name, value):
        super(Samples, self).__setitem__(str(name), value)

    def __contains__(self, name):
        return super(Samples, self).__contains__(str(name))

    def to_dict(self, params=None):
        if params is None:
            return dict(self)
        return {str(param): self[param] for param in params}

    @property
    def shape(self):
        for value in self.values():
            return np.shape(value)
        return ()
