"""Calculator plugin surface kept from desilike (reference: desilike/base.py -- ``BaseCalculator`` 1119-1323,
``InitConfig`` 30-121, ``vmap`` 232-383), re-designed for a batched GPU back-end.

In the reference every calculator runs its own ``calculate()`` per point inside ``BasePipeline.calculate``
(base.py:510-572).  Here calculators are *declarative*: they keep the reference's constructor arguments,
parameter names / priors and attributes, and contribute constants to one likelihood ``spec`` that is
uploaded once (``dl_create``); evaluation of any batch is a single ``dl_eval_batch`` call.
"""
import numpy as np

from .parameter import Parameter, ParameterCollection, Samples


class PipelineError(Exception):
    """Error raised when a calculator cannot be set up (name kept from desilike/base.py)."""


class InitConfig(dict):
    """Constructor arguments + parameters of a calculator; updating it invalidates compiled state (base.py:30-121)."""

    def __init__(self, owner, kwargs, params):
        super(InitConfig, self).__init__(kwargs)
        self._owner = owner
        self._params = params

    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, params):
        self._params = params if isinstance(params, ParameterCollection) else ParameterCollection(params)
        self._owner._invalidate()

    @staticmethod
    def _same(a, b):
        if a is b: return True
        try:
            return bool(np.array_equal(np.asarray(a, dtype='f8'), np.asarray(b, dtype='f8')))
        except (TypeError, ValueError):
            return False

    def update(self, *args, **kwargs):
        new = dict(*args, **kwargs)
        changed = any(key not in self or not self._same(self[key], value) for key, value in new.items())
        super(InitConfig, self).update(new)
        if changed: self._owner._invalidate()

    def __setitem__(self, key, value):
        changed = key not in self or not self._same(self[key], value)
        super(InitConfig, self).__setitem__(key, value)
        if changed: self._owner._invalidate()

    def setdefault(self, key, value):
        if key not in self:
            self[key] = value
        return self[key]


class Monitor(object):
    """Accumulated execution time and number of data points (the subset of desilike/utils.py:734-800 the pipeline uses: ``start`` / ``stop`` / ``add`` / ``reset`` /
    ``counter`` / ``get('time', average=...)``).  Here the data points are kernel intervals read off the device (dispatch-attached events), not host clocks."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._counter, self._time, self._start = 0, 0., None

    def start(self):
        import time
        self._start = time.time()

    def stop(self):
        import time
        self.add(time.time() - self._start)

    def add(self, seconds, count=1):
        self._counter += int(count)
        self._time += float(seconds)

    @property
    def counter(self):
        return self._counter

    def get(self, quantity='time', average=True):
        if quantity != 'time': raise ValueError('only time is monitored')
        if average: return self._time / self._counter if self._counter else 0.
        return self._time


class RuntimeInfo(object):
    """What the reference keeps per calculator in ``calculator.runtime_info`` for scheduling decisions (base.py:1070-1075, 695-735): ``monitor`` (time spent in the
    calculator's part of the evaluations) and ``speed`` (evaluations per second of that part; ``None`` until :meth:`BaseLikelihood._set_speed` measured it)."""

    def __init__(self, calculator):
        self.calculator, self.monitor, self.speed = calculator, Monitor(), None


class BaseCalculator(object):
    """Base calculator: ``init`` (arguments + params), ``params``, lazy ``initialize`` (base.py:1119-1323)."""

    _params = {}

    def __init__(self, *args, **kwargs):
        if args:
            raise TypeError('{} takes keyword arguments only'.format(self.__class__.__name__))
        self._dependents = []
        self._initialized = False
        self.runtime_info = RuntimeInfo(self)
        self.init = InitConfig(self, kwargs, ParameterCollection(self._default_params(**kwargs)))

    @classmethod
    def _default_params(cls, **kwargs):
        import copy
        return copy.deepcopy(cls._params)

    @property
    def params(self):
        return self.init.params

    @params.setter
    def params(self, params):
        self.init.params = params

    def _all_params(self):
        """Parameters of this calculator and of the calculators it requires (overridden by the theories: template + own)."""
        return self.init.params

    @property
    def all_params(self):
        """All parameters of the pipeline below this calculator (base.py:1302-1305): a live view, ``calculator.all_params['b1'].update(...)`` acts on the
        calculator's own parameter."""
        self.initialize()
        view = ParameterCollection()
        for collection in self._param_collections():
            for param in collection:
                if param.name not in view: view.data.append(param)
        return view

    def _param_collections(self):
        """The :class:`ParameterCollection` objects ``all_params`` is made of, own parameters last."""
        return [self.init.params]

    @all_params.setter
    def all_params(self, config):
        """``calculator.all_params = {...} | ParameterCollection | 'params.yaml'``: update the pipeline's parameters by name, patterns and meta entries
        (:meth:`ParameterCollection.update_config`; base.py:1307-1310)."""
        view = self.all_params
        before = view.names()
        added = view.update_config(config)
        for name in added: self.init.params.set(view[name])
        for name in before:
            if name not in view:
                for collection in self._param_collections():
                    if name in collection: collection.pop(name)
        self._invalidate()

    # ---- standalone evaluation: calculator(**params) (base.py:1194-1196) -------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        """``theory(b1=2., qpar=1.01).power``, ``observable(b1=2.).flattheory``: evaluate the pipeline below this calculator at these parameters (the others at
        their default values) and return the calculator, its products set as attributes (the reference: ``BaseCalculator.__call__`` -> ``get()`` = ``self``).
        The evaluation is the device one: a likelihood with unit precision is compiled around the calculator (kept until its parameters or arguments change), the
        theory vector comes from ``dl_eval_theory`` / the ``flattheory`` output of ``dl_eval_batch``."""
        params = {}
        for arg in args: params.update(arg)
        params.update(kwargs)
        self.initialize()
        signature = tuple((param.name, repr(param.__getstate__())) for param in self.all_params)
        cache = getattr(self, '_standalone', None)
        if cache is None or cache[0] != signature:
            cache = (signature,) + self._standalone_pipeline()
            self._standalone = cache
        likelihood = cache[1]
        likelihood(**params)
        self._standalone_products(likelihood)
        return self

    def _standalone_pipeline(self):
        """(likelihood, [hidden calculators]) evaluating this calculator; theories and observables override."""
        raise NotImplementedError('{} cannot be evaluated on its own: call the observable or the likelihood it belongs to'.format(self.__class__.__name__))

    def _standalone_products(self, likelihood):
        pass

    def _standalone_theory_pipeline(self):
        """For tracer theories: a hidden observable that takes the theory at its own ``k`` (or ``s``) and ``ells`` with no window, zero data, unit precision."""
        from .likelihoods import ObservablesGaussianLikelihood
        from .observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable, TracerCorrelationFunctionMultipolesObservable
        self.initialize()
        ells = tuple(self.ells)
        if hasattr(self, 's') and getattr(self, '_standalone_space', 'pk') == 'xi':
            x = np.array(self.s, dtype='f8')
            observable = TracerCorrelationFunctionMultipolesObservable(data=np.zeros(len(ells) * x.size), s=x, ells=ells, theory=self)
        else:
            x = np.array(self.k, dtype='f8')
            observable = TracerPowerSpectrumMultipolesObservable(data=np.zeros(len(ells) * x.size), k=x, ells=ells, theory=self)
        return ObservablesGaussianLikelihood(observables=[observable], precision=np.ones(len(ells) * x.size)), [observable]

    def _invalidate(self):
        self._initialized = False
        self._standalone = None
        for dep in getattr(self, '_dependents', []):
            dep._invalidate()

    def _require(self, calculator):
        """Register ``calculator`` as a requirement: its updates invalidate ``self`` (base.py:1024-1029)."""
        if self not in calculator._dependents:
            calculator._dependents.append(self)
        return calculator

    def _param_signature(self):
        return tuple(sorted((name, repr(param.__getstate__())) for name, param in ((p.name, p) for p in self.init.params)))


def _check_params(params):
    """dict name -> array: broadcast scalars, return (dict of 1-D arrays, batch shape) (base.py:124-159)."""
    shapes = [np.shape(value) for value in params.values()]
    shape = ()
    for s in shapes:
        if s != () and shape not in ((), s):
            raise ValueError('input shapes are different: {}'.format(dict(zip(params, shapes))))
        if s != ():
            shape = s
    size = int(np.prod(shape, dtype='i8')) if shape else 1
    flat = {name: np.broadcast_to(np.asarray(value, dtype='f8'), shape).reshape(size) if shape else np.full(1, value, dtype='f8') for name, value in params.items()}
    return flat, shape


def vmap(calculate, backend=None, errors='raise', mpicomm=None, mpi_max_chunk_size=100, **kwargs):
    """Vectorise ``calculate`` over a dict of parameter arrays (reference: base.py:232-383).

    For the likelihoods of this package the whole batch is ONE GPU evaluation, whatever ``backend`` says
    ('jax' / 'mpi' / None are accepted for drop-in compatibility); ``errors`` keeps its meaning:
    per-point failures (non-finite results) are reported, never raised, when errors != 'raise'.
    Any other callable falls back to the reference's plain Python loop (backend=None semantics).
    """
    errors = str(errors)
    batched = getattr(calculate, '_evaluate_dict', None)

    if batched is not None:
        def wrapper(params, **kw):
            kw = {**kwargs, **kw}
            kw.pop('mpicomm', None)
            flat, shape = _check_params(params)
            results, errs = batched(flat, shape, errors=errors, **kw)
            if errors == 'return':
                return results, errs
            return results
        wrapper.__wrapped__vmap__ = calculate
        return wrapper

    def wrapper(params, **kw):
        kw = {**kwargs, **kw}
        kw.pop('mpicomm', None)
        flat, shape = _check_params(params)
        size = len(next(iter(flat.values()))) if flat else 0
        results, errs = [], {}
        for i in range(size):
            try:
                results.append(calculate({name: value[i] for name, value in flat.items()}, **kw))
            except Exception as exc:
                if errors == 'raise':
                    raise
                import traceback
                errs[i] = (exc, traceback.format_exc())
                results.append(None)
        ref = next((res for res in results if res is not None), None)
        if ref is not None:
            fill = np.nan * np.asarray(ref) if errors == 'nan' else ref
            results = [fill if res is None else res for res in results]
            results = np.asarray(results).reshape(shape + np.shape(ref))
        if errors == 'return':
            return results, errs
        return results

    return wrapper
