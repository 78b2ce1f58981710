"""Stand-in for the un-vendored ``cosmoprimo.emulators.tools`` (TEST INFRASTRUCTURE ONLY; never shipped to the GPU box).

The reference's ``desilike.emulators`` subclasses these third-party classes (``/root/reference/desilike/emulators/__init__.py:55, 430, 510``); its OWN code fixes the
attribute surface that is restated here, nothing more:

* Taylor engine: ``center``, ``powers``, ``derivatives`` (written by the reference's ``_fit_no_operation``, emulators/__init__.py:471-507), ``params``, ``xshape``, ``yshape``;
* MLP engine: ``model_operations`` = list of :class:`Operation` whose ``_locals`` hold ``kernel`` / ``bias`` and whose expressions are the reference's own strings
  (emulators/conversion.py:20-35), ``xoperations`` / ``yoperations`` = min-max scalers with ``_locals['limits']`` (conversion.py:75-79; ``operation._locals`` /
  ``operation.update(locals=...)``: full_shape.py:1440-1441);
* emulator: ``engines``, ``xoperations``, ``yoperations``, ``defaults``, ``fixed``, ``varied_params``, ``in_calculator_state``, ``predict(params) -> state``,
  ``deepcopy`` (emulators/__init__.py:150-208, 386-409; full_shape.py:1416-1443);
* ``Emulator.from_state(state)`` on the state dictionary emulators/conversion.py:44-98 assembles (the jaxeffort layout): ``engines[name]`` = dict(name='mlp', params, xshape,
  yshape, xoperations, yoperations, model_operations, model_yoperations) with every operation in its ``__getstate__`` form; operations whose expressions use the inputs
  by name (``X['logA']``, conversion.py:88-92): ``X`` is the dictionary of the engine's input parameters; kernels stacked over leading axes (conversion.py:58-66) go
  through the reference's own layer expression by broadcasting.

With it ``Emulator.to_calculator()`` -- the reference's own code -- builds a real ``EmulatedCalculator`` that the reference's velocileptors tracer classes accept as ``pt``.
The forward pass below is plain NumPy evaluation of the operations' own expression strings; the engine is third-party: parity of the engine itself stays unpinned.
"""
import copy

import numpy as np


class Operation(object):
    """``direct`` / ``inverse``: Python expressions of ``v`` (the array or state dictionary), ``X`` (inputs) and the ``locals``; ``jnp`` is NumPy here."""

    def __init__(self, direct='v', inverse=None, locals=None):
        self._direct, self._inverse, self._locals = direct, inverse, dict(locals or {})

    def _eval(self, expression, v, X=None):
        if not expression: return v
        scope = {'np': np, 'jnp': np, 'v': v, 'X': X}
        scope.update(self._locals)
        toret = None
        for piece in expression.split(';'):     # "statement; statement; v" (conversion.py:50-51)
            piece = piece.strip()
            if not piece: continue
            try:
                toret = eval(piece, scope)
            except SyntaxError:
                exec(piece, scope); toret = scope['v']
        return toret

    def __call__(self, v, X=None):
        return self._eval(self._direct, v, X=X)

    def inverse(self, v, X=None):
        return self._eval(self._inverse, v, X=X)

    def update(self, **kwargs):
        for name, value in kwargs.items(): setattr(self, '_' + name, value)

    def clone(self, **kwargs):
        new = self.copy()
        new.update(**kwargs)
        return new

    def copy(self):
        return copy.deepcopy(self)

    def __getstate__(self):
        return {'direct': self._direct, 'inverse': self._inverse, 'locals': self._locals}

    def __setstate__(self, state):
        self._direct, self._inverse, self._locals = state['direct'], state['inverse'], dict(state['locals'])

    @classmethod
    def from_state(cls, state):
        new = cls.__new__(cls)
        new.__setstate__(state)
        return new


class PCAOperation(Operation):
    pass


class BaseEmulatorEngine(object):
    name = 'base'

    def __init__(self, params=None, xshape=None, yshape=None, xoperations=None, yoperations=None):
        self.params, self.xshape, self.yshape = params, xshape, yshape
        self.xoperations, self.yoperations = list(xoperations or []), list(yoperations or [])

    def predict(self, X):
        """``X``: the inputs as an array in the order of ``params``, or a dictionary name -> value (operations may then address them by name, conversion.py:88-92)."""
        if isinstance(X, dict):
            X = {name: X[name] for name in self.params}
            v = np.array([X[name] for name in self.params], dtype='f8')
        else:
            v = np.asarray(X, dtype='f8')
        for operation in self.xoperations: v = operation(v, X=X)
        v = self._predict_no_operation(v)
        for operation in self.yoperations[::-1]: v = operation.inverse(v, X=X)
        return v

    @classmethod
    def from_state(cls, state):
        """The engine dictionary of emulators/conversion.py:94-96."""
        new = cls.__new__(cls)
        state = dict(state)
        state.pop('name', None)
        for name in ['xoperations', 'yoperations', 'model_operations', 'model_yoperations']:
            if name in state: state[name] = [operation if isinstance(operation, Operation) else Operation.from_state(operation) for operation in state[name]]
        for name in ['xshape', 'yshape']:
            if name in state: state[name] = tuple(state[name])
        new.__dict__.update(state)
        return new


class PointEmulatorEngine(BaseEmulatorEngine):
    name = 'point'


class TaylorEmulatorEngine(BaseEmulatorEngine):
    """``y = sum_t derivatives[t] prod_p (x_p - center_p)^powers[t, p]``, derivatives already divided by the factorials (emulators/__init__.py:471-507)."""
    name = 'taylor'

    def _predict_no_operation(self, x):
        monomials = np.prod((x - self.center)**self.powers, axis=-1)
        return np.tensordot(monomials, self.derivatives, axes=(0, 0))


class MLPEmulatorEngine(BaseEmulatorEngine):
    name = 'mlp'

    def _predict_no_operation(self, x):
        v = x
        for operation in self.model_operations: v = operation(v)
        return np.asarray(v).reshape(self.yshape)


def get_engine(name):
    return {'point': PointEmulatorEngine, 'taylor': TaylorEmulatorEngine, 'mlp': MLPEmulatorEngine}[name]


class Emulator(object):

    def predict(self, params):
        X = np.array([params[name] for name in self.varied_params], dtype='f8')
        # (an engine built by ``from_state`` names its inputs: it takes the dictionary; the fitted engines of the other fixtures keep the array path -- same arithmetic)
        state = {name: engine.predict({n: params[n] for n in engine.params} if getattr(engine, '_by_name', False) else X) for name, engine in self.engines.items()}
        state.update(self.fixed)
        for operation in self.yoperations[::-1]: state = operation.inverse(state, X=params)
        return state

    def deepcopy(self):
        return copy.deepcopy(self)

    def __getstate__(self):
        return dict(self.__dict__)

    def __setstate__(self, state):
        state = dict(state)
        engines = {}
        for name, engine in state.get('engines', {}).items():
            if isinstance(engine, dict):
                engine = get_engine(engine['name']).from_state(engine)
                engine._by_name = True
            engines[name] = engine
        state['engines'] = engines
        for name in ['xoperations', 'yoperations']:
            state[name] = [operation if isinstance(operation, Operation) else Operation.from_state(operation) for operation in state.get(name, [])]
        self.__dict__.update(state)

    @classmethod
    def from_state(cls, state):
        new = cls.__new__(cls)
        new.__setstate__(state)
        return new
