"""Emulated calculators: forward pass of Taylor / MLP emulators on the GPU.

In the reference, ``desilike.emulators.Emulator`` wraps the third-party ``cosmoprimo.emulators.tools`` engines (un-vendored) and
``EmulatedCalculator`` (emulators/__init__.py:394-418) replaces any calculator by ``emulator.predict(params)`` -> state arrays.  Training /
sampling of the emulator is out of scope; this module holds the *fitted* engines in the layouts the reference writes:

* :class:`TaylorEmulatorEngine` -- ``center [P]``, ``powers [n_terms, P]``, ``derivatives [n_terms, *yshape]`` already divided by the factorials
  (emulators/__init__.py:471-507): ``y = sum_t derivatives[t] prod_p (x_p - c_p)^powers[t, p]``;
* :class:`MLPEmulatorEngine` -- min-max x-scaler, dense layers ``v @ kernel + bias`` with silu / relu / tanh between them, inverse min-max
  y-scaler (emulators/conversion.py:20-35, 63-96).

MI355X design: an emulated array never materialises.  The final linear layer of the engine (Taylor: the derivative table), the sum over the
velocileptors bias monomials, the k-interpolation, the window matrix and the Cholesky factor of the precision are constant and linear, so
they are multiplied together ONCE on the host (:meth:`EmulatedCalculator.fold`); per point the GPU evaluates only the small trunk
(hidden layers / Taylor monomials), forms the features ``phi[(h, m)] = basis_h * mono_m`` and the fp64 MFMA GEMM applies the folded matrix.
"""
import numpy as np

ACTIVATIONS = {'silu': 0, 'relu': 1, 'tanh': 2}


class TaylorEmulatorEngine(object):

    def __init__(self, center, powers, derivatives):
        self.center = np.asarray(center, dtype='f8')
        self.powers = np.asarray(powers, dtype='i4').reshape(-1, self.center.size)
        self.derivatives = np.asarray(derivatives, dtype='f8')
        if self.derivatives.shape[0] != self.powers.shape[0]:
            raise ValueError('derivatives and powers must have the same number of terms')
        self.yshape = self.derivatives.shape[1:]

    n_basis = property(lambda self: self.powers.shape[0])

    def basis_matrix(self):
        """[n_basis, prod(yshape)]: output = basis . matrix."""
        return self.derivatives.reshape(self.n_basis, -1)

    def spec(self, scalar=False):
        spec = dict(type=np.array([1], dtype='i4'), center=self.center, powers=self.powers)
        if scalar:
            spec['coef'] = self.derivatives.reshape(self.n_basis)
        return spec


class MLPEmulatorEngine(object):

    def __init__(self, xlimits, layers, activation='silu', ylimits=None, yshape=None):
        self.xlimits = np.asarray(xlimits, dtype='f8').reshape(-1, 2)
        self.layers = [(np.asarray(kernel, dtype='f8'), np.asarray(bias, dtype='f8')) for kernel, bias in layers]
        if activation not in ACTIVATIONS:
            raise ValueError('activation must be one of {}'.format(list(ACTIVATIONS)))
        self.activation = activation
        nout = self.layers[-1][0].shape[1]
        self.yshape = tuple(yshape) if yshape is not None else (nout,)
        if int(np.prod(self.yshape, dtype='i8')) != nout:
            raise ValueError('yshape does not match the output layer')
        ylimits = np.array([[0., 1.]]) if ylimits is None else np.asarray(ylimits, dtype='f8')
        self.ylimits = np.broadcast_to(ylimits.reshape(-1, 2), (nout, 2))

    n_basis = property(lambda self: self.layers[-1][0].shape[0] + 1)

    def basis_matrix(self):
        """[hidden + 1, n_out]: final layer with the y-scaler folded in; the last row multiplies the constant basis function 1."""
        kernel, bias = self.layers[-1]
        scale, lo = self.ylimits[:, 1] - self.ylimits[:, 0], self.ylimits[:, 0]
        return np.vstack([kernel * scale, bias * scale + lo])

    def spec(self, scalar=False):
        layers = self.layers if scalar else self.layers[:-1]
        if not layers:
            raise ValueError('the MLP needs at least one hidden layer')
        widths = [self.xlimits.shape[0]] + [kernel.shape[1] for kernel, bias in layers]
        weights = np.concatenate([np.concatenate([kernel.ravel(), bias.ravel()]) for kernel, bias in layers])
        spec = dict(type=np.array([0], dtype='i4'), xlimits=self.xlimits, widths=np.array(widths, dtype='i4'), act=np.array([ACTIVATIONS[self.activation]], dtype='i4'), weights=weights)
        if scalar:
            spec['ylimits'] = self.ylimits[0]
        return spec


class EmulatedCalculator(object):
    """Stand-in for a calculator whose state arrays are emulated (emulators/__init__.py:394-418).

    Parameters
    ----------
    params : list of str
        Names of the parameters the emulator depends on (its inputs, in order); they must be parameters of the likelihood.
    engines : dict
        name -> engine for each emulated array: 'pktable' [n_ell, n_k, 19] (velocileptors tables) or 'power' [n_ell, n_k], and optionally
        the scalars 'sigma8', 'fsigma8' (floats are accepted for constants).
    k, ells, z : fixed attributes of the emulated calculator.
    param_specs : dict, optional
        name -> Parameter keyword arguments (prior, ref, value...) for the emulator's own parameters.
    """

    def __init__(self, params, engines, k, ells=(0, 2, 4), z=1., param_specs=None):
        self.param_names = list(params)
        self.engines = dict(engines)
        self.k = np.asarray(k, dtype='f8')
        self.ells = tuple(ells)
        self.z = z
        self.param_specs = dict(param_specs or {})
        self.table_name = 'pktable' if 'pktable' in self.engines else 'power'
        yshape = self.engines[self.table_name].yshape
        expected = (len(self.ells), self.k.size) + ((19,) if self.table_name == 'pktable' else ())
        if tuple(yshape) != expected:
            raise ValueError('{} engine has output shape {}, expected {}'.format(self.table_name, yshape, expected))

    def engine_specs(self):
        specs = {'emu0': self.engines[self.table_name].spec(scalar=False)}
        for ie, name in [(1, 'sigma8'), (2, 'fsigma8')]:
            engine = self.engines.get(name, None)
            if engine is None: continue
            if np.ndim(engine) == 0 and not hasattr(engine, 'spec'):
                specs['emu{:d}'.format(ie)] = {'const': [float(engine)]}
            else:
                specs['emu{:d}'.format(ie)] = engine.spec(scalar=True)
        return specs
