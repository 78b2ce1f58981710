"""Emulated calculators: forward pass of Taylor / MLP emulators on the GPU.

In the reference, ``desilike.emulators.Emulator`` wraps the third-party ``cosmoprimo.emulators.tools`` engines (un-vendored) and
``EmulatedCalculator`` (emulators/__init__.py:394-418) replaces any calculator by ``emulator.predict(params)`` -> state arrays.  This module holds the
*fitted* engines in the layouts the reference writes, and fits them on the GPU: the Taylor engine (:func:`fit_taylor`: the whole finite-difference stencil is one GPU
batch) and the MLP engine (:func:`fit_mlp`: fp64 Adam with hand-written forward / backward kernels, ``dl_mlp_*``):

* :class:`TaylorEmulatorEngine` -- ``center [P]``, ``powers [n_terms, P]``, ``derivatives [n_terms, *yshape]`` already divided by the factorials
  (emulators/__init__.py:471-507): ``y = sum_t derivatives[t] prod_p (x_p - c_p)^powers[t, p]``;
* :class:`MLPEmulatorEngine` -- min-max x-scaler, dense layers ``v @ kernel + bias`` with silu / relu / tanh between them, inverse min-max
  y-scaler (emulators/conversion.py:20-35, 63-96).

MI355X design: an emulated array never materialises.  The final linear layer of the engine (Taylor: the derivative table), the sum over the
velocileptors bias monomials, the k-interpolation, the window matrix and the Cholesky factor of the precision are constant and linear, so
they are multiplied together ONCE on the host (:meth:`EmulatedCalculator.fold`); per point the GPU evaluates only the small trunk
(hidden layers / Taylor monomials), forms the features ``phi[(h, m)] = basis_h * mono_m`` and the fp64 MFMA GEMM applies the folded matrix.
"""
import re

import numpy as np

ACTIVATIONS = {'silu': 0, 'relu': 1, 'tanh': 2}
STACKED_COMPONENTS = (('11', 3), ('loop', 9), ('ct', 4), ('st', 3))     # engines of the jaxeffort layout and their bias monomials: jnp.split(pktable, [3, 12, 16], axis=2), emulators/conversion.py:50


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


def finite_difference_weights(nodes, order):
    """Weights w with sum_i w[i] f(nodes[i]) ~ f^(order)(0): Fornberg's recursion (Math. Comp. 51, 1988) for arbitrary nodes."""
    nodes = np.asarray(nodes, dtype='f8')
    n = nodes.size
    c = np.zeros((n, order + 1), dtype='f8')
    c[0, 0] = 1.
    c1 = 1.
    for i in range(1, n):
        c2 = 1.
        mn = min(i, order)
        for j in range(i):
            c3 = nodes[i] - nodes[j]
            c2 *= c3
            if j == i - 1:
                for k in range(mn, 0, -1):
                    c[i, k] = c1 * (k * c[i - 1, k - 1] - nodes[i - 1] * c[i - 1, k]) / c2
                c[i, 0] = -c1 * nodes[i - 1] * c[i - 1, 0] / c2
            for k in range(mn, 0, -1):
                c[j, k] = (nodes[i] * c[j, k] - k * c[j, k - 1]) / c3
            c[j, 0] = nodes[i] * c[j, 0] / c3
        c1 = c2
    return c[:, order]


def fit_taylor(function, center, delta, order=3, accuracy=2):
    r"""Fit a :class:`TaylorEmulatorEngine` of total order ``order`` around ``center`` by central finite differences (what the reference's
    ``TaylorEmulatorEngine.get_default_samples`` / ``_fit_no_operation`` produce through ``Differentiation``, emulators/__init__.py:430-507).

    ``function(points [N, P]) -> values [N, *yshape]`` is called ONCE with the whole tensor stencil (nodes ``center_p + i delta_p``, ``|i| <= (order + 1) // 2 - 1 +
    accuracy // 2``): with a GPU theory behind it (e.g. ``Context.eval_theory_host``) the fit is a single batch.  Derivatives
    :math:`\partial^\alpha f`, :math:`|\alpha| \le` ``order``, are tensor products of 1-D stencils; the engine stores them divided by :math:`\alpha!`.
    """
    import itertools
    from math import factorial
    center, delta = np.asarray(center, dtype='f8'), np.broadcast_to(np.asarray(delta, dtype='f8'), np.shape(center))
    ndim = center.size
    m = (order + 1) // 2 - 1 + max(accuracy // 2, 1)
    offsets = np.arange(-m, m + 1)
    grid = np.stack(np.meshgrid(*([offsets] * ndim), indexing='ij'), axis=-1).reshape(-1, ndim)
    values = np.asarray(function(center + grid * delta), dtype='f8')
    yshape = values.shape[1:]
    values = values.reshape((offsets.size,) * ndim + (-1,))
    weights = [finite_difference_weights(offsets.astype('f8'), d) for d in range(order + 1)]     # in units of delta^-d
    powers, derivatives = [], []
    for alpha in itertools.product(range(order + 1), repeat=ndim):
        if sum(alpha) > order: continue
        term = values
        for axis, d in enumerate(alpha):   # contract the leading axis each time: the remaining parameter axes move up
            term = np.tensordot(weights[d] / delta[axis]**d, term, axes=(0, 0))
        powers.append(alpha)
        derivatives.append(term.reshape(yshape) / np.prod([factorial(d) for d in alpha]))
    return TaylorEmulatorEngine(center, np.array(powers, dtype='i4'), np.array(derivatives))


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


class StackedMLPEmulatorEngine(object):
    """One engine of the layout the reference ships for its emulated perturbation-theory tables (emulators/conversion.py:44-98, the conversion of the jaxeffort
    emulators): ONE NETWORK PER (z, ell), all with the same hidden layers, stacked along leading axes (``merge_operations``, conversion.py:58-66).

    Parameters
    ----------
    xlimits : [P, 2] -- the engine's one min-max scaler of the inputs (conversion.py:72-75).
    layers : list of (kernel [n_z, n_ell, in, out], bias [n_z, n_ell, out]).
    activation : 'silu' | 'relu' | 'tanh' (conversion.py:27-34).
    ylimits : [n_z, n_ell, n_m, n_k, 2] -- inverse min-max scaler of the outputs (conversion.py:76-79).
    amplitude : (input name, factor, power) or None -- the outputs are multiplied by ``(exp(X[name]) * factor)**power`` (conversion.py:88-92: ``v * exp(logA) * 1e-10``
        for '11' and 'ct', its square for 'loop', nothing for 'st').
    """

    def __init__(self, xlimits, layers, activation, ylimits, amplitude=None):
        self.xlimits = np.asarray(xlimits, dtype='f8').reshape(-1, 2)
        self.layers = [(np.asarray(kernel, dtype='f8'), np.asarray(bias, dtype='f8')) for kernel, bias in layers]
        if activation not in ACTIVATIONS: raise ValueError('activation must be one of {}'.format(list(ACTIVATIONS)))
        self.activation = activation
        self.ylimits = np.asarray(ylimits, dtype='f8')
        if self.ylimits.ndim != 5 or self.ylimits.shape[-1] != 2: raise ValueError('ylimits must have shape [n_z, n_ell, n_m, n_k, 2]')
        self.yshape = self.ylimits.shape[:-1]
        stack = self.yshape[:2]
        for kernel, bias in self.layers:
            if kernel.shape[:2] != stack or bias.shape[:2] != stack or kernel.shape[-1] != bias.shape[-1]: raise ValueError('every layer must be stacked [n_z, n_ell, ...]')
        if len(self.layers) < 2: raise ValueError('the networks need at least one hidden layer')
        if self.layers[-1][0].shape[-1] != self.yshape[2] * self.yshape[3]: raise ValueError('the output layer does not match ylimits')
        if self.layers[0][0].shape[-2] != self.xlimits.shape[0]: raise ValueError('the input layer does not match xlimits')
        self.amplitude = None if amplitude is None or not amplitude[2] else (str(amplitude[0]), float(amplitude[1]), int(amplitude[2]))

    hidden = property(lambda self: [kernel.shape[-1] for kernel, bias in self.layers[:-1]])


def fit_mlp(x, y, hidden=(64, 64, 64), activation='silu', nsteps=3000, batch=None, lr=2e-3, lr_decay=0.2, seed=0, device=0, yshape=None, return_loss=False):
    """Train an :class:`MLPEmulatorEngine` on samples ``x [S, P]`` -> ``y [S, ...]`` on the GPU (what the reference's ``Emulator(..., engine=MLPEmulatorEngine(...)).fit()`` does
    through the third-party engine, emulators/__init__.py:510-533).  Min-max scalers of inputs and outputs as emulators/conversion.py:75-79 (constant outputs get a zero
    range: reproduced exactly); weights ~ N(0, 1 / fan_in) from ``RandomState(seed)``, zero biases; Adam on the mean squared error of the scaled outputs, ``nsteps`` steps on
    consecutive chunks of ``batch`` samples (default: all) of a fixed shuffle, learning rate decayed geometrically to ``lr * lr_decay``; fp64 throughout.
    ``x``, ``y``: numpy arrays or float64 tensors already on the GPU (``dl_eval_theory`` outputs never leave it)."""
    import torch
    from ._lib import MLPTrainer
    dev = torch.device('cuda', device)
    xt = torch.as_tensor(x, dtype=torch.float64, device=dev)
    yt = torch.as_tensor(y, dtype=torch.float64, device=dev)
    if yshape is None: yshape = tuple(yt.shape[1:])
    yt = yt.reshape(yt.shape[0], -1)
    nsamples, nin, nout = xt.shape[0], xt.shape[1], yt.shape[1]
    xlo, xhi = xt.min(dim=0).values, xt.max(dim=0).values
    ylo, yhi = yt.min(dim=0).values, yt.max(dim=0).values
    xscale = torch.where(xhi > xlo, xhi - xlo, torch.ones_like(xlo))
    yscale = torch.where(yhi > ylo, yhi - ylo, torch.ones_like(ylo))
    rng = np.random.RandomState(seed)
    order = torch.as_tensor(rng.permutation(nsamples), device=dev)
    xs = ((xt - xlo) / xscale)[order].contiguous()
    ys = ((yt - ylo) / yscale)[order].contiguous()
    layers, last = [], nin
    for width in list(hidden) + [nout]:
        layers.append((rng.standard_normal((last, width)) / last**0.5, np.zeros(width)))
        last = width
    trainer = MLPTrainer(layers, activation=activation, device=device)
    if batch is None: batch = nsamples
    losses, nstages = [], 8
    for stage in range(nstages):   # geometric learning-rate schedule in a few stages (each stage = one enqueued dl_mlp_train call)
        steps = nsteps // nstages + (nsteps % nstages if stage == nstages - 1 else 0)
        if steps: losses.append(trainer.train(xs, ys, batch, steps, lr=lr * lr_decay**(stage / max(nstages - 1, 1))))
    fitted = trainer.layers()
    trainer.close()
    xlimits = np.column_stack([xlo.cpu().numpy(), (xlo + xscale).cpu().numpy()])
    ylimits = np.column_stack([ylo.cpu().numpy(), torch.where(yhi > ylo, yhi, ylo).cpu().numpy()])
    engine = MLPEmulatorEngine(xlimits=xlimits, layers=fitted, activation=activation, ylimits=ylimits, yshape=yshape)
    return (engine, np.concatenate(losses)) if return_loss else engine


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
        self.stacked = all(name in self.engines for name, nm in STACKED_COMPONENTS)      # the layout of emulators/conversion.py:44-98
        if self.stacked:
            self.table_name = 'stacked'
            self.z = np.atleast_1d(np.asarray(z, dtype='f8'))
            for name, nm in STACKED_COMPONENTS:
                expected = (self.z.size, len(self.ells), nm, self.k.size)
                if tuple(self.engines[name].yshape) != expected: raise ValueError('engine {} has output shape {}, expected {}'.format(name, self.engines[name].yshape, expected))
            first = self.engines[STACKED_COMPONENTS[0][0]]
            for name, nm in STACKED_COMPONENTS:
                engine = self.engines[name]
                if engine.hidden != first.hidden or engine.activation != first.activation or not np.array_equal(engine.xlimits, first.xlimits):
                    raise ValueError('the engines of a stacked emulator share hidden layers, activation and input scaler')
            return
        self.table_name = 'pktable' if 'pktable' in self.engines else 'power'
        yshape = self.engines[self.table_name].yshape
        expected = (len(self.ells), self.k.size) + ((19,) if self.table_name == 'pktable' else ())
        if tuple(yshape) != expected:
            raise ValueError('{} engine has output shape {}, expected {}'.format(self.table_name, yshape, expected))

    @classmethod
    def from_state(cls, state, param_specs=None):
        """From the state dictionary ``convert_jaxeffort_to_desilike`` assembles and ``Emulator.save`` writes (emulators/conversion.py:44-98; ``np.load(fn, allow_pickle=True)[()]``):
        ``state['engines'][component]`` = dict(name='mlp', params, yshape, xoperations, yoperations, model_operations) with every operation as dict(direct, inverse, locals);
        ``state['fixed']``: ells, k, z.  The operations are recognised by their ``locals`` (``kernel`` / ``bias``: a dense layer; ``limits``: a min-max scaler) and, for the
        activations and the amplitude rescale, by the expression strings the converter writes (conversion.py:27-34, 88-92)."""
        expressions = {'v / (1 + jnp.exp(-v))': 'silu', 'jnp.maximum(v, 0.)': 'relu', 'jnp.tanh(v)': 'tanh'}
        engines, params = {}, None
        for name, engine in state['engines'].items():
            if engine.get('name', 'mlp') != 'mlp': raise NotImplementedError('engine {}: kind {}'.format(name, engine.get('name')))
            layers, activations = [], []
            for operation in engine['model_operations']:
                local = operation['locals']
                if 'kernel' in local: layers.append((local['kernel'], local['bias']))
                elif operation['direct'] in expressions: activations.append(expressions[operation['direct']])
                else: raise NotImplementedError('model operation {}'.format(operation['direct']))
            if len(set(activations)) != 1: raise NotImplementedError('one activation for all hidden layers')
            (xoperation,) = engine['xoperations']
            amplitude, ylimits = None, None
            for operation in engine['yoperations']:
                if 'limits' in operation['locals']: ylimits = operation['locals']['limits']
                elif "X['" in operation['inverse']:                                   # "v * jnp.exp(X['logA']) * 1e-10" / "v * (jnp.exp(X['logA']) * 1e-10)**2"
                    inverse = operation['inverse']
                    input_name = inverse.split("X['")[1].split("']")[0]
                    power = 2 if inverse.rstrip().endswith('**2') else 1
                    factor = float(re.search(r"\]\)\s*\*\s*([0-9.eE+\-]+)", inverse).group(1))
                    amplitude = (input_name, factor, power)
                else: raise NotImplementedError('y-operation {}'.format(operation['inverse']))
            engines[name] = StackedMLPEmulatorEngine(xoperation['locals']['limits'], layers, activations[0], ylimits, amplitude=amplitude)
            if params is None: params = [str(n) for n in engine['params']]
        fixed = state['fixed']
        return cls(params, engines, k=fixed['k'], ells=tuple(fixed['ells']), z=fixed['z'], param_specs=param_specs)

    def engine_specs(self):
        specs = {} if self.stacked else {'emu0': self.engines[self.table_name].spec(scalar=False)}
        for ie, name in [(1, 'sigma8'), (2, 'fsigma8')]:
            engine = self.engines.get(name, None)
            if engine is None: continue
            if np.ndim(engine) == 0 and not hasattr(engine, 'spec'):
                specs['emu{:d}'.format(ie)] = {'const': [float(engine)]}
            else:
                specs['emu{:d}'.format(ie)] = engine.spec(scalar=True)
        return specs


def emulate_power(likelihood, iobs=0, order=3, accuracy=2, delta_scale=1., params=None, engine='taylor', nsamples=4096, **mlp_kwargs):
    """``engine='mlp'``: MLP emulator trained on the GPU (:func:`fit_mlp`) on ``nsamples`` points of the R_d quasi-random sequence over the box ``value +- delta_scale x
    proposal`` of each parameter (the reference's ``QMCSampler`` + ``MLPEmulatorEngine``, samplers/qmc.py, emulators/__init__.py:510-533), the theory evaluated by
    ``dl_eval_theory`` as one batch whose output never leaves the device.  Default ``engine='taylor'``:
    Taylor emulator of the theory multipoles ``power [n_ell, n_k]`` of observable ``iobs`` of a GPU likelihood, as an :class:`EmulatedCalculator` for
    :class:`desilike_amd.theories.galaxy_clustering.EmulatedTracerPowerSpectrumMultipoles` (the reference's ``Emulator(theory, engine=TaylorEmulatorEngine(order)).fit()``,
    emulators/__init__.py:131-240): expansion around the parameters' default values with steps ``Parameter.delta`` (scaled by ``delta_scale``), the stencil
    evaluated by ``dl_eval_theory`` as one batch."""
    likelihood.initialize()
    ctx = likelihood._get_context()
    varied = likelihood.varied_params
    theory = likelihood.observables[iobs].wmatrix.theory
    names = [param.name for param in varied] if params is None else list(params)
    index = [varied.names().index(name) for name in names]
    center_all = np.array([param.value for param in varied], dtype='f8')
    delta = np.array([delta_scale * 0.5 * (varied[name].delta[1] + varied[name].delta[2]) for name in names], dtype='f8')

    def function(points):
        theta = np.repeat(center_all[None, :], len(points), axis=0)
        theta[:, index] = points
        return ctx.eval_theory_host(theta, iobs=iobs)

    if engine == 'mlp':
        import torch
        from .samplers import RQuasiRandomSequence
        half = np.array([delta_scale * varied[name].proposal for name in names], dtype='f8')
        lower, upper = np.array([max(c - h, varied[name].prior.limits[0]) for c, h, name in zip(center_all[index], half, names)]), np.array([min(c + h, varied[name].prior.limits[1]) for c, h, name in zip(center_all[index], half, names)])
        points = lower + RQuasiRandomSequence(len(names)).random(nsamples) * (upper - lower)
        theta = np.repeat(center_all[None, :], nsamples, axis=0)
        theta[:, index] = points
        device = torch.device('cuda', ctx.device)
        n_ell, n_kin = ctx.info('n_ell_obs{:d}'.format(iobs)), ctx.info('n_kin_obs{:d}'.format(iobs))
        power = torch.empty((nsamples, n_ell, n_kin), dtype=torch.float64, device=device)
        ctx.eval_theory(torch.as_tensor(theta, dtype=torch.float64, device=device).contiguous(), power, iobs=iobs)
        engine = fit_mlp(torch.as_tensor(points, dtype=torch.float64, device=device), power, device=ctx.device, **mlp_kwargs)
    else:
        engine = fit_taylor(function, center_all[index], delta, order=order, accuracy=accuracy)
    specs = {}
    for name in names:
        param = varied[name]
        specs[name] = dict(value=param.value, prior=dict(dist=param.prior.dist, limits=list(param.prior.limits), **({'loc': param.prior.loc, 'scale': param.prior.scale} if param.prior.dist == 'norm' else {})),
                           ref=dict(dist=param.ref.dist, limits=list(param.ref.limits), **({'loc': param.ref.loc, 'scale': param.ref.scale} if param.ref.dist == 'norm' else {})))
    return EmulatedCalculator(names, {'power': engine}, k=theory.k, ells=theory.ells, z=getattr(theory, 'z', 1.), param_specs=specs)
