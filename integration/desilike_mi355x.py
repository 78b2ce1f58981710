"""Reference-side binding: what a desilike maintainer adds to run the likelihood node of an EXISTING desilike pipeline on an MI355X.

This module imports ``desilike`` (the reference), never ``desilike_amd``'s Python mirror: it talks to ``libdesilike_amd.so`` through ctypes only
(include/desilike_amd.h).  Two pieces:

* :func:`extract_config` -- walks an *initialised* reference ``ObservablesGaussianLikelihood`` and returns the flat ``{key: array}`` set of ``dl_config`` (keys
  documented in the header).  Covered: Kaiser / EFT-like Kaiser tracer power spectrum AND correlation function multipoles on ShapeFit / Standard / Fixed templates;
  damped-BAO tracer power spectrum and correlation function multipoles ('standard' wiggle model, 'power' broadband terms); any window handled by
  ``WindowedPowerSpectrumMultipoles``; one or several observables, tracer namespaces; parameters solved analytically ('.marg' / '.best' / '.auto': shot-noise like
  terms, counter / stochastic terms, broadband terms); **emulated perturbation-theory nodes** -- the LPT / REPT velocileptors tracer power spectrum multipoles whose
  ``pt`` is an ``EmulatedCalculator`` (emulators/__init__.py:394-418) with Taylor or MLP engines (BASELINE configs[2]): :func:`_extract_emulated` -- one network per array,
  or the layout the reference ships (emulators/conversion.py:44-98: engines '11' / 'loop' / 'ct' / 'st', one network per (z, ell) stacked in each, amplitude rescale by the input
  ``logA``, redshift selection / blend of full_shape.py:1416-1443): :func:`_extract_stacked`.  For correlation functions ``get_corr`` (FFTLog) is linear in P_ell: the binding applies THE REFERENCE'S OWN
  ``theory.get_corr`` -- cosmoprimo's transform in a real installation -- to the unit vectors of the theory's k grid once, and hands the resulting operator to the
  device folded into the window matrix (desilike/theories/galaxy_clustering/base.py:127-136);
* :class:`MI355XGaussianLikelihood` -- a ``BaseGaussianLikelihood`` whose ``calculate`` is ONE ``dl_eval_batch_host`` call; ``evaluate(values [B, P])`` is the
  batched entry for samplers.

tests/golden/make_boundary_fixture.py runs :func:`extract_config` on real initialised reference likelihoods in the build container and commits the key set
as fixtures; tests/test_gpu_boundary.py creates contexts from those fixtures alone and checks the log-likelihoods the reference computed.
"""
import ctypes

import numpy as np

# enumerations of include/desilike_amd.h
DL_TEMPLATE_FIXED, DL_TEMPLATE_SHAPEFIT, DL_TEMPLATE_TURNOVER, DL_TEMPLATE_BANDS = 0, 1, 2, 3
DL_THEORY_KAISER, DL_THEORY_EFT_KAISER, DL_THEORY_BAO_DAMPED, DL_THEORY_EMULATED, DL_THEORY_TNS, DL_THEORY_PNG = 0, 1, 2, 3, 4, 5
DL_APMODE = {'qparqper': 0, 'qiso': 1, 'qap': 2, 'qisoqap': 3}


def _apmode(template):
    apeffect = getattr(template, 'apeffect', None)
    return DL_APMODE[getattr(apeffect, 'mode', 'qparqper')]


def _has(calculator, name):
    try:
        getattr(calculator, name)
        return True
    except AttributeError:
        return False


def hankel_operator(theory):
    """``[n_ell][n_s, n_k]``: the reference's ``get_corr`` (tgc/base.py:127-136: interpolation to the FFTLog grid, high-k tail, FFTLog, interpolation to s -- linear
    in P_ell) applied to the unit vectors of the theory's k grid; every multipole's row gets the same unit vector in one call."""
    nk, n_ell = len(theory.kin), len(theory.ells)
    columns = []
    for j in range(nk):
        unit = np.zeros((n_ell, nk), dtype='f8')
        unit[:, j] = 1.
        columns.append(np.asarray(theory.get_corr(unit), dtype='f8'))       # [n_ell, n_s]
    return np.stack(columns, axis=-1)                                        # [n_ell, n_s, n_k]


def _block_diag(blocks):
    n, m = sum(b.shape[0] for b in blocks), sum(b.shape[1] for b in blocks)
    out = np.zeros((n, m), dtype='f8')
    r = c = 0
    for b in blocks:
        out[r:r + b.shape[0], c:c + b.shape[1]] = b
        r += b.shape[0]; c += b.shape[1]
    return out


# ---- emulated perturbation-theory node (BASELINE configs[2]) ---------------------------------------------------------------------------
_VP_NAMES = ['b1', 'b2', 'bs', 'b3', 'alpha0', 'alpha2', 'alpha4', 'alpha6', 'sn0', 'sn2', 'sn4']       # velocileptors 'pars', full_shape.py:1290-1293
_ACTIVATIONS = [('silu', 0, lambda v: v / (1. + np.exp(-v))), ('relu', 1, lambda v: np.maximum(v, 0.)), ('tanh', 2, np.tanh)]   # emulators/conversion.py:27-34


def _affine(function, shape, what):
    """(offset, slope) of an elementwise affine map probed at 0, 1 and 2: the min-max scalers of the engines (emulators/conversion.py:75-79) -- whatever their
    expression strings are.  Anything else (scalers that use the inputs ``X``, PCA, ...) is refused."""
    try:
        v0, v1, v2 = (np.asarray(function(np.full(shape, value, dtype='f8')), dtype='f8') for value in (0., 1., 2.))
    except Exception as exc:
        raise NotImplementedError('{}: only constant elementwise affine operations are supported on the device ({})'.format(what, exc))
    if v0.shape != tuple(shape) or not np.allclose(v2 - v1, v1 - v0, rtol=1e-12, atol=1e-300):
        raise NotImplementedError('{}: not an elementwise affine operation'.format(what))
    return v0, v1 - v0


def _engine_description(engine):
    """The fitted state of one emulator engine, read off the attributes the reference itself writes and reads:

    * Taylor: ``center [P]``, ``powers [T, P]``, ``derivatives [T, *yshape]`` already divided by the factorials (emulators/__init__.py:471-507);
    * MLP: ``model_operations``: dense layers ``(v[..., None, :] @ kernel)[..., 0, :] + bias`` with ``_locals = {kernel, bias}`` and activations between them
      (emulators/conversion.py:20-35; ``operation._locals``: full_shape.py:1441), ``xoperations`` / ``yoperations``: min-max scalers (conversion.py:75-79).

    Returns ``dict(kind='taylor', center, powers, table [T, n_out])`` or ``dict(kind='mlp', xlimits [P, 2], act, hidden [(kernel, bias)...], table [H + 1, n_out])`` where
    ``table`` is the LAST LINEAR map of the engine (Taylor: the derivatives; MLP: final layer with the y-scaler folded in, last row = bias): output = basis . table."""
    yshape = tuple(engine.yshape)
    nout = int(np.prod(yshape, dtype='i8'))
    if hasattr(engine, 'derivatives'):
        if getattr(engine, 'xoperations', None) or getattr(engine, 'yoperations', None):
            raise NotImplementedError('Taylor engine with operations')      # the reference asserts the same: emulators/__init__.py:435-436
        center = np.asarray(engine.center, dtype='f8')
        powers = np.asarray(engine.powers, dtype='i4').reshape(-1, center.size)
        return dict(kind='taylor', center=center, powers=powers, table=np.asarray(engine.derivatives, dtype='f8').reshape(len(powers), nout))
    if not hasattr(engine, 'model_operations'):
        raise NotImplementedError('engine {}: Taylor and MLP engines are covered'.format(type(engine).__name__))
    layers, acts = [], []
    probe = np.array([-1.3, -0.2, 0., 0.7, 2.1])
    for operation in engine.model_operations:
        if 'kernel' in operation._locals:
            kernel, bias = np.asarray(operation._locals['kernel'], dtype='f8'), np.asarray(operation._locals['bias'], dtype='f8')
            if kernel.ndim != 2: raise NotImplementedError('stacked MLP kernels in a scalar engine')
            layers.append((kernel, bias.reshape(kernel.shape[1])))
        else:
            out = np.asarray(operation(probe), dtype='f8')
            match = [code for name, code, function in _ACTIVATIONS if np.allclose(out, function(probe), rtol=1e-14, atol=1e-15)]
            if not match: raise NotImplementedError('activation not one of silu / relu / tanh')
            acts.append(match[0])
    if len(layers) < 2 or len(acts) != len(layers) - 1 or len(set(acts)) != 1:
        raise NotImplementedError('MLP engines: >= 1 hidden layer, one activation for all of them')
    nin = layers[0][0].shape[0]
    xoperations = list(getattr(engine, 'xoperations', []))
    yoperations = list(getattr(engine, 'yoperations', []))

    def xscale(v):
        for operation in xoperations: v = operation(v)
        return v

    def yunscale(v):
        for operation in yoperations[::-1]: v = operation.inverse(v)
        return v

    x0, xslope = _affine(xscale, (nin,), 'xoperations')               # scaled = x0 + xslope x = (x - lo) / (hi - lo)
    lo = -x0 / xslope
    y0, yslope = _affine(yunscale, yshape, 'yoperations')             # y = y0 + yslope v
    kernel, bias = layers[-1]
    table = np.vstack([kernel * yslope.reshape(nout), bias * yslope.reshape(nout) + y0.reshape(nout)])
    return dict(kind='mlp', xlimits=np.column_stack([lo, lo + 1. / xslope]), act=acts[0], hidden=layers[:-1], table=table, last=(kernel, bias), ylimits=np.column_stack([y0.reshape(nout), (y0 + yslope).reshape(nout)]))


def _engine_keys(description, scalar):
    """``obs<i>.emu<e>.*`` keys (include/desilike_amd.h): the table engine (e = 0) ships its trunk only -- the last linear map is folded into the window matrix."""
    if description['kind'] == 'taylor':
        keys = dict(type=np.array([1], dtype='i4'), center=description['center'], powers=description['powers'])
        if scalar: keys['coef'] = description['table'].reshape(-1)
        return keys
    layers = description['hidden'] + ([description['last']] if scalar else [])
    widths = [layers[0][0].shape[0]] + [kernel.shape[1] for kernel, bias in layers]
    keys = dict(type=np.array([0], dtype='i4'), xlimits=description['xlimits'], widths=np.array(widths, dtype='i4'), act=np.array([description['act']], dtype='i4'),
                weights=np.concatenate([np.concatenate([kernel.ravel(), bias.ravel()]) for kernel, bias in layers]))
    if scalar: keys['ylimits'] = description['ylimits'][0]
    return keys


def _call(operation, v, X, inverse=False):
    """An emulator operation on ``v`` with the inputs ``X`` at hand (operations may use them by name: emulators/conversion.py:88-92)."""
    return (operation.inverse if inverse else operation)(v, X=X)


def _stacked_engine(engine, X0, what):
    """One MLP engine whose kernels may be stacked over leading axes (emulators/conversion.py:58-66): ``kernel [*S, in, out]``, ``bias [*S, out]``; the network output
    ``[*S, out]`` is reshaped to ``yshape`` (a flat reinterpretation).  Everything else is PROBED, not parsed: the x-scaler is an elementwise affine map; the y-operations are
    ``y = s(X) (y0 + yslope v)`` elementwise in ``v`` with ONE amplitude ``s`` per engine that is log-linear in the inputs, ``log s = sum_j a_j x_j + b`` (conversion.py:88-92:
    ``v * exp(logA) * 1e-10`` and its square are of this form); anything else raises.  Returns dict(stack, layers [(kernel [T, in, out], bias [T, out])], act, x0, xslope,
    y0, yslope [n_out_total] (flat ``yshape`` order, at s = 1), loga [n_x + 1])."""
    yshape = tuple(engine.yshape)
    ntot = int(np.prod(yshape, dtype='i8'))
    layers, acts = [], []
    probe = np.array([-1.3, -0.2, 0., 0.7, 2.1])
    for operation in engine.model_operations:
        if 'kernel' in operation._locals:
            kernel, bias = np.asarray(operation._locals['kernel'], dtype='f8'), np.asarray(operation._locals['bias'], dtype='f8')
            stack = kernel.shape[:-2]
            if layers and stack != layers[0][0].shape[:-2]: raise NotImplementedError('{}: layers stacked differently'.format(what))
            layers.append((kernel, bias.reshape(stack + (kernel.shape[-1],))))
        else:
            out = np.asarray(operation(probe), dtype='f8')
            match = [code for name, code, function in _ACTIVATIONS if np.allclose(out, function(probe), rtol=1e-14, atol=1e-15)]
            if not match: raise NotImplementedError('activation not one of silu / relu / tanh')
            acts.append(match[0])
    if len(layers) < 2 or len(acts) != len(layers) - 1 or len(set(acts)) != 1:
        raise NotImplementedError('MLP engines: >= 1 hidden layer, one activation for all of them')
    stack = layers[0][0].shape[:-2]
    T = int(np.prod(stack, dtype='i8'))
    nin, nout = layers[0][0].shape[-2], layers[-1][0].shape[-1]
    if T * nout != ntot: raise NotImplementedError('{}: network outputs do not fill yshape'.format(what))
    names = [str(n) for n in engine.params]

    def xscale(v):
        for operation in getattr(engine, 'xoperations', []): v = _call(operation, v, X0)
        return v

    def yunscale(v, X):
        for operation in list(getattr(engine, 'yoperations', []))[::-1]: v = _call(operation, v, X, inverse=True)
        return np.asarray(v, dtype='f8')

    x0, xslope = _affine(xscale, (nin,), what + ' xoperations')

    def affine_y(X):
        v0, v1, v2 = (yunscale(np.full(yshape, value, dtype='f8'), X).reshape(ntot) for value in (0., 1., 2.))
        if not np.allclose(v2 - v1, v1 - v0, rtol=1e-12, atol=1e-300): raise NotImplementedError('{} yoperations: not elementwise affine'.format(what))
        return v0, v1 - v0

    y0, yslope = affine_y(X0)
    ref = np.concatenate([y0, yslope])
    live = ref != 0.
    loga = np.zeros(nin + 1)
    for j, name in enumerate(names):
        step = 0.01 * max(abs(X0[name]), 1.)
        ratios = []
        for mult in (1., 2.):
            X = dict(X0); X[name] = X0[name] + mult * step
            cur = np.concatenate(affine_y(X))
            if np.any(cur[~live] != 0.): raise NotImplementedError('{} yoperations: offsets that depend on the inputs'.format(what))
            ratio = cur[live] / ref[live]
            if ratio.size and not np.allclose(ratio, ratio[0], rtol=1e-12, atol=0.): raise NotImplementedError('{} yoperations: more than one amplitude per engine'.format(what))
            ratios.append(ratio[0] if ratio.size else 1.)
        if ratios[0] <= 0. or abs(np.log(ratios[1]) - 2. * np.log(ratios[0])) > 1e-10 * max(1., abs(np.log(ratios[1]))):
            raise NotImplementedError('{} yoperations: amplitude not log-linear in {}'.format(what, name))
        loga[j] = np.log(ratios[1]) / (2. * step)
        if abs(loga[j]) < 1e-13: loga[j] = 0.
    # y0, yslope above hold the amplitude at X0: divide it out (s(X0) = exp(loga . x0 + b) with b chosen such that what is stored is at s = 1)
    s0 = np.exp(sum(loga[j] * X0[name] for j, name in enumerate(names)))
    if np.any(loga[:nin] != 0.): y0, yslope = y0 / s0, yslope / s0
    return dict(stack=stack, T=T, nin=nin, nout=nout, yshape=yshape, act=acts[0], x0=x0, xslope=xslope, y0=y0, yslope=yslope, loga=loga,
                layers=[(kernel.reshape((T,) + kernel.shape[-2:]), bias.reshape(T, -1)) for kernel, bias in layers])


def _table_operator(emulator, shapes, X0, select):
    """The constant linear map from the engines' outputs to the table the tracer combines, PROBED through the emulator's own operations (the split / concatenate /
    moveaxis pair of emulators/conversion.py:50-51, the redshift blend of full_shape.py:1443, the tracer's own selection -- ``select(state)``): every element of the final
    table is a weighted sum of at most one element per slab (index along the first axis) of every engine output.  Per engine and slab: (weights, flat source index or -1).
    Verified on random outputs before it is trusted."""
    def push(outputs):
        state = {name: np.array(value) for name, value in outputs.items()}
        state.update(emulator.fixed)
        for operation in list(emulator.yoperations)[::-1]: state = _call(operation, state, X0, inverse=True)
        return np.asarray(select(state), dtype='f8')

    zeros = {name: np.zeros(shape) for name, shape in shapes.items()}
    base = push(zeros)
    if np.any(base != 0.): raise NotImplementedError('emulator-level operations with a constant term')
    table = {}
    for name, shape in shapes.items():
        slab = int(np.prod(shape[1:], dtype='i8'))
        table[name] = []
        for index in range(shape[0]):
            ones, codes = dict(zeros), dict(zeros)
            ones[name] = np.zeros(shape); ones[name][index] = 1.
            codes[name] = np.zeros(shape); codes[name][index] = np.arange(1., slab + 1.).reshape(shape[1:])
            weight, value = push(ones), push(codes)
            live = weight != 0.
            source = np.full(weight.shape, -1, dtype='i8')
            code = value[live] / weight[live]
            if not np.allclose(code, np.round(code), rtol=0., atol=1e-6): raise NotImplementedError('emulator-level operations that mix elements of one slab')
            source[live] = np.round(code).astype('i8') - 1 + index * slab
            table[name].append((weight, source))
    rng = np.random.RandomState(0)
    outputs = {name: rng.standard_normal(shape) for name, shape in shapes.items()}
    expected, mine = push(outputs), np.zeros_like(base)
    for name in shapes:
        flat = outputs[name].ravel()
        for weight, source in table[name]: mine += np.where(source >= 0, weight * flat[np.maximum(source, 0)], 0.)
    if not np.allclose(mine, expected, rtol=1e-12, atol=1e-12 * np.abs(expected).max()): raise NotImplementedError('emulator-level operations are not the sparse linear map they were probed as')
    return table


def _extract_stacked(cfg, p, obs, theory, pt, column, value_of, sindex, solved):
    """The emulator layout the reference ships (emulators/conversion.py:44-98) under its velocileptors tracers: SEVERAL table engines ('11', 'loop', 'ct', 'st': one per group
    of bias monomials), each holding one network per (z, ell); outputs rescaled by an amplitude that depends on the inputs (conversion.py:88-92); emulator-level operations that
    assemble ``pktable [n_ell, n_k, 19, n_z]`` (50-51) and select / blend redshifts (full_shape.py:1416-1443, inserted by the emulated REPT node).  Everything after the
    networks' last hidden layers is linear with constant coefficients times ONE scalar per engine: the final layers, the y-scalers, the assembly, the blend, the tracer's
    redshift selection (full_shape.py:1486), the k-interpolation (1598) and the window are multiplied together here; networks that end up with a zero operator (redshifts
    not bracketing the tracer's, multipoles it does not use) are dropped.  Device keys: ``obs<i>.emu0.type = 2`` (include/desilike_amd.h)."""
    from desilike.jax import interp1d
    wm = obs.wmatrix
    emulator = pt.emulator
    name = type(theory).__name__
    rept = name.startswith('REPT')
    if not (rept or name.startswith('LPT')) or 'Tracer' not in name or hasattr(theory, 'get_corr'):
        raise NotImplementedError('emulated node under {}: the LPT / REPT velocileptors tracer power spectrum multipoles are covered'.format(name))
    physical = bool(theory.is_physical_prior)
    tnames = [n for n in emulator.engines if n not in ('sigma8', 'fsigma8')]
    xnames = [str(n) for n in emulator.engines[tnames[0]].params]
    ptnames = {param.basename: param.name for param in pt.all_params}
    X0 = {n: float(pt.all_params[ptnames.get(n, n)].value) for n in xnames}
    engines = {}
    for n in tnames:
        if [str(q) for q in emulator.engines[n].params] != xnames: raise NotImplementedError('engines with different inputs')
        if not hasattr(emulator.engines[n], 'model_operations'): raise NotImplementedError('engine {}: several table engines must be MLPs'.format(n))
        engines[n] = _stacked_engine(emulator.engines[n], X0, 'engine ' + n)
    first = engines[tnames[0]]
    widths = [first['nin']] + [kernel.shape[-1] for kernel, bias in first['layers'][:-1]]
    for n in tnames:
        e = engines[n]
        if [e['nin']] + [kernel.shape[-1] for kernel, bias in e['layers'][:-1]] != widths or e['act'] != first['act']:
            raise NotImplementedError('stacked engines: every network must have the same hidden layers and activation')
        if not (np.array_equal(e['x0'], first['x0']) and np.array_equal(e['xslope'], first['xslope'])):
            # another min-max scaler of the inputs: expressed through the first engine's, in the first layer u = x0 + xslope x -> scaled_e = c + r u
            r = e['xslope'] / first['xslope']
            c = e['x0'] - r * first['x0']
            kernel, bias = e['layers'][0]
            e['layers'][0] = (kernel * r[None, :, None], bias + np.einsum('j,tjo->to', c, kernel))
    H = widths[-1]

    ptells = list(pt.ells)
    index = [ptells.index(ell) for ell in theory.ells]

    def select(state):   # what the tracer combines: REPTVelocileptorsPowerSpectrumMultipoles.combine_bias_terms_poles picks its redshift (full_shape.py:1486), then its multipoles (1597)
        table = np.asarray(state['pktable'])
        zz = np.asarray(state.get('z', pt.z))
        if rept and zz.ndim: table = table[..., [float(v) for v in zz].index(float(theory.z))]
        if table.ndim != 3 or table.shape[-1] != 19: raise NotImplementedError('pktable of shape {}'.format(table.shape))
        return table[index]

    operator = _table_operator(emulator, {n: engines[n]['yshape'] for n in tnames}, X0, select)
    kpt, k = np.asarray(pt.k, dtype='f8'), np.asarray(theory.k, dtype='f8')
    nell, nkpt = len(index), kpt.size
    interp = np.asarray(interp1d(k, kpt, np.eye(kpt.size)), dtype='f8')                     # the reference's own interpolation (desilike/jax.py:211-265)
    window = None if wm.matrix_full is None else np.asarray(wm.matrix_full, dtype='f8')
    groups, scale, weights, blocks, folds = [], [], [], [], []
    ntrunk = 0
    for n in tnames:
        e = engines[n]
        kl, bl = e['layers'][-1]                                                            # [T, H, nout], [T, nout]
        jac = np.zeros((e['T'], H, nell * nkpt * 19))                                       # d table / d (amplitude x last hidden layer of network t)
        const = np.zeros(nell * nkpt * 19)
        for weight, source in operator[n]:
            weight, source = weight.ravel(), source.ravel()
            live = np.flatnonzero(source >= 0)
            f = source[live]
            t, o = f // e['nout'], f % e['nout']
            const[live] += weight[live] * (e['y0'][f] + e['yslope'][f] * bl[t, o])
            jac[t, :, live] += (weight[live] * e['yslope'][f])[:, None] * kl[t, :, o]
        used = np.flatnonzero(np.any(jac.reshape(e['T'], -1) != 0., axis=1))               # networks that reach the tracer's table at all
        monos = np.flatnonzero(np.any(jac.reshape(-1, nell * nkpt, 19) != 0., axis=(0, 1)) | np.any(const.reshape(nell * nkpt, 19) != 0., axis=0))
        if monos.size == 0: continue
        m0, m1 = int(monos[0]), int(monos[-1]) + 1
        basis = np.concatenate([jac[used].reshape(used.size * H, nell, nkpt, 19), const.reshape(1, nell, nkpt, 19)], axis=0)[..., m0:m1]     # [K_g, ell, kpt, m]
        fold = np.einsum('kq,hlqm->lkhm', interp, basis).reshape(nell * k.size, -1)        # columns (h, m): group by group
        folds.append(fold)
        blocks.append(fold if window is None else window.dot(fold))
        groups.append([ntrunk, ntrunk + used.size, m0, m1])
        scale.append(np.concatenate([e['loga'][:e['nin']], [0.]]))
        for t in used: weights.append(np.concatenate([np.concatenate([kernel[t].ravel(), bias[t].ravel()]) for kernel, bias in e['layers'][:-1]]))
        ntrunk += used.size
    cfg[p + 'theory'] = np.array([DL_THEORY_EMULATED], dtype='i4')
    cfg[p + 'transform'] = np.array([1 if getattr(obs, 'transform', None) == 'cubic' else 0], dtype='i4')
    cfg[p + 'mono_mode'] = np.array([{(True, False): 1, (True, True): 2, (False, False): 3, (False, True): 4}[(physical, rept)]], dtype='i4')
    cfg[p + 'vconst'] = np.array([theory.snd, theory.fsat, theory.options['sigv'] if physical else 1., theory.nd], dtype='f8')
    cfg[p + 'in.x'] = np.array([column(ptnames.get(n, n), value_of(ptnames.get(n, n), 0.)) for n in xnames], dtype='f8')
    names = {param.basename: param.name for param in theory.all_params}
    vp = [names.get(n + ('p' if physical else ''), n + ('p' if physical else '')) for n in _VP_NAMES]
    defaults = dict(theory.required_bias_params)
    cfg[p + 'in.vp'] = np.array([column(n, value_of(n, defaults.get(b + ('p' if physical else ''), 0.))) for n, b in zip(vp, _VP_NAMES)], dtype='f8')
    lo = -first['x0'] / first['xslope']
    cfg[p + 'emu0.type'] = np.array([2], dtype='i4')
    cfg[p + 'emu0.widths'], cfg[p + 'emu0.act'] = np.array(widths, dtype='i4'), np.array([first['act']], dtype='i4')
    cfg[p + 'emu0.xlimits'] = np.column_stack([lo, lo + 1. / first['xslope']])
    cfg[p + 'emu0.weights'] = np.concatenate(weights)
    cfg[p + 'emu0.groups'], cfg[p + 'emu0.scale'] = np.array(groups, dtype='i4'), np.array(scale, dtype='f8')
    for ie, ename in [(1, 'sigma8'), (2, 'fsigma8')]:
        if ename in emulator.engines:
            if emulator.yoperations or tuple(emulator.engines[ename].yshape) not in ((), (1,)): raise NotImplementedError('scalar engines under emulator-level operations')
            if [str(q) for q in emulator.engines[ename].params] != xnames: raise NotImplementedError('engines with different inputs')
            keys = _engine_keys(_engine_description(emulator.engines[ename]), scalar=True)
        elif physical: keys = {'const': np.array([float(getattr(pt, ename))], dtype='f8')}
        else: keys = {'const': np.array([1.], dtype='f8')}                                 # not used by the direct basis (full_shape.py:1593-1594)
        for key, value in keys.items(): cfg[p + 'emu{:d}.{}'.format(ie, key)] = value
    offset = None if getattr(wm, 'offset', None) is None else np.asarray(wm.offset, dtype='f8')
    shotnoisein = np.asarray(wm.shotnoisein, dtype='f8')
    if np.any(shotnoisein != 0.):
        vector = np.repeat(shotnoisein, len(wm.kin))
        extra = vector if window is None else window.dot(vector)
        offset = extra if offset is None else offset + extra
    cfg[p + 'wmatrix'] = np.hstack(blocks)
    if offset is not None: cfg[p + 'offset'] = offset
    if getattr(wm, 'kmask', None) is not None: cfg[p + 'kmask'] = np.asarray(wm.kmask, dtype='i4')
    cfg[p + 'shotnoise_out'] = np.asarray(wm.shotnoiseout, dtype='f8')
    cfg[p + 'flatdata'] = np.asarray(obs.flatdata, dtype='f8')
    if solved: cfg[p + 'marg.vp'] = np.array([sindex(n) for n in vp], dtype='i4')
    # the description against the emulator itself, at points other than the one it was probed at: pktable as the tracer sees it
    rng = np.random.RandomState(1)
    limits = cfg[p + 'emu0.xlimits']
    for trial in range(3):
        x = limits[:, 0] + rng.uniform(0.2, 0.8, len(xnames)) * (limits[:, 1] - limits[:, 0])
        params = {param.basename: float(param.value) for param in pt.all_params}
        params.update(dict(zip(xnames, x)))
        expected = select({**emulator.fixed, **emulator.predict(params)})
        mine = np.zeros((nell * k.size, 19))
        u = first['x0'] + first['xslope'] * x
        activation = [function for name_, code, function in _ACTIVATIONS if code == first['act']][0]
        for (t0, t1, m0, m1), loga, block in zip(groups, scale, folds):
            basis = []
            for t in range(t0, t1):
                v, pos = u, 0
                w = weights[t]
                for nin_, nout_ in zip(widths[:-1], widths[1:]):
                    v = v.dot(w[pos:pos + nin_ * nout_].reshape(nin_, nout_)) + w[pos + nin_ * nout_:pos + nin_ * nout_ + nout_]
                    pos += nin_ * nout_ + nout_
                    v = activation(v)
                basis.append(v)
            basis = np.concatenate(basis + [[1.]])
            mine[:, m0:m1] += np.exp(loga[:-1].dot(x) + loga[-1]) * block.reshape(nell * k.size, basis.size, m1 - m0).transpose(0, 2, 1).dot(basis)
        expected = np.einsum('kq,lqm->lkm', interp, expected).reshape(nell * k.size, 19)
        if not np.allclose(mine, expected, rtol=1e-11, atol=1e-12 * np.abs(expected).max()):
            raise NotImplementedError('stacked emulator: the folded description does not reproduce emulator.predict (max deviation {:.3e})'.format(np.abs(mine - expected).max()))


def _extract_emulated(cfg, p, obs, theory, pt, column, value_of, sindex, solved):
    """Velocileptors-type tracer theory whose ``pt`` is an ``EmulatedCalculator`` (emulators/__init__.py:394-418): ``pktable [n_ell, n_kpt, 19]``, ``sigma8``, ``fsigma8``
    come from ``pt.emulator.engines``; the tracer combines the 19 bias monomials (full_shape.py:1182-1186) and interpolates to its own k (full_shape.py:1312, 1598).
    Everything after the engines' trunks is constant and linear -- last layer (Taylor: derivative table), k-interpolation, window matrix: multiplied together here once
    and handed over as ``obs<i>.wmatrix`` with ``n_basis * 19`` columns; the device evaluates the trunk and the monomials per point."""
    from desilike.jax import interp1d
    wm = obs.wmatrix
    emulator = pt.emulator
    if getattr(emulator, 'xoperations', None):
        raise NotImplementedError('emulator-level operations on the inputs are not covered')
    engines = emulator.engines
    stacked = any(np.ndim(operation._locals.get('kernel', 0.)) > 2 for engine in engines.values() for operation in getattr(engine, 'model_operations', []))
    if getattr(emulator, 'yoperations', None) or 'pktable' not in engines or stacked:     # the layout of emulators/conversion.py:44-98
        return _extract_stacked(cfg, p, obs, theory, pt, column, value_of, sindex, solved)
    name = type(theory).__name__
    rept = name.startswith('REPT')
    if not (rept or name.startswith('LPT')) or 'Tracer' not in name or hasattr(theory, 'get_corr'):
        raise NotImplementedError('emulated node under {}: the LPT / REPT velocileptors tracer power spectrum multipoles are covered'.format(name))
    physical = bool(theory.is_physical_prior)
    table = _engine_description(engines['pktable'])
    xnames = [str(n) for n in engines['pktable'].params]
    ptnames = {param.basename: param.name for param in pt.all_params}
    cfg[p + 'theory'] = np.array([DL_THEORY_EMULATED], dtype='i4')
    cfg[p + 'transform'] = np.array([1 if getattr(obs, 'transform', None) == 'cubic' else 0], dtype='i4')
    cfg[p + 'mono_mode'] = np.array([{(True, False): 1, (True, True): 2, (False, False): 3, (False, True): 4}[(physical, rept)]], dtype='i4')
    cfg[p + 'vconst'] = np.array([theory.snd, theory.fsat, theory.options['sigv'] if physical else 1., theory.nd], dtype='f8')      # full_shape.py:1154-1157
    cfg[p + 'in.x'] = np.array([column(ptnames.get(n, n), value_of(ptnames.get(n, n), 0.)) for n in xnames], dtype='f8')
    names = {param.basename: param.name for param in theory.all_params}
    vp = [names.get(n + ('p' if physical else ''), n + ('p' if physical else '')) for n in _VP_NAMES]
    defaults = dict(theory.required_bias_params)                                                                                   # 0, except b1 = 1 (full_shape.py:1290-1293)
    cfg[p + 'in.vp'] = np.array([column(n, value_of(n, defaults.get(b + ('p' if physical else ''), 0.))) for n, b in zip(vp, _VP_NAMES)], dtype='f8')
    for ie, ename in [(0, 'pktable'), (1, 'sigma8'), (2, 'fsigma8')]:
        if ename in engines:
            if ie and [str(n) for n in engines[ename].params] != xnames: raise NotImplementedError('engines with different inputs')
            keys = _engine_keys(table if ie == 0 else _engine_description(engines[ename]), scalar=ie > 0)
        else:                                                          # a fixed output of the emulated calculator (emulator.fixed -> attribute: emulators/__init__.py:386-388)
            keys = {'const': np.array([float(getattr(pt, ename))], dtype='f8')}
        for key, value in keys.items(): cfg[p + 'emu{:d}.{}'.format(ie, key)] = value
    # fold: theory vector [n_ell * n_k] = fold . phi, phi[(h, m)] = basis_h(x) monomial_m(pars)
    kpt, k = np.asarray(pt.k, dtype='f8'), np.asarray(theory.k, dtype='f8')
    ptells = list(pt.ells)
    index = [ptells.index(ell) for ell in theory.ells]
    nb = table['table'].shape[0]
    tab = table['table'].reshape(nb, len(ptells), kpt.size, 19)[:, index]                                                         # [h, ell, kpt, m]
    interp = np.asarray(interp1d(k, kpt, np.eye(kpt.size)), dtype='f8')                                                           # the reference's own interpolation (desilike/jax.py:211-265), linear in the table
    fold = np.einsum('kq,hlqm->lkhm', interp, tab).reshape(len(index) * k.size, nb * 19)
    window = None if wm.matrix_full is None else np.asarray(wm.matrix_full, dtype='f8')
    offset = None if getattr(wm, 'offset', None) is None else np.asarray(wm.offset, dtype='f8')
    shotnoisein = np.asarray(wm.shotnoisein, dtype='f8')
    if np.any(shotnoisein != 0.):    # W . (sn_in (x) 1) (window.py:471) does not map onto the feature columns: into the offset
        vector = np.repeat(shotnoisein, len(wm.kin))
        extra = vector if window is None else window.dot(vector)
        offset = extra if offset is None else offset + extra
    cfg[p + 'wmatrix'] = fold if window is None else window.dot(fold)
    if offset is not None: cfg[p + 'offset'] = offset
    if getattr(wm, 'kmask', None) is not None: cfg[p + 'kmask'] = np.asarray(wm.kmask, dtype='i4')
    cfg[p + 'shotnoise_out'] = np.asarray(wm.shotnoiseout, dtype='f8')
    cfg[p + 'flatdata'] = np.asarray(obs.flatdata, dtype='f8')
    if solved: cfg[p + 'marg.vp'] = np.array([sindex(n) for n in vp], dtype='i4')


def extract_config(likelihood):
    """Flat ``dl_config`` of an initialised reference likelihood: ``{key: float64 / int32 array}``, plus ``varied`` (theta column order) under ``'__varied__'`` and the
    analytically solved parameters under ``'__solved__'``."""
    varied = likelihood.varied_params.names()
    solved_params = [param for param in likelihood.all_params if param.solved and not str(param.derived).startswith('.prec')]
    solved = [param.name for param in solved_params]
    cfg = {}

    def column(name, default):
        return np.array([varied.index(name) if name in varied else -1, default], dtype='f8')

    def sindex(name):
        return solved.index(name) if name in solved else -1

    def value_of(name, default):
        return float(likelihood.all_params[name].value) if name in likelihood.all_params else default

    cfg['n_params'] = np.array([len(varied)], dtype='i4')
    cfg['n_obs'] = np.array([len(likelihood.observables)], dtype='i4')
    cfg['precision'] = np.asarray(likelihood.precision, dtype='f8')
    priors = []
    for param in likelihood.varied_params:
        prior = param.prior
        kinds = ['uniform', 'norm', 'expon', 'laplace', 'cauchy', 'logistic', 'halfnorm', 'halfcauchy', 'gumbel_r', 'gumbel_l']   # csrc/dl_prior.h
        if prior.dist not in kinds or (kinds.index(prior.dist) >= 2 and prior.is_limited()):
            raise NotImplementedError('prior {} of {} is not supported on the device'.format(prior, param.name))
        kind = kinds.index(prior.dist)
        priors.append([float(kind), prior.limits[0], prior.limits[1], prior.attrs.get('loc', 0.) if kind else 0., prior.attrs.get('scale', 1.) if kind else 1.])
    cfg['priors'] = np.array(priors, dtype='f8')
    if solved:
        # likelihoods/base.py:314-413: '.marg' (1) / '.best' (0) per solved parameter, Gaussian prior (loc, 1 / scale^2; flat: 0), expansion point = current value
        default = getattr(likelihood, 'solved_default', '.marg')
        kind, mprior, x0 = [], [], []
        for param in solved_params:
            derived = str(param.derived)
            if derived.startswith('.auto'): derived = derived.replace('.auto', default)
            kind.append(1 if derived.startswith('.marg') else 0)
            loc, scale = (param.prior.attrs['loc'], param.prior.attrs['scale']) if param.prior.dist == 'norm' else (0., np.inf)
            mprior.append([loc, scale**(-2)])
            x0.append(float(param.value))
        cfg['marg.kind'], cfg['marg.prior'], cfg['marg.x0'] = np.array(kind, dtype='i4'), np.array(mprior, dtype='f8'), np.array(x0, dtype='f8')
    for iobs, obs in enumerate(likelihood.observables):
        p = 'obs{:d}.'.format(iobs)
        wm = obs.wmatrix
        theory = wm.theory
        node = theory.pt if _has(theory, 'pt') else None
        if node is not None and _has(node, 'emulator'):                     # emulated perturbation-theory node (emulators/__init__.py:394-418): BASELINE configs[2]
            _extract_emulated(cfg, p, obs, theory, node, column, value_of, sindex, solved)
            continue
        xi = hasattr(theory, 'get_corr')                                    # correlation function multipoles: Hankel transform of a power spectrum theory
        ptheory = theory.power if xi else theory                           # the tracer power spectrum calculator ...
        pt = ptheory.pt if 'pt' in ptheory.__dict__ or hasattr(type(ptheory), 'pt') or _has(ptheory, 'pt') else ptheory   # ... (the BAO correlation function classes hold the wiggle calculator itself: bao.py:881-905)
        if pt is ptheory: ptheory = theory
        template = pt.template
        bao = hasattr(pt, 'smoothing_radius')
        eft = hasattr(ptheory, 'counterterm_matrix')
        shapefit = template.__class__.__name__.startswith('ShapeFit')
        tns = pt.__class__.__name__.startswith('TNS')                       # the reference's own one-loop producer (full_shape.py:836-971)
        pngvel = pt.__class__.__name__.startswith('PNGTracerVelocity')     # its tracer-velocity variant (primordial_non_gaussianity.py:196-330)
        png = pt.__class__.__name__.startswith('PNGTracerPower') or pngvel  # scale-dependent bias (primordial_non_gaussianity.py:12-116)
        cfg[p + 'theory'] = np.array([DL_THEORY_BAO_DAMPED if bao else DL_THEORY_TNS if tns else DL_THEORY_PNG if png else DL_THEORY_EFT_KAISER if eft else DL_THEORY_KAISER], dtype='i4')
        if tns:
            k = np.asarray(pt.k, dtype='f8')
            cfg[p + 'tns_k11'] = np.linspace(k[0] * 0.7, k[-1] * 1.3, int(len(k) * 1.6 + 0.5))                # full_shape.py:875 (a local of calculate)
            x, w = np.polynomial.legendre.leggauss(20)                                                         # utils.weights_mu(10, 'leggauss'): full_shape.py:757
            cfg[p + 'tns_mu'], cfg[p + 'tns_wmu'] = x[10:], (w[10:] + w[9::-1]) / 2.
            cfg[p + 'tns_fog'] = np.array([{'lorentzian': 0, 'gaussian': 1}[pt.options['fog']]], dtype='i4')
        turnover = template.__class__.__name__.startswith('TurnOver')      # power_template.py:1293-1340
        bands = template.__class__.__name__.startswith('BandVelocity')      # power_template.py:868-970
        cfg[p + 'template'] = np.array([DL_TEMPLATE_BANDS if bands else DL_TEMPLATE_TURNOVER if turnover else DL_TEMPLATE_SHAPEFIT if shapefit and not bao else DL_TEMPLATE_FIXED], dtype='i4')
        if turnover: cfg[p + 'kto_fid'], cfg[p + 'pkto_fid'] = np.array([template.kTO_fid], dtype='f8'), np.array([template.pkTO_dd_fid], dtype='f8')
        cfg[p + 'apmode'] = np.array([_apmode(template)], dtype='i4')
        cfg[p + 'transform'] = np.array([1 if getattr(obs, 'transform', None) == 'cubic' else 0], dtype='i4')
        cfg[p + 'eta'] = np.array([getattr(getattr(template, 'apeffect', None), 'eta', 1. / 3.)], dtype='f8')
        cfg[p + 'f_fid'] = np.array([template.f_fid], dtype='f8')
        cfg[p + 'a'] = np.array([getattr(template, 'a', 0.6)], dtype='f8')
        cfg[p + 'kp'] = np.array([getattr(template, 'kp', 0.03)], dtype='f8')
        cfg[p + 'nd'] = np.array([1. if bao else getattr(ptheory, 'nd', 1.)], dtype='f8')
        cfg[p + 'ells_in'] = np.asarray(ptheory.ells if xi else wm.ellsin, dtype='i4')
        cfg[p + 'kin'] = np.asarray(pt.k, dtype='f8')
        cfg[p + 'mu'], cfg[p + 'wmu_ell'] = np.asarray(pt.mu, dtype='f8'), np.asarray(pt.wmu, dtype='f8')
        if pngvel:   # 81 trapezoid nodes on [-1, 1]: the integrand times an odd Legendre polynomial is even in mu -- the nodes mu >= 0 with the mirror weights added
            half = cfg[p + 'mu'] >= -1e-12
            cfg[p + 'wmu_ell'] = np.where(np.abs(cfg[p + 'mu'][half]) > 0., 2., 1.) * cfg[p + 'wmu_ell'][:, half]
            cfg[p + 'mu'] = np.abs(cfg[p + 'mu'][half])
            cfg[p + 'png_velocity'], cfg[p + 'png_velfac'] = np.array([1], dtype='i4'), np.array([100. / (1. + float(pt.z))], dtype='f8')
        cfg[p + 'k_t'], cfg[p + 'pk_dd_fid'] = np.asarray(template.k, dtype='f8'), np.asarray(template.pk_dd_fid, dtype='f8')
        if png:
            if pt.method != 'prim' or pt.mode not in ('bphi', 'b-p'):
                raise NotImplementedError("PNG theory: method 'prim' and modes 'bphi' / 'b-p' are covered")
            kt, cosmo = np.asarray(template.k, dtype='f8'), template.cosmo
            pphi_prim = 9 / 25 * 2 * np.pi**2 / kt**3 * cosmo.get_primordial(mode='scalar').pk_interpolator()(kt) / cosmo.h**3          # primordial_non_gaussianity.py:83-86, at the fiducial template
            alpha = 1. / (np.asarray(template.pk_dd_fid, dtype='f8') / pphi_prim)**0.5
            # the first wavenumber only normalises the transfer function of the other method (line 95): not a knot of the interpolation
            cfg[p + 'k_t'], cfg[p + 'pk_dd_fid'], cfg[p + 'png_alpha'] = kt[1:], cfg[p + 'pk_dd_fid'][1:], alpha[1:]
            cfg[p + 'png_mode'] = np.array([{'bphi': 0, 'b-p': 1}[pt.mode]], dtype='i4')
        cfg[p + 'flatdata'] = np.asarray(obs.flatdata, dtype='f8')
        # parameter -> theta column (or constant): tracer namespaces prefix the bias / shot-noise parameters (full_shape.py:88-128)
        names = {param.basename: param.name for param in theory.all_params}
        names.update({param.basename: param.name for param in ptheory.all_params})
        names.update({param.basename: param.name for param in pt.all_params})
        # pass-through columns appended to the theory vector: broadband terms of the BAO classes (bao.py:495-534, 881-905), in the order of the multipoles
        pass_names, pass_matrix = [], None
        resummed = flexible = False
        if bao:
            # wiggle model (include/desilike_amd.h, obs<i>.bao_mode bits 4-8): 'standard' (bao.py:123-136); 8 | fix-damping 1 | move-all 2 | fog-damping 4 (137-150);
            # 16 | move-all | fog-damping: resummed wiggles (165-266); 32 | move-all: flexible wiggles (269-391)
            clsname, model = type(pt).__name__, str(getattr(pt, 'model', 'standard'))
            resummed, flexible = clsname.startswith('ResummedBAO'), clsname.startswith('FlexibleBAO')
            bits = 0
            if flexible: bits = 32 | (2 if 'move-all' in model else 0)
            elif resummed: bits = 16 | (2 if 'move-all' in model else 0) | (4 if 'fog-damping' in model else 0)
            elif model != 'standard':
                if any(word not in ('fix-damping', 'move-all', 'fog-damping') for word in model.replace('_', ' ').split()):
                    raise NotImplementedError('wiggle model {}: standard, fix-damping, move-all, fog-damping and their combinations are covered'.format(model))
                bits = 8 | (1 if 'fix-damping' in model else 0) | (2 if 'move-all' in model else 0) | (4 if 'fog-damping' in model else 0)
            cfg[p + 'pknow_dd_fid'] = np.asarray(template.pknow_dd_fid, dtype='f8')
            cfg[p + 'bao_mode'] = np.array([(1 if pt.mode == 'reciso' else 0) | (bits << 4)], dtype='i4')
            cfg[p + 'smoothing_radius'] = np.array([pt.smoothing_radius], dtype='f8')
            if resummed:      # damping scales of the resummed wiggles: constants of the (fixed) BAO template, set by ResummedPowerSpectrumWiggles.calculate (bao.py:186-199)
                wig = pt.wiggles
                if not _has(wig, 'sigma_dd2'): wig()      # (reading parameter collections above re-initialises calculators: their cached state is gone until the next calculation)
                cfg[p + 'resummed'] = np.array([wig.sigma_dd2, wig.sigma_nl2, getattr(wig, 'sigma_x2', 0.), wig.shotnoise * wig.sigma_sn2], dtype='f8')
            if flexible:      # multiplicative terms ml{ell}_{i}: kernels K_i(k), the multipole each multiplies, L_ell(mu) (bao.py:337-367)
                from scipy import special
                ml_names, ml_rows, ml_ell = [], [], []
                for ill, ell in enumerate(pt.ells):
                    for name, row in zip(pt.wiggles_orders[ell], np.asarray(pt.wiggles_matrix[ell], dtype='f8').reshape(len(pt.wiggles_orders[ell]), -1)):
                        ml_names.append(names.get(name, name)); ml_rows.append(row); ml_ell.append(ill)
                cfg[p + 'ml_matrix'] = np.array(ml_rows, dtype='f8').reshape(len(ml_names), len(pt.k))
                cfg[p + 'ml_ell'] = np.array(ml_ell, dtype='i4')
                cfg[p + 'legendre'] = np.array([special.legendre(ell)(np.asarray(pt.mu, dtype='f8')) for ell in pt.ells], dtype='f8')
                cfg[p + 'in.ml'] = np.array([column(name, value_of(name, 0.)) for name in ml_names], dtype='f8')
            pass_names = [name for ell in theory.ells for name in theory.broadband_orders[ell]]
            nx = len(theory.s) if xi else len(theory.k)
            pass_matrix = np.zeros((len(theory.ells), nx, len(pass_names)), dtype='f8')
            for ill, ell in enumerate(theory.ells):
                for name, row in zip(theory.broadband_orders[ell], np.asarray(theory.broadband_matrix[ell])):
                    pass_matrix[ill, :, pass_names.index(name)] = row
            pass_matrix = pass_matrix.reshape(-1, len(pass_names))
            if xi and ptheory is not theory and _has(ptheory, 'broadband_orders'):
                # kernel broadbands of a correlation function (bao.py:859-861, 'pcs2' ...): the power spectrum class under the Hankel transform carries the Fourier-space kernels
                # al*, the correlation function class adds the powers of s bl* -- the kernels' columns are their Hankel transforms
                hankel = hankel_operator(theory)
                knames = [name for ell in ptheory.ells for name in ptheory.broadband_orders[ell]]
                kmat = np.zeros((len(theory.ells), nx, len(knames)), dtype='f8')
                for ill, ell in enumerate(ptheory.ells):
                    for name, row in zip(ptheory.broadband_orders[ell], np.asarray(ptheory.broadband_matrix[ell], dtype='f8')):
                        kmat[ill, :, knames.index(name)] = hankel[ill].dot(row)
                pass_names, pass_matrix = knames + pass_names, np.hstack([kmat.reshape(-1, len(knames)), pass_matrix])
            pass_names = [names.get(name, name) for name in pass_names]
        xi_apply = None
        if xi:
            window = _block_diag(list(hankel_operator(theory)))
            if any(getattr(wm, name, None) is not None for name in ['matrix_full', 'matrix_diag', 'smask', 'offset']):
                # binned / masked / fiber-collided correlation function windows (window.py:717-733): ``_apply`` is affine in the theory multipoles [n_ellin, n_sin] -- its
                # linear part, column by column, folds into the operator; its constant part is the offset
                nin = (len(theory.ells), len(theory.s))
                zero = np.asarray(wm._apply(np.zeros(nin, dtype='f8')), dtype='f8')
                xi_apply = lambda flat: np.asarray(wm._apply(np.asarray(flat, dtype='f8').reshape(nin)), dtype='f8') - zero      # noqa: E731
                window = np.column_stack([xi_apply(window[:, j]) for j in range(window.shape[1])])
                if np.any(zero != 0.): cfg[p + 'offset'] = zero
        else:
            window = None if wm.matrix_full is None else np.asarray(wm.matrix_full, dtype='f8')
            if getattr(wm, 'kmask', None) is not None: cfg[p + 'kmask'] = np.asarray(wm.kmask, dtype='i4')
            if getattr(wm, 'offset', None) is not None: cfg[p + 'offset'] = np.asarray(wm.offset, dtype='f8')
            cfg[p + 'shotnoise_in'], cfg[p + 'shotnoise_out'] = np.asarray(wm.shotnoisein, dtype='f8'), np.asarray(wm.shotnoiseout, dtype='f8')
        if pass_names:
            if window is None: window = np.eye(pass_matrix.shape[0])
            if xi and xi_apply is not None: pass_matrix = np.column_stack([xi_apply(pass_matrix[:, j]) for j in range(pass_matrix.shape[1])])   # (the broadband terms go through the window too)
            window = np.hstack([window, window.dot(pass_matrix) if not xi else pass_matrix])
            cfg[p + 'in.pass'] = np.array([column(name, value_of(name, 0.)) for name in pass_names], dtype='f8')
        if window is not None: cfg[p + 'wmatrix'] = window
        defaults = dict(qpar=1., qper=1., qiso=1., qap=1., df=1., dm=0., dn=0., sigmapar=9. if bao else 0., sigmaper=6. if bao else 0.)
        if tns:
            for key in ['sigmapar', 'sigmaper']: defaults.pop(key)
            defaults.update(sigmav=0., b2=0., bs=0., b3=0.)
        if png:
            for key in ['sigmapar', 'sigmaper']: defaults.pop(key)
            defaults.update(fnl_loc=0., sigmas=0.)
            if pngvel: defaults.update(bv=1., sigmau=0.)
        if turnover: defaults.update(m=0.6, n=0.9, qto=1., dpto=1.)
        if bands:
            cfg[p + 'band_templates'] = np.asarray(template.templates, dtype='f8')
            band_names = [names.get('dptt{:d}'.format(i), 'dptt{:d}'.format(i)) for i in range(len(template.templates))]
            cfg[p + 'in.band'] = np.array([column(name, value_of(name, 1.)) for name in band_names], dtype='f8')
        if bao: defaults.update(dbeta=1., sigmas=0.)
        else: defaults.update(sn0=0.)
        if resummed: names.setdefault('dres', names.get('d', 'd')); defaults.update(dres=1.)      # (growth rescaling `d` of the resummed wiggles, bao.py:201: the key's name is dres)
        if xi and not bao: defaults.pop('sn0')                             # no stochastic parameter for correlation functions (full_shape.py:336-364)
        for key, default in defaults.items():
            pname = names.get(key, key)
            cfg[p + 'in.' + key] = column(pname, value_of(pname, default))
        b1 = names.get('b1', 'b1')
        cfg[p + 'in.b1X'] = cfg[p + 'in.b1Y'] = column(b1, value_of(b1, 1.))
        if png:   # auto-spectrum: the X and Y tracer inputs are the same parameters
            cfg[p + 'in.sigmasY'] = cfg[p + 'in.sigmas']
            for key in ['p', 'bphi']:
                pname = names.get(key, key)
                cfg[p + 'in.' + key + 'X'] = cfg[p + 'in.' + key + 'Y'] = column(pname, value_of(pname, 1.))
        if eft:
            cfg[p + 'ct_matrix'], cfg[p + 'sn_matrix'] = np.asarray(ptheory.counterterm_matrix, dtype='f8'), np.asarray(ptheory.stochastic_matrix, dtype='f8')
            ct_names = [names.get(str(n), str(n)) for n in ptheory.counterterm_params]
            sn_names = [names.get(str(n), str(n)) for n in ptheory.stochastic_params]
            cfg[p + 'in.ct'] = np.array([[column(n, value_of(n, 0.))] * 2 for n in ct_names], dtype='f8')      # auto-spectrum: the X and Y tracer inputs of a term are the same parameter
            cfg[p + 'in.sn'] = np.array([column(n, value_of(n, 0.)) for n in sn_names], dtype='f8')
        if solved:
            # which inputs of this observable are solved analytically (they must enter its theory linearly: full_shape.py:545-550, 628-634, bao.py:495-534)
            if p + 'in.sn0' in cfg: cfg[p + 'marg.sn0'] = np.array([sindex(names.get('sn0', 'sn0'))], dtype='i4')
            if pass_names: cfg[p + 'marg.pass'] = np.array([sindex(name) for name in pass_names], dtype='i4')
            if eft:
                cfg[p + 'marg.ct'] = np.array([[sindex(n)] * 2 for n in ct_names], dtype='i4')
                cfg[p + 'marg.sn'] = np.array([sindex(n) for n in sn_names], dtype='i4')
    cfg['__varied__'] = np.array(varied)
    cfg['__solved__'] = np.array(solved)
    return cfg


class Library(object):
    """ctypes view of ``libdesilike_amd.so`` (the symbols of include/desilike_amd.h this binding needs)."""

    def __init__(self, path='libdesilike_amd.so'):
        lib = self.lib = ctypes.CDLL(path)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        lib.dl_config_new.restype = ctypes.c_void_p
        lib.dl_config_set_f64.argtypes = [ctypes.c_void_p, ctypes.c_char_p, dp, ctypes.c_int64]
        lib.dl_config_set_i32.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ip, ctypes.c_int64]
        lib.dl_config_free.argtypes = [ctypes.c_void_p]
        lib.dl_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]
        lib.dl_destroy.argtypes = [ctypes.c_void_p]
        lib.dl_last_error.restype, lib.dl_last_error.argtypes = ctypes.c_char_p, [ctypes.c_void_p]
        lib.dl_eval_batch_host.argtypes = [ctypes.c_void_p, dp, ctypes.c_int64, dp, dp, dp, ip, dp]
        lp = ctypes.POINTER(ctypes.c_int64)
        lib.dl_mh_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ip, ip, ip, ip, ctypes.c_int32, ctypes.c_double, ctypes.c_uint64,
                                     ctypes.c_double, ctypes.c_int64]
        lib.dl_mh_destroy.argtypes = [ctypes.c_void_p]
        lib.dl_mh_set_covariance.argtypes = [ctypes.c_void_p, dp, ctypes.c_void_p]
        lib.dl_mh_set_state.argtypes = [ctypes.c_void_p, dp, dp, lp, lp, ctypes.c_int64, ctypes.c_void_p]
        lib.dl_mh_run_host.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, dp, dp, lp, ip, ctypes.c_void_p]
        lib.dl_mh_get_state.argtypes = [ctypes.c_void_p, dp, dp, lp, lp, ip, ctypes.c_void_p]

    def create(self, cfg, device=0):
        """``dl_create`` from a flat key -> array set; returns the opaque context handle."""
        lib = self.lib
        handle_cfg = lib.dl_config_new()
        keep = []
        try:
            for key, value in cfg.items():
                if key.startswith('__'): continue
                value = np.ascontiguousarray(value)
                if value.dtype.kind in 'iub':
                    value = np.ascontiguousarray(value.ravel(), dtype=np.int32); keep.append(value)
                    rc = lib.dl_config_set_i32(handle_cfg, key.encode(), value.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), value.size)
                else:
                    value = np.ascontiguousarray(value.ravel(), dtype=np.float64); keep.append(value)
                    rc = lib.dl_config_set_f64(handle_cfg, key.encode(), value.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), value.size)
                if rc != 0: raise RuntimeError(lib.dl_last_error(None).decode())
            ctx = ctypes.c_void_p()
            if lib.dl_create(ctypes.byref(ctx), int(device), handle_cfg) != 0:
                raise RuntimeError(lib.dl_last_error(None).decode())
        finally:
            lib.dl_config_free(handle_cfg)
        return ctx

    def eval_batch(self, ctx, theta, n_solved=0):
        """(loglikelihood [B], logprior [B], status [B]) of ``theta [B, P]``: one ``dl_eval_batch_host`` call; ``n_solved`` > 0: also the values of the analytically
        solved parameters ``[B, n_solved]`` (likelihoods/base.py:361-365)."""
        theta = np.ascontiguousarray(np.atleast_2d(theta), dtype='f8')
        B = theta.shape[0]
        loglike, logprior, status = np.empty(B), np.empty(B), np.empty(B, dtype=np.int32)
        solved = np.empty((B, n_solved)) if n_solved else None
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        if self.lib.dl_eval_batch_host(ctx, theta.ctypes.data_as(dp), B, loglike.ctypes.data_as(dp), logprior.ctypes.data_as(dp), None, status.ctypes.data_as(ip),
                                       solved.ctypes.data_as(dp) if n_solved else None) != 0:
            raise RuntimeError(self.lib.dl_last_error(ctx).decode())
        return (loglike, logprior, status, solved) if n_solved else (loglike, logprior, status)


class MetropolisHastings(object):
    """What desilike's ``MHSampler`` (samplers/mcmc.py:25-127) does around ``BlockProposer`` (199-328), through ``dl_mh_*``: ``nchains`` chains x ``vectorize``
    speculative proposals per try as one batch on the GPU.  ``covariance``: proposal covariance of the varied parameters; ``blocks`` / ``oversample_factors`` as
    ``BlockProposer`` takes them (sizes in sorted order, slowest first) with ``order`` = sorted position -> column of the varied parameters."""

    def __init__(self, library, ctx, covariance, nchains=1, vectorize=1, blocks=None, oversample_factors=None, order=None, proposal_scale=2.4, seed=0, max_tries=1000):
        self.library, lib = library, library.lib
        covariance = np.asarray(covariance, dtype='f8')
        self.ndim = ndim = covariance.shape[0]
        ip = ctypes.POINTER(ctypes.c_int32)
        blocks = np.ascontiguousarray([ndim] if blocks is None else blocks, dtype=np.int32)
        over = np.ascontiguousarray(np.ones(len(blocks)) if oversample_factors is None else oversample_factors, dtype=np.int32)
        order = np.ascontiguousarray(np.arange(ndim) if order is None else order, dtype=np.int32)
        self.handle, self.nchains = ctypes.c_void_p(), int(nchains)
        if lib.dl_mh_create(ctypes.byref(self.handle), ctx, self.nchains, int(vectorize), None, order.ctypes.data_as(ip), blocks.ctypes.data_as(ip), over.ctypes.data_as(ip),
                            len(blocks), float(proposal_scale), ctypes.c_uint64(int(seed)), 0., int(max_tries)) != 0:
            raise RuntimeError(lib.dl_last_error(None).decode())
        cholesky = np.ascontiguousarray(np.linalg.cholesky(covariance[np.ix_(order, order)]))                  # BlockProposer.set_covariance, mcmc.py:312
        if lib.dl_mh_set_covariance(self.handle, cholesky.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), None) != 0:
            raise RuntimeError(lib.dl_last_error(None).decode())

    def sample(self, start, iterations=300, thin_by=1):
        """``iterations`` tries of every chain from ``start [nchains, ndim]`` (first call) or from where the chains are; returns per chain
        (coords [n, ndim], weight [n], log_prob [n]): ``get_chain / get_weight / get_log_prob`` of ``MHSampler``."""
        lib, dp, lp = self.library.lib, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
        if start is not None:
            start = np.ascontiguousarray(np.asarray(start, dtype='f8').reshape(self.nchains, self.ndim))
            if lib.dl_mh_set_state(self.handle, start.ctypes.data_as(dp), None, None, None, 0, None) != 0: raise RuntimeError(lib.dl_last_error(None).decode())
        coords, logp = np.empty((self.nchains, iterations, self.ndim)), np.empty((self.nchains, iterations))
        weight, count = np.empty((self.nchains, iterations), dtype=np.int64), np.empty(self.nchains, dtype=np.int32)
        if lib.dl_mh_run_host(self.handle, int(iterations), int(thin_by), coords.ctypes.data_as(dp), logp.ctypes.data_as(dp), weight.ctypes.data_as(lp),
                              count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), None) != 0:
            raise RuntimeError(lib.dl_last_error(None).decode())
        return [(coords[c, :count[c]].copy(), weight[c, :count[c]].copy(), logp[c, :count[c]].copy()) for c in range(self.nchains)]

    def close(self):
        if self.handle: self.library.lib.dl_mh_destroy(self.handle); self.handle = None


def make_calculator():
    """The calculator class, created on demand (it subclasses the reference's ``BaseGaussianLikelihood``: ``desilike`` must be importable)."""
    from desilike.likelihoods.base import BaseGaussianLikelihood

    class MI355XGaussianLikelihood(BaseGaussianLikelihood):
        """Reads the constants off an *initialised* reference likelihood once, then evaluates on the GPU (``BaseCalculator`` contract: desilike/base.py:1119-1323)."""

        def initialize(self, likelihood, device=0, library='libdesilike_amd.so'):
            likelihood()                                           # initialise the reference pipeline once (CPU)
            self.config = extract_config(likelihood)
            self.varied = [str(name) for name in self.config['__varied__']]
            self.flatdata, self.precision = np.asarray(likelihood.flatdata), np.asarray(likelihood.precision)
            self._library = Library(library)
            self._ctx = self._library.create(self.config, device=device)

        def calculate(self, **params):                             # desilike hands Python floats keyed by parameter name
            loglike, logprior, status = self._library.eval_batch(self._ctx, [[params[name] for name in self.varied]])
            self.loglikelihood = float(loglike[0]) if status[0] < 2 else -np.inf

        def evaluate(self, values):
            """``values [B, P]`` -> log-posterior [B] with the samplers' conventions (desilike/samplers/base.py:144-200)."""
            loglike, logprior, status = self._library.eval_batch(self._ctx, values)
            return np.where(status == 0, loglike + logprior, -np.inf)

    return MI355XGaussianLikelihood
