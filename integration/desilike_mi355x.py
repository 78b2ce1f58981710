"""Reference-side binding: what a desilike maintainer adds to run the likelihood node of an EXISTING desilike pipeline on an MI355X.

This module imports ``desilike`` (the reference), never ``desilike_amd``'s Python mirror: it talks to ``libdesilike_amd.so`` through ctypes only
(include/desilike_amd.h).  Two pieces:

* :func:`extract_config` -- walks an *initialised* reference ``ObservablesGaussianLikelihood`` (Kaiser / EFT-like Kaiser tracer theories on
  ShapeFit / Standard / Fixed templates, any window handled by ``WindowedPowerSpectrumMultipoles``, one or several observables, tracer namespaces)
  and returns the flat ``{key: array}`` set of ``dl_config`` (keys documented in the header);
* :class:`MI355XGaussianLikelihood` -- a ``BaseGaussianLikelihood`` whose ``calculate`` is ONE ``dl_eval_batch_host`` call; ``evaluate(values [B, P])`` is the
  batched entry for samplers.

tests/golden/make_boundary_fixture.py runs :func:`extract_config` on real initialised reference likelihoods in the build container and commits the key set
as fixtures; tests/test_gpu_boundary.py creates contexts from those fixtures alone and checks the log-likelihoods the reference computed.
"""
import ctypes

import numpy as np

# enumerations of include/desilike_amd.h
DL_TEMPLATE_FIXED, DL_TEMPLATE_SHAPEFIT = 0, 1
DL_THEORY_KAISER, DL_THEORY_EFT_KAISER = 0, 1
DL_APMODE = {'qparqper': 0, 'qiso': 1, 'qap': 2, 'qisoqap': 3}


def _apmode(template):
    apeffect = getattr(template, 'apeffect', None)
    return DL_APMODE[getattr(apeffect, 'mode', 'qparqper')]


def extract_config(likelihood):
    """Flat ``dl_config`` of an initialised reference likelihood: ``{key: float64 / int32 array}``, plus ``varied`` (theta column order) under ``'__varied__'``."""
    varied = likelihood.varied_params.names()
    cfg = {}

    def column(name, default):
        return np.array([varied.index(name) if name in varied else -1, default], dtype='f8')

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
    for iobs, obs in enumerate(likelihood.observables):
        p = 'obs{:d}.'.format(iobs)
        wm = obs.wmatrix
        theory = wm.theory
        pt, template = theory.pt, theory.pt.template
        eft = hasattr(theory, 'counterterm_matrix')
        shapefit = template.__class__.__name__.startswith('ShapeFit')
        cfg[p + 'theory'] = np.array([DL_THEORY_EFT_KAISER if eft else DL_THEORY_KAISER], dtype='i4')
        cfg[p + 'template'] = np.array([DL_TEMPLATE_SHAPEFIT if shapefit else DL_TEMPLATE_FIXED], dtype='i4')
        cfg[p + 'apmode'] = np.array([_apmode(template)], dtype='i4')
        cfg[p + 'transform'] = np.array([1 if getattr(obs, 'transform', None) == 'cubic' else 0], dtype='i4')
        cfg[p + 'eta'] = np.array([getattr(getattr(template, 'apeffect', None), 'eta', 1. / 3.)], dtype='f8')
        cfg[p + 'f_fid'] = np.array([template.f_fid], dtype='f8')
        cfg[p + 'a'] = np.array([getattr(template, 'a', 0.6)], dtype='f8')
        cfg[p + 'kp'] = np.array([getattr(template, 'kp', 0.03)], dtype='f8')
        cfg[p + 'nd'] = np.array([theory.nd], dtype='f8')
        cfg[p + 'ells_in'] = np.asarray(wm.ellsin, dtype='i4')
        cfg[p + 'kin'] = np.asarray(pt.k, dtype='f8')
        cfg[p + 'mu'], cfg[p + 'wmu_ell'] = np.asarray(pt.mu, dtype='f8'), np.asarray(pt.wmu, dtype='f8')
        cfg[p + 'k_t'], cfg[p + 'pk_dd_fid'] = np.asarray(template.k, dtype='f8'), np.asarray(template.pk_dd_fid, dtype='f8')
        if wm.matrix_full is not None: cfg[p + 'wmatrix'] = np.asarray(wm.matrix_full, dtype='f8')
        if getattr(wm, 'kmask', None) is not None: cfg[p + 'kmask'] = np.asarray(wm.kmask, dtype='i4')
        if getattr(wm, 'offset', None) is not None: cfg[p + 'offset'] = np.asarray(wm.offset, dtype='f8')
        cfg[p + 'shotnoise_in'], cfg[p + 'shotnoise_out'] = np.asarray(wm.shotnoisein, dtype='f8'), np.asarray(wm.shotnoiseout, dtype='f8')
        cfg[p + 'flatdata'] = np.asarray(obs.flatdata, dtype='f8')
        # parameter -> theta column (or constant): tracer namespaces prefix the bias / shot-noise parameters (full_shape.py:88-128)
        names = {param.basename: param.name for param in theory.all_params}
        names.update({param.basename: param.name for param in pt.all_params})
        defaults = dict(qpar=1., qper=1., qiso=1., qap=1., df=1., dm=0., dn=0., sigmapar=0., sigmaper=0., sn0=0.)
        for key, default in defaults.items():
            pname = names.get(key, key)
            value = likelihood.all_params[pname].value if pname in likelihood.all_params else default
            cfg[p + 'in.' + key] = column(pname, value)
        b1 = names.get('b1', 'b1')
        value = likelihood.all_params[b1].value if b1 in likelihood.all_params else 1.
        cfg[p + 'in.b1X'] = cfg[p + 'in.b1Y'] = column(b1, value)
        if eft:
            cfg[p + 'ct_matrix'], cfg[p + 'sn_matrix'] = np.asarray(theory.counterterm_matrix, dtype='f8'), np.asarray(theory.stochastic_matrix, dtype='f8')
            ct = [column(names.get(str(n), str(n)), likelihood.all_params[names.get(str(n), str(n))].value) for n in theory.counterterm_params]
            sn = [column(names.get(str(n), str(n)), likelihood.all_params[names.get(str(n), str(n))].value) for n in theory.stochastic_params]
            cfg[p + 'in.ct'] = np.array([[c, c] for c in ct], dtype='f8')      # auto-spectrum: the X and Y tracer inputs of a term are the same parameter
            cfg[p + 'in.sn'] = np.array(sn, dtype='f8')
    cfg['__varied__'] = np.array(varied)
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

    def eval_batch(self, ctx, theta):
        """(loglikelihood [B], logprior [B], status [B]) of ``theta [B, P]``: one ``dl_eval_batch_host`` call."""
        theta = np.ascontiguousarray(np.atleast_2d(theta), dtype='f8')
        B = theta.shape[0]
        loglike, logprior, status = np.empty(B), np.empty(B), np.empty(B, dtype=np.int32)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        if self.lib.dl_eval_batch_host(ctx, theta.ctypes.data_as(dp), B, loglike.ctypes.data_as(dp), logprior.ctypes.data_as(dp), None, status.ctypes.data_as(ip), None) != 0:
            raise RuntimeError(self.lib.dl_last_error(ctx).decode())
        return loglike, logprior, status


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
