"""Linear power spectrum templates (reference: desilike/theories/galaxy_clustering/power_template.py).

Host-side mirrors: they hold the fiducial tables and the parameter definitions (names, defaults, priors as in
the reference's ``power_template.yaml``); the per-point arithmetic -- AP rescaling
(theories/galaxy_clustering/base.py:211-223, 341-353), ShapeFit factor (power_template.py:747-761),
``f = f_fid * df`` (592-596, 372-376) -- runs in the HIP theory kernel (csrc/dl_fullshape.h, phases 0-1).
"""
import numpy as np

from ...base import BaseCalculator
from ...fiducial import get_fiducial

_AP = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.99, 1.01]), delta=0.008, latex=r'q_{\parallel}'),
       'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.99, 1.01]), delta=0.008, latex=r'q_{\perp}'),
       'qiso': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.99, 1.01]), delta=0.008, latex=r'q_{\mathrm{iso}}'),
       'qap': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.99, 1.01]), delta=0.008, latex=r'q_{\mathrm{ap}}')}
_DF = {'df': dict(value=1., prior=dict(limits=[0., 2.]), ref=dict(limits=[0.95, 1.05]), delta=0.02, latex='df')}
_APMODES = {'qparqper': (0, ['qpar', 'qper']), 'qiso': (1, ['qiso']), 'qap': (2, ['qap']), 'qisoqap': (3, ['qiso', 'qap'])}


class BasePowerSpectrumTemplate(BaseCalculator):
    """Base template: fiducial ``pk_dd_fid`` on ``k`` and AP parameters (power_template.py:69-172).

    Parameters
    ----------
    k : array, default=None
        Wavenumbers of the template table; set by the theory (full_shape.py:29) when left to ``None``.
    z : float, default=1.
        Effective redshift (bookkeeping only: the fiducial provider is already evaluated at z).
    apmode : str, default='qparqper'
        'qiso', 'qap', 'qisoqap', 'qparqper' (theories/galaxy_clustering/base.py:293-300).
        'geometry' / 'bao' need a Boltzmann code and are out of scope.
    fiducial : TabulatedFiducial, SyntheticFiducial, dict, default='DESI'
        Provider of ``pk_dd(k)``, ``pknow_dd(k)``, ``f`` (see :mod:`desilike_amd.fiducial`).
    """
    _kind = 0  # DL_TEMPLATE_FIXED
    _own_params = {}

    @classmethod
    def _default_params(cls, apmode='qparqper', **kwargs):
        import copy
        if apmode not in _APMODES:
            raise ValueError('unknown mode {}; it must be one of {} (geometry / bao need a Boltzmann code: out of scope)'.format(apmode, list(_APMODES)))
        params = {name: copy.deepcopy(_AP[name]) for name in _APMODES[apmode][1]}
        params.update(copy.deepcopy(cls._own_params))
        return params

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        self.z = float(init.get('z', 1.))
        self.apmode = init.get('apmode', 'qparqper')
        self.eta = float(init.get('eta', 1. / 3.))
        self.fiducial = get_fiducial(init.get('fiducial', 'DESI'))
        k = init.get('k', None)
        if k is None: k = np.logspace(-3., 1., 400)
        self.k = np.array(k, dtype='f8')
        self.with_now = init.get('with_now', self._with_now_default)
        self.only_now = bool(init.get('only_now', False))
        self.pk_dd_fid = np.asarray(self.fiducial.pk_dd(self.k), dtype='f8')
        if self.with_now or self.only_now:
            self.pknow_dd_fid = np.asarray(self.fiducial.pknow_dd(self.k), dtype='f8')
        self.f_fid = float(self.fiducial.f)
        self._initialized = True
        return self

    _with_now_default = False

    def _template_spec(self):
        """Template part of an observable spec (C-ABI keys obs<i>.template, .apmode, .eta, .k_t, .pk_dd_fid, .f_fid, .a, .kp)."""
        self.initialize()
        pk = self.pknow_dd_fid if self.only_now else self.pk_dd_fid  # power_template.py:118-120
        return dict(template=np.array([self._kind], dtype='i4'), apmode=np.array([_APMODES[self.apmode][0]], dtype='i4'), eta=[self.eta],
                    k_t=self.k, pk_dd_fid=pk, f_fid=[self.f_fid], a=[getattr(self, 'a', 0.6)], kp=[getattr(self, 'kp', 0.03)])

    # names of the kernel inputs this template feeds, mapped to parameter basenames
    _input_names = ['qpar', 'qper', 'qiso', 'qap', 'df', 'dm', 'dn']
    _extra_inputs = {}      # kernel input -> parameter basename, for templates with inputs of their own


class FixedPowerSpectrumTemplate(BasePowerSpectrumTemplate):
    """Fixed template: no varied parameter, qpar = qper = 1 (power_template.py:176-202)."""

    @classmethod
    def _default_params(cls, **kwargs):
        return {}


class StandardPowerSpectrumTemplate(BasePowerSpectrumTemplate):
    """Standard template in terms of ``df`` and AP parameters (power_template.py:553-599)."""
    _own_params = _DF


class BAOPowerSpectrumTemplate(BasePowerSpectrumTemplate):
    """BAO template: AP parameters (and ``df`` when passed explicitly), no-wiggle table available (power_template.py:339-389)."""
    _with_now_default = 'peakaverage'


class ShapeFitPowerSpectrumTemplate(BasePowerSpectrumTemplate):
    """ShapeFit template (power_template.py:696-764): ``pk_dd = pk_dd_fid exp(dm / a tanh(a ln(k / kp)) + dn ln(k / kp))``, ``f = f_fid df``."""
    _kind = 1  # DL_TEMPLATE_SHAPEFIT
    _with_now_default = 'peakaverage'
    _own_params = {'dm': dict(value=0., prior=dict(limits=[-3., 3.]), ref=dict(limits=[-0.01, 0.01]), delta=0.01, latex='dm'),
                   'dn': dict(fixed=True, prior=dict(limits=[-0.5, 0.5]), ref=dict(dist='norm', loc=0., scale=0.1), latex='dn'),
                   **_DF}

    @classmethod
    def _default_params(cls, apmode='qparqper', **kwargs):
        # parameter order of the reference's power_template.yaml:279-338: dm, dn, AP parameters, df
        params = super(ShapeFitPowerSpectrumTemplate, cls)._default_params(apmode=apmode, **kwargs)
        order = ['dm', 'dn'] + _APMODES[apmode][1] + ['df']
        return {name: params[name] for name in order}

    def initialize(self):
        if self._initialized:
            return self
        self.a = float(self.init.get('a', 0.6))
        self.kp = float(self.init.get('kp', 0.03))
        return super(ShapeFitPowerSpectrumTemplate, self).initialize()


def find_turn_over(k, pk):
    """Turn-over of a tabulated spectrum: vertex of the parabola through the three points around the maximum in (log10 k, log10 P) (the reference's estimate,
    power_template.py:1205-1220).  Returns the wavenumber; the power there is read off the spectrum itself."""
    k, pk = np.asarray(k, dtype='f8'), np.asarray(pk, dtype='f8')
    imax = int(np.argmax(pk))
    if imax == 0 or imax == k.size - 1: raise ValueError('the maximum of the spectrum sits at the end of its table')
    x, y = np.log10(k[imax - 1:imax + 2]), np.log10(pk[imax - 1:imax + 2])
    # Lagrange form of the parabola: y(t) = sum_i y_i prod_{j != i} (t - x_j) / (x_i - x_j); its vertex
    w = np.array([y[0] / ((x[0] - x[1]) * (x[0] - x[2])), y[1] / ((x[1] - x[0]) * (x[1] - x[2])), y[2] / ((x[2] - x[0]) * (x[2] - x[1]))])
    curvature = w.sum()
    if not curvature < 0.: raise ValueError('no maximum between the three highest points')
    vertex = (w[0] * (x[1] + x[2]) + w[1] * (x[0] + x[2]) + w[2] * (x[0] + x[1])) / (2. * curvature)
    return 10.**vertex


class TurnOverPowerSpectrumTemplate(BasePowerSpectrumTemplate):
    r"""Turn-over template (power_template.py:1293-1340; arXiv:2302.07484): the spectrum around its maximum as two half-parabolas in log-log,

    .. math:: P(k) = P_{TO}^{1 - m x^2} \; (x > 0), \quad P_{TO}^{1 - n x^2} \; (x \le 0), \qquad x = \log_{10} k / \log_{10} k_{TO} - 1,

    with :math:`k_{TO} = q_{TO} k_{TO}^{fid}`, :math:`P_{TO} = dp_{TO} P_{TO}^{fid}`, ``f = f_fid df`` and the single Alcock-Paczynski parameter ``qap``.  The fiducial
    turn-over is found on the fiducial spectrum (:func:`find_turn_over`), or given: ``kTO_fid``, ``pkTO_dd_fid``."""
    _kind = 2  # DL_TEMPLATE_TURNOVER
    _own_params = {'m': dict(value=0.57, prior=dict(limits=[-1., 10.]), ref=dict(limits=[0., 1.]), delta=0.01, latex='m'),
                   'n': dict(value=0.89, prior=dict(limits=[0., 10.]), ref=dict(limits=[0.5, 1.]), delta=0.01, latex='n'),
                   'dpto': dict(value=1., fixed=True, prior=dict(limits=[0., 2.]), ref=dict(limits=[0.9, 1.1]), delta=0.01, latex=r'(P / P^{\mathrm{fid}})(k_{\mathrm{TO}})'),
                   'qto': dict(value=1., prior=dict(limits=[0.5, 1.5]), ref=dict(limits=[0.99, 1.01]), delta=0.008, latex=r'q_{\mathrm{TO}}')}
    _extra_inputs = {'m': 'm', 'n': 'n', 'qto': 'qto', 'dpto': 'dpto'}

    @classmethod
    def _default_params(cls, **kwargs):
        # power_template.yaml:424-477: m, n, dpto, qto, qap (fixed), df (fixed)
        import copy
        params = copy.deepcopy(cls._own_params)
        params['qap'] = dict(copy.deepcopy(_AP['qap']), fixed=True)
        params['df'] = dict(copy.deepcopy(_DF['df']), fixed=True)
        return params

    def initialize(self):
        if self._initialized:
            return self
        self.init['apmode'] = 'qap'                      # power_template.py:1324
        super(TurnOverPowerSpectrumTemplate, self).initialize()
        kTO, pkTO = self.init.get('kTO_fid', None), self.init.get('pkTO_dd_fid', None)
        if kTO is None:
            grid = np.geomspace(1e-4, 10., 1201)
            kTO = find_turn_over(grid, self.fiducial.pk_dd(grid))
        if pkTO is None: pkTO = float(np.ravel(self.fiducial.pk_dd(np.array([kTO])))[0])
        self.kTO_fid, self.pkTO_dd_fid = float(kTO), float(pkTO)
        return self

    def _template_spec(self):
        spec = super(TurnOverPowerSpectrumTemplate, self)._template_spec()
        spec.update(kto_fid=[self.kTO_fid], pkto_fid=[self.pkTO_dd_fid])
        return spec


class BandVelocityPowerSpectrumTemplate(BasePowerSpectrumTemplate):
    r"""Velocity-divergence power spectrum in bands (power_template.py:868-970): around the pivots ``kp`` the fiducial :math:`P_{\theta\theta}` is modulated by tent
    functions with amplitudes ``dptt0``, ``dptt1``, ... (1 = fiducial), :math:`P_{\theta\theta} = P^{fid}_{\theta\theta} [1 + \sum_i (dptt_i - 1) T_i(k)]`,
    :math:`P_{dd} = P_{\theta\theta} / (f_{fid} df)^2`; single Alcock-Paczynski parameter ``qap``.

    ``kp``: the pivots, or (first, last) with the number of bands given by ``nbands`` (default: the ``dptt*`` parameters passed in ``params``; the reference counts
    them in its parameter file).  The fiducial provider gives :math:`P_{dd}`: the fiducial :math:`P_{\theta\theta}` is ``f_fid^2 P_dd`` (linear theory), or pass
    ``pk_tt_fid`` on ``k``."""
    _kind = 3  # DL_TEMPLATE_BANDS
    _base_param_name = 'dptt'

    @classmethod
    def _default_params(cls, nbands=None, kp=None, **kwargs):
        import copy
        if nbands is None: nbands = len(kp) if kp is not None and len(np.atleast_1d(kp)) != 2 else 0
        params = {'qap': copy.deepcopy(_AP['qap']), 'df': dict(copy.deepcopy(_DF['df']), fixed=True)}
        for i in range(int(nbands)):
            params['{}{:d}'.format(cls._base_param_name, i)] = dict(value=1., prior=dict(limits=[0., 3.]), ref=dict(dist='norm', loc=1., scale=0.01), delta=0.005)
        return params

    def initialize(self):
        if self._initialized:
            return self
        self.init['apmode'] = 'qap'                      # power_template.py:894
        super(BandVelocityPowerSpectrumTemplate, self).initialize()
        names = sorted((param.basename for param in self.params if param.basename.startswith(self._base_param_name) and param.basename[len(self._base_param_name):].isdigit()),
                       key=lambda name: int(name[len(self._base_param_name):]))
        nkp = len(names)
        kp = self.init.get('kp', None)
        if kp is None:
            if not nkp: raise ValueError('No parameter {}* found'.format(self._base_param_name))
            step = (self.k[-1] - self.k[0]) / nkp
            kp = (self.k[0] + step / 2., self.k[-1] - step / 2.)
        kp = np.array(kp, dtype='f8')
        if nkp and kp.size == 2: kp = np.linspace(kp[0], kp[1], nkp)
        if kp.size != nkp: raise ValueError('{:d} (!= {:d} parameters {}*) points have been provided'.format(kp.size, nkp, self._base_param_name))
        if names != ['{}{:d}'.format(self._base_param_name, i) for i in range(nkp)]: raise ValueError('Found parameters {}, expected {}0 .. {}{:d}'.format(names, self._base_param_name, self._base_param_name, nkp - 1))
        if kp[0] < self.k[0] or kp[-1] > self.k[-1]: raise ValueError('the pivots must lie inside the theory wavenumbers')
        self.kp, self._band_names = kp, names
        # tent functions (power_template.py:931-939)
        edges = np.concatenate([[self.k[0]], kp, [self.k[-1]]])
        tents = []
        for ip, pivot in enumerate(kp):
            distance = self.k - pivot
            tents.append(np.maximum(1. - np.where(distance < 0., distance / (edges[ip] - pivot), distance / (edges[ip + 2] - pivot)), 0.))
        self.templates = np.array(tents)
        pk_tt_fid = self.init.get('pk_tt_fid', None)
        self.pk_tt_fid = self.f_fid**2 * self.pk_dd_fid if pk_tt_fid is None else np.asarray(pk_tt_fid, dtype='f8')
        self.pk_dd_fid = self.pk_tt_fid / self.f_fid**2     # power_template.py:945
        return self

    @property
    def _extra_inputs(self):
        self.initialize()
        return {'band': list(self._band_names)}

    def _template_spec(self):
        spec = super(BandVelocityPowerSpectrumTemplate, self)._template_spec()
        spec.update(band_templates=self.templates)
        return spec
