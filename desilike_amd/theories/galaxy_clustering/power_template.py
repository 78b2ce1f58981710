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
