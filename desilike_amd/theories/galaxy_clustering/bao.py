"""BAO wiggle models with broadband terms (reference: desilike/theories/galaxy_clustering/bao.py).

Host-side mirrors of ``DampedBAOWigglesTracerPowerSpectrumMultipoles`` (422-560) and
``DampedBAOWigglesTracerCorrelationFunctionMultipoles`` (790-960) for the default wiggle model 'standard' (117-140), reconstruction modes
'', 'recsym', 'reciso', and the 'power' broadband (495-534, 881-905).  Per point the GPU evaluates the wiggle multipoles P_ell(k_in)
(``dl_bao_kernel``); everything downstream is linear and constant, so it is folded into the window matrix once at compile time:
the broadband matrices, and for the correlation function the whole P_ell -> xi_ell chain of ``get_corr``
(theories/galaxy_clustering/base.py:127-136: log-k interpolation, damped tail, FFTLog, interpolation to s) as one Hankel operator
(:func:`desilike_amd.fftlog.hankel_operator`).  Broadband parameterisations: powers of k / s ('power', 'power3', 'even-power') and sums of
mass-assignment-like kernels in Fourier space ('ngp', 'cic', 'tsc', 'pcs'; 'pcs2' for the correlation function), bao.py:468-523, 833-905 -- all
constant matrices.  Wiggle models: 'standard' and the 'fix-damping' / 'move-all' / 'fog-damping' family of
``DampedBAOWigglesPowerSpectrumMultipoles`` (bao.py:117-151); the Resummed / Flexible classes are not implemented.
"""
import re

import numpy as np

from ...base import BaseCalculator
from ...parameter import ParameterCollection
from ... import utils
from .power_template import BAOPowerSpectrumTemplate

_BAO_PARAMS = {'b1': dict(prior=dict(limits=[0.2, 4.]), ref=dict(limits=[1.5, 2.5]), proposal=0.1, latex='b'),
               'dbeta': dict(value=1., prior=dict(limits=[0.7, 1.3]), ref=dict(limits=[0.8, 1.2]), delta=0.02, proposal=0.1, latex=r'd\beta'),
               'sigmas': dict(value=0., prior=dict(limits=[0., 10.]), ref=dict(limits=[0., 1.]), latex=r'\Sigma_{s}'),
               'sigmapar': dict(value=9., prior=dict(limits=[0.1, 10.]), latex=r'\Sigma_{\parallel}', fixed=True),
               'sigmaper': dict(value=6., prior=dict(limits=[0.1, 10.]), latex=r'\Sigma_{\perp}', fixed=True)}


_KERNELS = ('ngp', 'cic', 'tsc', 'pcs')


def _kernel_func(x, kernel='tsc'):
    """B-spline kernels of order 0-3 at distance ``x >= 0`` from the node, in units of the node spacing (bao.py:43-60)."""
    x = np.asarray(x, dtype='f8')
    if kernel == 'ngp':
        return np.where(x < 0.5, 1., 0.)
    if kernel == 'cic':
        return np.where(x < 1., 1. - x, 0.)
    if kernel == 'tsc':
        return np.where(x < 0.5, 0.75 - x**2, np.where(x < 1.5, 0.5 * (1.5 - x)**2, 0.))
    if kernel == 'pcs':
        return np.where(x < 1., (4. - 6. * x**2 + 3. * x**3) / 6., np.where(x < 2., (2. - x)**3 / 6., 0.))
    raise ValueError('Unknown kernel: {}'.format(kernel))


def _get_orders(base, params, ells):
    """{ell: {parameter name: index}} of the parameters '<base><ell>_<index>'; parameters of multipoles not in use are dropped (bao.py:24-41)."""
    orders = {ell: {} for ell in ells}
    for param in list(params):
        match = re.match(base + '(.*)_(.*)$', param.basename)
        if match:
            ell, index = int(match.group(1)), int(match.group(2))
            if ell in orders: orders[ell][param.name] = index
            else: del params[param.name]
    return orders


class _BaseDampedBAOTracer(BaseCalculator):
    _kind = 2   # DL_THEORY_BAO_DAMPED
    _klim = (1e-4, 1., 2000)   # template knots, bao.py:67
    _powers = ()
    _ref_limits = (-1e2, 1e2)

    _resummed = False
    _flexible = False
    _wants_shotnoise = False
    _default_model = 'standard'

    @classmethod
    def _default_params(cls, broadband='power', **kwargs):
        """bao.py:462-481 (power spectrum) / 833-853 (correlation function)."""
        import copy
        broadband = str(broadband)
        params = copy.deepcopy(_BAO_PARAMS)
        if cls._flexible:   # bao.yaml: b1, dbeta (fixed: degenerate) + multiplicative wiggle terms ml{ell}_{i} (bao.py:310-322)
            params = {name: params[name] for name in ['b1', 'dbeta']}
            params['dbeta'].update(fixed=True, ref=dict(limits=[0.95, 1.05]))
            wiggles = str(kwargs.get('wiggles', 'pcs'))
            for ell in (0, 2, 4):
                if wiggles == 'power':
                    for pow in range(-3, 2):
                        params['ml{:d}_{:d}'.format(ell, pow)] = dict(value=0., ref=dict(limits=[-1e2, 1e2]), delta=0.005, latex='a_{{{:d}, {:d}}}'.format(ell, pow))
                else:
                    for ik in range(-2, 10):
                        params['ml{:d}_{:d}'.format(ell, ik)] = dict(value=0., prior=dict(dist='norm', loc=0., scale=1e4), ref=dict(limits=[-1e-2, 1e-2]), delta=0.005,
                                                                       latex='a_{{{:d}, {:d}}}'.format(ell, ik))
        if cls._resummed:   # bao.yaml: b1, dbeta, sigmas, d
            params = {name: params[name] for name in ['b1', 'dbeta', 'sigmas']}
            params['dbeta'].update(ref=dict(limits=[0.95, 1.05]))
            if cls._space == 'pk':   # (the correlation function class has no 'd' parameter in bao.yaml: d = 1)
                params['d'] = dict(value=1., fixed=True, prior=dict(limits=[0., 4.]), ref=dict(limits=[0.8, 1.2]), latex='d')
        if 'power' in broadband:
            for ell in (0, 2, 4):
                for pow in cls._powers:
                    param = dict(value=0., ref=dict(limits=list(cls._ref_limits)), delta=0.005, latex='a_{{{:d}, {:d}}}'.format(ell, pow))
                    if broadband == 'power3' and pow not in (-2, -1, 0): param.update(fixed=True)
                    if cls._space == 'xi' and broadband == 'even-power' and pow not in (0, 2): param.update(fixed=True)
                    params['al{:d}_{:d}'.format(ell, pow)] = param
        elif broadband[:3] in _KERNELS:
            for ell in (0, 2, 4):
                if cls._space == 'pk':
                    for ik in range(-2, 10):   # loose prior "just to regularize the fit"
                        params['al{:d}_{:d}'.format(ell, ik)] = dict(value=0., prior=dict(dist='norm', loc=0., scale=1e4), ref=dict(limits=[-1e-2, 1e-2]), delta=0.005,
                                                                       latex='a_{{{:d}, {:d}}}'.format(ell, ik))
                else:
                    for ik in range(-2, 3):    # infinite prior
                        param = dict(value=0., prior=None, ref=dict(limits=[-1e2, 1e2]), delta=0.005, latex='a_{{{:d}, {:d}}}'.format(ell, ik))
                        if broadband == 'pcs2' and (ell == 0 or ik not in (0, 1)): param.update(fixed=True)
                        params['al{:d}_{:d}'.format(ell, ik)] = param
            if cls._space == 'xi':   # the powers of s come after the Fourier-space kernels (they belong to the correlation function calculator, bao.py:885-890)
                for ell in (0, 2, 4):
                    for ik in (0, 2):
                        params['bl{:d}_{:d}'.format(ell, ik)] = dict(value=0., ref=dict(limits=[-1e-3, 1e-3]), delta=0.005, latex='b_{{{:d}, {:d}}}'.format(ell, ik))
        else:
            raise ValueError('Unknown kernel: {}'.format(broadband))
        return params

    def _init_wiggles(self, k):
        init = self.init
        self.ells = tuple(init.get('ells', (0, 2)))
        self.mode = str(init.get('mode', ''))
        if self.mode not in ['', 'recsym', 'reciso']:
            raise ValueError('Reconstruction mode {} must be one of {}'.format(self.mode, ['', 'recsym', 'reciso']))
        self.model = str(init.get('model', self._default_model))   # 'standard' (bao.py:123-136) or any combination of 'fix-damping', 'move-all', 'fog-damping' (137-150)
        self._model_bits = 0
        if self._flexible:
            self._model_bits = 32 | (2 if 'move-all' in self.model else 0)
        elif self._resummed:
            self._model_bits = 16 | (2 if 'move-all' in self.model else 0) | (4 if 'fog-damping' in self.model else 0)
        elif self.model != 'standard':
            self._model_bits = 8 | (1 if 'fix-damping' in self.model else 0) | (2 if 'move-all' in self.model else 0) | (4 if 'fog-damping' in self.model else 0)
        self.smoothing_radius = float(init.get('smoothing_radius', 15.))
        self.kin = np.array(k, dtype='f8')
        self.mu, wmu = utils.weights_mu(init.get('mu', 10), method='leggauss')   # bao.py:109
        self.wmu = utils.multipole_weights(self.mu, wmu, self.ells)
        template = init.get('template', None)
        if template is None:
            template = self.init['template'] = BAOPowerSpectrumTemplate()
        self.template = self._require(template)
        tk = template.init.get('k', None)
        knots = np.geomspace(min(self._klim[0], self.kin[0] / 2, tk[0] if tk is not None else 1.), max(self._klim[1], self.kin[-1] * 2, tk[0] if tk is not None else 0.), self._klim[2])
        template.init.update(k=knots, with_now=template.init.get('with_now', None) or 'peakaverage')
        if init.get('z', None) is not None: template.init.update(z=init['z'])
        template.initialize()
        self.z = template.z
        if template.apmode != 'qparqper':
            pass
        if self._resummed:
            self._set_resummation(float(init.get('shotnoise', 0.)))
        if self._flexible:
            self._set_flexible_wiggles()
        # broadband orders (bao.py:24-41): parameters al{ell}_{pow} of the multipoles in use; others are dropped
        self.broadband = str(init.get('broadband', 'power'))
        self.broadband_orders = _get_orders('al', self.init.params, self.ells)

    def _set_resummation(self, shotnoise):
        """Damping scales of the resummed wiggles (ResummedPowerSpectrumWiggles.calculate, bao.py:186-199): constants of the (fixed) BAO template."""
        from scipy import special, integrate
        template = self.template
        k, pklin = template.k, template.pknow_dd_fid
        j0 = special.jn(0, template.fiducial.rs_drag * k)
        sk = np.exp(-0.5 * (k * self.smoothing_radius)**2) if self.mode else 0.
        skc = 1. - sk
        self.shotnoise = shotnoise
        self.sigma_sn2 = 1. / self.smoothing_radius / 6. / np.pi**1.5
        self.sigma_nl2 = 1. / (3. * np.pi**2) * integrate.simpson((1. - j0) * pklin, x=k)
        self.sigma_dd2 = 1. / (3. * np.pi**2) * integrate.simpson((1. - j0) * skc**2 * pklin, x=k)
        self.sigma_x2 = 1. / (3. * np.pi**2) * integrate.simpson((1. - j0) * skc * pklin, x=k) if self.mode == 'reciso' else 0.

    def _set_flexible_wiggles(self):
        """Kernels of the multiplicative wiggle terms (FlexibleBAOWigglesPowerSpectrumMultipoles.set_params, bao.py:337-358): K_i(k) = (k / kp)^pow or a B-spline node;
        nodes whose kernel vanishes on the theory wavenumbers are dropped with their parameters."""
        from scipy import special
        init = self.init
        self.wiggles = str(init.get('wiggles', 'pcs'))
        kp = init.get('kp', None)
        self.kp = 2. * np.pi / self.template.fiducial.rs_drag if kp is None else float(kp)
        orders = _get_orders('ml', self.init.params, self.ells)
        names, ells, rows = [], [], []
        for ill, ell in enumerate(self.ells):
            for name, index in orders[ell].items():
                if self.wiggles == 'power':
                    kernel = (self.kin / self.kp)**index
                elif self.wiggles in _KERNELS:
                    kernel = _kernel_func(np.abs(self.kin / self.kp - index), kernel=self.wiggles)
                    if np.allclose(kernel, 0., rtol=0., atol=1e-8):
                        del self.init.params[name]
                        continue
                else:
                    raise ValueError('Unknown kernel: {}'.format(self.wiggles))
                names.append(name); ells.append(ill); rows.append(kernel)
        self.wiggles_params, self.wiggles_ells = names, np.array(ells, dtype='i4')
        self.wiggles_matrix = np.array(rows, dtype='f8').reshape(len(names), len(self.kin))
        self.legendre = np.array([special.eval_legendre(ell, self.mu) for ell in self.ells], dtype='f8')

    def _pknow_fid(self, k):
        """No-wiggle fiducial power at ``k``: cubic interpolation in log10 k on the template knots (``_interp(template, 'pknow_dd_fid', k)``, bao.py:18-19)."""
        from scipy import interpolate
        return interpolate.interp1d(np.log10(self.template.k), self.template.pknow_dd_fid, kind='cubic', fill_value='extrapolate', assume_sorted=True)(np.log10(k))

    def _kernel_broadband_matrix(self, k, kp, kernel):
        """[n_ell * len(k), n_bb]: kernel(|k / kp - ik|) scaled by the no-wiggle power at the node (bao.py:505-516); nodes whose kernel vanishes on ``k`` are dropped,
        and so are their parameters."""
        columns, names = [], []
        for ill, ell in enumerate(self.ells):
            kept = {}
            for name, ik in self.broadband_orders[ell].items():
                kern = _kernel_func(np.abs(k / kp - ik), kernel=kernel)
                if not np.allclose(kern, 0., rtol=0., atol=1e-8):
                    column = np.zeros((len(self.ells), len(k)), dtype='f8')
                    column[ill] = kern * self._pknow_fid(np.clip(ik * kp, k[0], k[-1]))
                    columns.append(column.ravel()); names.append(name)
                    kept[name] = ik
                else:
                    del self.init.params[name]
            self.broadband_orders[ell] = kept
        matrix = np.array(columns, dtype='f8').T if columns else np.zeros((len(self.ells) * len(k), 0), dtype='f8')
        return names, matrix

    def _broadband_matrix(self, x, xp):
        """[n_ell * len(x), n_bb]: columns (x / xp)^pow of each multipole's broadband parameters (bao.py:497-499)."""
        names = [name for ell in self.ells for name in self.broadband_orders[ell]]
        matrix = np.zeros((len(self.ells), len(x), len(names)), dtype='f8')
        for ill, ell in enumerate(self.ells):
            for name, pow in self.broadband_orders[ell].items():
                matrix[ill, :, names.index(name)] = (x / xp)**pow
        return names, matrix.reshape(-1, len(names))

    def _theory_spec(self):
        self.initialize()
        template = self.template
        spec = dict(theory=np.array([self._kind], dtype='i4'), nd=[1.], ells_in=np.array(self.ells, dtype='i4'), kin=self.kin, mu=self.mu, wmu_ell=self.wmu,
                    bao_mode=np.array([(1 if self.mode == 'reciso' else 0) | (self._model_bits << 4)], dtype='i4'), smoothing_radius=[self.smoothing_radius], pknow_dd_fid=template.pknow_dd_fid)
        spec.update(template._template_spec())
        spec['template'] = np.array([0], dtype='i4')   # the BAO template never changes P(k) (power_template.py:372-376)
        if self._resummed:
            spec['resummed'] = np.array([self.sigma_dd2, self.sigma_nl2, self.sigma_x2, self.shotnoise * self.sigma_sn2], dtype='f8')
        if self._flexible:
            spec.update(ml_matrix=self.wiggles_matrix, ml_ell=self.wiggles_ells, legendre=self.legendre)
        return spec

    def _input_map(self):
        toret = {name: name for name in ['qpar', 'qper', 'qiso', 'qap', 'df', 'dbeta', 'sigmas', 'sigmapar', 'sigmaper']}
        toret['b1X'] = toret['b1Y'] = 'b1'
        if self._resummed: toret['dres'] = 'd'
        if self._flexible: toret['ml'] = list(self.wiggles_params)
        toret['pass'] = list(self._broadband_names)
        return toret

    def _all_params(self):
        self.initialize()
        ap = [param for param in self.template.params if param.basename in ('qpar', 'qper', 'qiso', 'qap')]
        others = [param for param in self.template.params if param not in ap]
        return ParameterCollection(ap + others) + self.params

    def _param_collections(self):
        return [self.template.init.params, self.init.params]


    @property
    def _standalone_space(self):
        return self._space

    def _standalone_pipeline(self):
        return self._standalone_theory_pipeline()

    def _standalone_products(self, likelihood):
        """``power [n_ell, n_k]`` (``corr [n_ell, n_s]`` for the correlation function classes) at the last call (full_shape.py:502-510, tgc/base.py:127-136)."""
        flat = np.array(likelihood.observable_flattheory(0))
        if self._space == 'xi': self.corr = flat.reshape(len(self.ells), -1)
        else: self.power = flat.reshape(len(self.ells), -1)


class DampedBAOWigglesTracerPowerSpectrumMultipoles(_BaseDampedBAOTracer):
    """BAO power spectrum multipoles with broadband terms (bao.py:422-560, 117-151)."""
    _powers = range(-3, 2)
    _space = 'pk'

    def initialize(self):
        if self._initialized:
            return self
        k = self.init.get('k', None)
        if k is None: k = np.linspace(0.01, 0.2, 101)
        self._init_wiggles(k)
        self.k = self.kin
        kp = self.init.get('kp', None)
        self.kp = 2. * np.pi / self.template.fiducial.rs_drag if kp is None else float(kp)   # bao.py:488
        if 'power' in self.broadband:
            self._broadband_names, self.broadband_matrix = self._broadband_matrix(self.k, self.kp)
        else:
            self._broadband_names, self.broadband_matrix = self._kernel_broadband_matrix(self.k, self.kp, self.broadband)
        self._initialized = True
        return self

    def _fold(self):
        """Theory vector [n_ell * n_k] = fold . [P_ell(k_in), broadband parameters]."""
        return np.hstack([np.eye(len(self.ells) * len(self.k)), self.broadband_matrix])


class DampedBAOWigglesTracerCorrelationFunctionMultipoles(_BaseDampedBAOTracer):
    """BAO correlation function multipoles with broadband terms (bao.py:790-960; Hankel transform tgc/base.py:46-139)."""
    _powers = range(-2, 3)
    _ref_limits = (-1e-3, 1e-3)
    _space = 'xi'

    def initialize(self):
        if self._initialized:
            return self
        s = self.init.get('s', None)
        if s is None: s = np.linspace(20., 200, 101)
        self.s = np.array(s, dtype='f8')
        self.interp_order = {'linear': 1, 'cubic': 3}.get(self.init.get('interp_order', 1), self.init.get('interp_order', 1))
        if self.interp_order not in (1, 3):
            raise ValueError('interp_order must be one of [1, 3]')    # tgc/base.py:54-57
        kfft = np.logspace(-4., 3., 2048)                          # tgc/base.py:62
        kin = self.init.get('k', None)
        if kin is None: kin = np.geomspace(kfft[0], 0.6, int(300. / self.interp_order + 0.5))      # tgc/base.py:66
        self._init_wiggles(kin)
        sp = self.init.get('sp', None)
        self.sp = 2. * np.pi / 0.02 if sp is None else float(sp)   # bao.py:855
        self._kfft, self._hankel = kfft, None
        self._fourier_broadband = None
        if 'power' in self.broadband:
            self._broadband_names, self.broadband_matrix = self._broadband_matrix(self.s, self.sp)
        else:
            # kernels in Fourier space (bao.py:861-863: the power spectrum class with this broadband, Hankel-transformed with it) + powers of s for 'bl*' (885-890)
            self.broadband = self.broadband[:3]
            kp = self.init.get('kp', None)
            self.kp = 2. * np.pi / self.template.fiducial.rs_drag if kp is None else float(kp)
            names_k, self._fourier_broadband = self._kernel_broadband_matrix(self.kin, self.kp, self.broadband)
            self._bl_orders = _get_orders('bl', self.init.params, self.ells)
            names_s = [name for ell in self.ells for name in self._bl_orders[ell]]
            matrix_s = np.zeros((len(self.ells), len(self.s), len(names_s)), dtype='f8')
            for ill, ell in enumerate(self.ells):
                for name, pow in self._bl_orders[ell].items():
                    matrix_s[ill, :, names_s.index(name)] = (self.s / self.sp)**pow
            self._broadband_names, self._s_broadband = names_k + names_s, matrix_s.reshape(-1, len(names_s))
        self._initialized = True
        return self

    @property
    def broadband_matrix(self):
        if self._fourier_broadband is None:
            return self._broadband_matrix_s
        return np.hstack([self._hankel_block.dot(self._fourier_broadband), self._s_broadband])

    @broadband_matrix.setter
    def broadband_matrix(self, value):
        self._broadband_matrix_s = value

    @property
    def hankel(self):
        """Hankel operators H_ell [n_s, n_kin], built at first use by ONE batch of the device FFTLog (``dl_fftlog_apply``: all unit vectors of the input grid)."""
        if self._hankel is None:
            from ...fftlog import hankel_operator
            self._hankel = hankel_operator(self.kin, self.s, self.ells, k=self._kfft, engine='hip', interp_order=self.interp_order)
        return self._hankel

    @property
    def _hankel_block(self):
        from scipy import linalg
        return linalg.block_diag(*self.hankel)

    def _fold(self):
        return np.hstack([self._hankel_block, self.broadband_matrix])


class ResummedBAOWigglesTracerPowerSpectrumMultipoles(DampedBAOWigglesTracerPowerSpectrumMultipoles):
    """BAO power spectrum multipoles with resummed wiggles and broadband terms (bao.py:165-266, 670-717): the damping of the wiggles follows from the template
    (three integrals over the no-wiggle power, constants here), parameters b1, dbeta, sigmas, d; ``model`` may contain 'move-all', 'fog-damping'."""
    _resummed = True
    _wants_shotnoise = True   # "to be given shot noise by window matrix" (bao.py:236)


class ResummedBAOWigglesTracerCorrelationFunctionMultipoles(DampedBAOWigglesTracerCorrelationFunctionMultipoles):
    """BAO correlation function multipoles with resummed wiggles (bao.py:1051-1096)."""
    _resummed = True


class SimpleBAOWigglesTracerPowerSpectrumMultipoles(DampedBAOWigglesTracerPowerSpectrumMultipoles):
    """As :class:`DampedBAOWigglesTracerPowerSpectrumMultipoles` with the wiggles damped at the fiducial (k, mu): ``model='fix-damping'`` by default (bao.py:154-162, 630-668)."""
    _default_model = 'fix-damping'


class SimpleBAOWigglesTracerCorrelationFunctionMultipoles(DampedBAOWigglesTracerCorrelationFunctionMultipoles):
    """Correlation function counterpart of :class:`SimpleBAOWigglesTracerPowerSpectrumMultipoles` (bao.py:1008-1048)."""
    _default_model = 'fix-damping'


class FlexibleBAOWigglesTracerPowerSpectrumMultipoles(DampedBAOWigglesTracerPowerSpectrumMultipoles):
    """BAO power spectrum multipoles with terms multiplying the wiggles and no damping parameter (bao.py:269-391, 719-763): parameters b1, dbeta and
    ``ml{ell}_{i}`` (``wiggles`` = 'pcs' / 'tsc' / 'cic' / 'ngp' nodes of period ``kp``, or 'power' for powers of k / kp); ``model`` may contain 'move-all'."""
    _flexible = True


class FlexibleBAOWigglesTracerCorrelationFunctionMultipoles(DampedBAOWigglesTracerCorrelationFunctionMultipoles):
    """Correlation function counterpart of :class:`FlexibleBAOWigglesTracerPowerSpectrumMultipoles` (bao.py:1099-1144)."""
    _flexible = True
