"""Tracer power spectrum multipoles (reference: desilike/theories/galaxy_clustering/full_shape.py).

Host-side mirrors of ``KaiserTracerPowerSpectrumMultipoles`` (464-550) and
``EFTLikeKaiserTracerPowerSpectrumMultipoles`` (577-661): parameter definitions (``full_shape.yaml``),
multi-tracer namespaces (59-133), quadrature nodes and the EFT-like term matrices (586-626).
The per-point arithmetic runs in the HIP theory kernel (csrc/dl_fullshape.h, phases 2-3).
"""
import re

import numpy as np

from ...base import BaseCalculator
from ... import utils
from .power_template import StandardPowerSpectrumTemplate

_SIGMA = {'sigmapar': dict(value=0., prior=dict(limits=[0., 10.]), ref=dict(dist='norm', loc=5., scale=2.), latex=r'\Sigma_{\parallel}', fixed=True),
          'sigmaper': dict(value=0., prior=dict(limits=[0., 10.]), ref=dict(dist='norm', loc=5., scale=2.), latex=r'\Sigma_{\perp}', fixed=True)}
_B1 = {'b1': dict(prior=dict(limits=[0., 4.]), ref=dict(limits=[1., 2.]), latex='b_{1}')}
_SN0 = {'sn0': dict(prior=dict(dist='norm', loc=0., scale=1000.), ref=dict(dist='norm', loc=0., scale=0.1), latex='s_{n, 0}')}


def multitracer_namespace(tracers, ntracers=2):
    """(namespace of tracer X, of tracer Y, of the cross / stochastic terms); full_shape.py:88-113."""
    tracers = tracers or []
    if isinstance(tracers, str):
        tracers = [tracers]
    n = len(tracers)
    if n == 0:
        return ('',) * (ntracers + 1)
    if n == 1:
        return (tracers[0],) * (ntracers + 1)
    if n == ntracers:
        return tuple(tracers) + ('x'.join(tracers),)
    if n == ntracers + 1:
        return tuple(tracers)
    raise ValueError('`tracers` should be a string or a list of maximum {} names ({} auto and 1 cross)'.format(ntracers + 1, ntracers))


class KaiserTracerPowerSpectrumMultipoles(BaseCalculator):
    r"""
    Kaiser tracer power spectrum multipoles (full_shape.py:513-550):
    :math:`P_\ell = b_{1X} b_{1Y} P^{dd}_\ell + (b_{1X} + b_{1Y}) P^{dt}_\ell + P^{tt}_\ell + \delta_{\ell 0}\, s_{n,0} / \bar n`.

    Parameters
    ----------
    k : array, default=None
        Theory wavenumbers (set by the window / observable when left to ``None``).
    ells : tuple, default=(0, 2, 4)
    mu : int, default=8
        Number of Gauss-Legendre nodes in (0, 1) (full_shape.py:484).
    template : BasePowerSpectrumTemplate, default=StandardPowerSpectrumTemplate()
    shotnoise : float, default=1e4
        ``nd = 1 / shotnoise`` (full_shape.py:162).
    tracers : str, list, default=None
        Tracer namespace(s) for the bias parameters (full_shape.py:59-133).
    """
    _kind = 0  # DL_THEORY_KAISER
    _damping_fid = False
    _klim = (1e-3, 1., 500)   # template knots, full_shape.py:19
    _deterministic_bias_params = ['b1']
    _stochastic_bias_params = ['sn0']
    _own_params = {**_B1, **_SN0, **_SIGMA}

    @classmethod
    def _default_params(cls, tracers=None, **kwargs):
        import copy
        params = copy.deepcopy(cls._own_params)
        if not tracers:
            return params
        *tracer_namespaces, cross_namespace = multitracer_namespace(tracers)
        toret = {}
        for name, conf in params.items():
            if name in cls._deterministic_bias_params:
                for namespace in dict.fromkeys(tracer_namespaces):
                    toret['{}.{}'.format(namespace, name)] = copy.deepcopy(conf)
            elif name in cls._stochastic_bias_params:
                toret['{}.{}'.format(cross_namespace, name)] = conf
            else:
                toret[name] = conf
        return toret

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        self.tracers = init.get('tracers', None)
        shotnoise = init.get('shotnoise', 1e4)
        if np.ndim(shotnoise) != 0:
            shotnoise = np.sqrt(np.prod(shotnoise))  # cross-correlation: geometric mean (full_shape.py:155-157)
        self.nd = 1. / float(shotnoise)
        k = init.get('k', None)
        if k is None: k = self._default_k()
        self.k = np.array(k, dtype='f8')
        self.ells = tuple(init.get('ells', (0, 2, 4)))
        self.mu, wmu = utils.weights_mu(init.get('mu', 8), method=init.get('method', 'leggauss'))
        self.wmu = utils.multipole_weights(self.mu, wmu, self.ells)
        template = init.get('template', None)
        if template is None:
            template = self.init['template'] = StandardPowerSpectrumTemplate()
        self.template = self._require(template)
        # template knots with margin for the AP effect (full_shape.py:29)
        tk = template.init.get('k', None)
        kin = np.geomspace(min(self._klim[0], self.k[0] / 2, tk[0] if tk is not None else 1.), max(self._klim[1], self.k[-1] * 2, tk[-1] if tk is not None else 0.), self._klim[2])   # (a template shared by several theories ends up with knots that cover them all)
        template.init.update(k=kin)
        if init.get('z', None) is not None:
            template.init.update(z=init['z'])
        template.initialize()
        self.z = template.z
        self._set_eft()
        self._initialized = True  # after template.init.update, which invalidates dependents
        return self

    def _default_k(self):
        return np.linspace(0.01, 0.2, 101)

    def _set_eft(self):
        self.counterterm_params, self.stochastic_params = [], []
        self.counterterm_matrix = self.stochastic_matrix = None

    def _bias_names(self):
        """Parameter names feeding b1X, b1Y, sn0 (full_shape.py:114-128)."""
        nsX, nsY, nsC = (ns + '.' if ns else '' for ns in multitracer_namespace(self.tracers))
        return {'b1X': nsX + 'b1', 'b1Y': nsY + 'b1', 'sn0': nsC + 'sn0'}

    def _theory_spec(self):
        self.initialize()
        spec = dict(theory=np.array([self._kind], dtype='i4'), nd=[self.nd], ells_in=np.array(self.ells, dtype='i4'), kin=self.k, mu=self.mu, wmu_ell=self.wmu)
        if self._damping_fid: spec['damping_fid'] = np.array([1], dtype='i4')
        spec.update(self.template._template_spec())
        return spec

    def _input_map(self):
        """kernel input name -> parameter name (+ EFT lists)."""
        toret = {name: name for name in ['qpar', 'qper', 'qiso', 'qap', 'df', 'dm', 'dn', 'sigmapar', 'sigmaper']}
        toret.update(getattr(self.template, '_extra_inputs', {}))      # inputs of the template's own (turn-over: m, n, qto, dpto)
        toret.update(self._bias_names())
        return toret

    def _all_params(self):
        """Parameters in the reference's pipeline order: APEffect, template, theory."""
        from ...parameter import ParameterCollection
        self.initialize()
        ap = [param for param in self.template.params if param.basename in ('qpar', 'qper', 'qiso', 'qap')]
        others = [param for param in self.template.params if param not in ap]
        return ParameterCollection(ap + others) + self.params

    def _param_collections(self):
        return [self.template.init.params, self.init.params]

    _standalone_space = 'pk'

    def _standalone_pipeline(self):
        return self._standalone_theory_pipeline()

    def _standalone_products(self, likelihood):
        """``power [n_ell, n_k]`` (``corr [n_ell, n_s]`` for the correlation function classes) at the last call (full_shape.py:502-510, tgc/base.py:127-136)."""
        flat = np.array(likelihood.observable_flattheory(0))
        if self._standalone_space == 'xi': self.corr = flat.reshape(len(self.ells), -1)
        else: self.power = flat.reshape(len(self.ells), -1)


class SimpleTracerPowerSpectrumMultipoles(KaiserTracerPowerSpectrumMultipoles):
    r"""Kaiser tracer multipoles with FIXED damping, "essentially used for Fisher forecasts" (full_shape.py:367-414): the Gaussian damping
    :math:`\exp[-k^2 (\Sigma_\parallel^2 \mu^2 + \Sigma_\perp^2 (1 - \mu^2)) / 2]` is evaluated at the fiducial :math:`(k, \mu)`, not the AP-distorted ones, and
    :math:`s_{n,0} / \bar n` is added to :math:`P(k, \mu)` before the projection (identical after it: the quadrature integrates the Legendre polynomials exactly)."""
    _damping_fid = True


class EFTLikeKaiserTracerPowerSpectrumMultipoles(KaiserTracerPowerSpectrumMultipoles):
    """Kaiser multipoles with EFT-like counter terms ``ct{ell}_{pow}`` and stochastic terms ``sn{ell}_{pow}`` (full_shape.py:577-661)."""
    _kind = 1  # DL_THEORY_EFT_KAISER
    _deterministic_bias_params = ['b1', 'ct0_2', 'ct2_2', 'ct4_2']
    _stochastic_bias_params = ['sn0', 'sn0_2', 'sn2_2', 'sn4_2']
    _own_params = {**_B1,
                   **{'ct{:d}_2'.format(ell): dict(prior=dict(dist='norm', loc=0., scale=100.), ref=dict(dist='norm', loc=0., scale=1.), latex='c_{{{:d}, 2}}'.format(ell)) for ell in (0, 2, 4)},
                   **_SN0,
                   **{'sn{:d}_2'.format(ell): dict(prior=dict(dist='norm', loc=0., scale=1000.), ref=dict(dist='norm', loc=0., scale=0.1), latex='s_{{{:d}, 2}}'.format(ell)) for ell in (0, 2, 4)},
                   **_SIGMA}

    def _set_eft(self):
        self.kp = 1.
        _, _, nsC = multitracer_namespace(self.tracers)
        nsX, nsY, _ = multitracer_namespace(self.tracers)

        def get_params_matrix(base):
            # full_shape.py:589-621: columns k^pow for ct{ell}_{pow} / sn{ell}_{pow}, constant for '<base>0'
            coeffs = {ell: {} for ell in self.ells}
            for param in list(self.init.params):
                name = param.basename
                match = re.match(base + '(.*)_(.*)', name)
                if match:
                    ell, pow = int(match.group(1)), int(match.group(2))
                    if ell in self.ells:
                        coeffs[ell][name] = (self.k / self.kp)**pow
                    else:
                        del self.init.params[param.name]
                elif name == base + '0' and 0 in self.ells:
                    coeffs[0][name] = np.ones_like(self.k)
            names = list(dict.fromkeys(name for ell in self.ells for name in coeffs[ell]))
            matrix = np.zeros((len(self.ells), len(self.k), len(names)), dtype='f8')
            for ill, ell in enumerate(self.ells):
                for name, k_i in coeffs[ell].items():
                    matrix[ill, :, names.index(name)] = k_i
            return names, matrix

        self.counterterm_params, self.counterterm_matrix = get_params_matrix('ct')
        self.stochastic_params, self.stochastic_matrix = get_params_matrix('sn')

    def _theory_spec(self):
        spec = super(EFTLikeKaiserTracerPowerSpectrumMultipoles, self)._theory_spec()
        if self.counterterm_params: spec['ct_matrix'] = self.counterterm_matrix
        if self.stochastic_params: spec['sn_matrix'] = self.stochastic_matrix
        return spec

    def _input_map(self):
        toret = super(EFTLikeKaiserTracerPowerSpectrumMultipoles, self)._input_map()
        nsX, nsY, nsC = (ns + '.' if ns else '' for ns in multitracer_namespace(self.tracers))
        toret['ct'] = [(nsX + name, nsY + name) for name in self.counterterm_params]
        toret['sn'] = [nsC + name for name in self.stochastic_params]
        return toret


class _CorrelationFunctionFromPowerSpectrum(object):
    r"""Correlation function multipoles as Hankel transforms of the power spectrum multipoles (full_shape.py:336-364 on top of tgc/base.py:46-139):
    the device evaluates P_\ell on ``kin = geomspace(1e-4, 0.6, 300)`` (tgc/base.py:62-66) and ``get_corr`` -- interpolation to the FFTLog grid, high-k tail,
    FFTLog, interpolation to ``s``, all linear in P_\ell -- is the constant operator ``hankel`` folded into the window matrix.  The operator is built at first use
    by ONE batch of the device FFTLog (``dl_fftlog_apply``).  The stochastic terms have no parameter here (``_stochastic_bias_params = []`` in the reference)."""
    _standalone_space = 'xi'
    _stochastic_bias_params = []

    def _default_k(self):
        return np.geomspace(1e-4, 0.6, int(300. / getattr(self, 'interp_order', 1) + 0.5))   # tgc/base.py:66

    def initialize(self):
        if self._initialized:
            return self
        s = self.init.get('s', None)
        if s is None: s = np.linspace(20., 200, 101)
        self.s = np.array(s, dtype='f8')
        self.interp_order = {'linear': 1, 'cubic': 3}.get(self.init.get('interp_order', 1), self.init.get('interp_order', 1))
        if self.interp_order not in (1, 3):
            raise ValueError('interp_order must be one of [1, 3]')    # tgc/base.py:54-57
        self._kfft, self._hankel = np.logspace(-4., 3., 2048), None
        super(_CorrelationFunctionFromPowerSpectrum, self).initialize()
        self.kin = self.k
        return self

    @property
    def hankel(self):
        if self._hankel is None:
            from ...fftlog import hankel_operator
            self._hankel = hankel_operator(self.kin, self.s, self.ells, k=self._kfft, engine='hip', interp_order=self.interp_order)
        return self._hankel

    def _fold(self):
        from scipy import linalg
        return linalg.block_diag(*self.hankel)


class KaiserTracerCorrelationFunctionMultipoles(_CorrelationFunctionFromPowerSpectrum, KaiserTracerPowerSpectrumMultipoles):
    """Kaiser tracer correlation function multipoles (full_shape.py:553-574): parameters b1, sigmapar, sigmaper."""
    _own_params = {**_B1, **_SIGMA}


class EFTLikeKaiserTracerCorrelationFunctionMultipoles(_CorrelationFunctionFromPowerSpectrum, EFTLikeKaiserTracerPowerSpectrumMultipoles):
    """EFT-like Kaiser tracer correlation function multipoles (full_shape.py:664-687): parameters b1, ct{ell}_2, sigmapar, sigmaper."""
    _own_params = {name: conf for name, conf in EFTLikeKaiserTracerPowerSpectrumMultipoles._own_params.items() if not name.startswith('sn')}


# ----------------------------------------------------------------------------------------------------------------------
# scale-dependent bias from local primordial non-Gaussianity (primordial_non_gaussianity.py)
# ----------------------------------------------------------------------------------------------------------------------
class PNGTracerPowerSpectrumMultipoles(KaiserTracerPowerSpectrumMultipoles):
    r"""
    Kaiser tracer power spectrum multipoles with the scale-dependent bias sourced by local primordial non-Gaussianity (primordial_non_gaussianity.py:12-116):
    :math:`b \to b_1 + b_{f_\mathrm{NL}} \alpha(k')`, Lorentzian damping :math:`1 / (1 + \sigma_s^2 k'^2 \mu'^2 / 2)` per tracer, shot noise added before the projection.

    Parameters: ``mode='b-p'`` (parameters fnl_loc, p: :math:`b_{f_\mathrm{NL}} = 2 \cdot 1.686 (b_1 - p) f_\mathrm{NL}`) or ``'bphi'`` (fnl_loc, bphi); b1, sigmas, sn0.
    ``method='prim'``: :math:`\alpha = \sqrt{P_\phi / P_{dd}}` from the fiducial's primordial spectrum (the transfer-function method needs growth factors of a cosmology
    engine: out of scope).  The reference's mode 'bfnl' does not run in the reference itself (a tuple is multiplied with an array, line 106) and is not offered.
    Template knots as in the reference: 1000 log-spaced wavenumbers (its extra first knot at 1e-4 only normalises the transfer function of the other method, line 95).
    """
    _kind = 5  # DL_THEORY_PNG
    _klim = (1e-3, 1., 1000)   # primordial_non_gaussianity.py:72
    _deterministic_bias_params = ['b1', 'sigmas', 'bphi', 'p']
    _stochastic_bias_params = ['sn0']
    _own_params = {'fnl_loc': dict(prior=dict(limits=[-300., 300.]), ref=dict(limits=[-10., 10.]), delta=1., latex=r'f_{\mathrm{NL}}^{\mathrm{loc}}'),      # primordial_non_gaussianity.yaml
                   'bphi': dict(prior=dict(limits=[-10., 10.]), ref=dict(limits=[3., 4.]), delta=0.1, latex=r'b_{\phi}'),
                   'p': dict(value=1., prior=dict(limits=[0., 3.]), ref=dict(limits=[0.5, 1.5]), delta=0.1, latex='p'),
                   'b1': dict(value=2., prior=dict(limits=[0.1, 10.]), ref=dict(limits=[1.5, 2.5]), delta=0.1, latex='b_{1}'),
                   'sn0': dict(prior=dict(dist='norm', loc=0., scale=1000.), ref=dict(dist='norm', loc=0., scale=0.1), delta=0.05, latex='s_{n, 0}'),
                   'sigmas': dict(value=0., prior=dict(limits=[0., 10.]), ref=dict(limits=[1., 4.]), delta=0.2, latex=r'\Sigma_{s}')}

    @classmethod
    def _default_params(cls, tracers=None, mode='b-p', **kwargs):
        params = super(PNGTracerPowerSpectrumMultipoles, cls)._default_params(tracers=tracers)
        if mode not in ('bphi', 'b-p'):
            raise ValueError('Unknown mode {}; it must be one of ["bphi", "b-p"]'.format(mode))   # primordial_non_gaussianity.py:57-66 ('bfnl': see the class docstring)
        drop = 'p' if mode == 'bphi' else 'bphi'
        return {name: conf for name, conf in params.items() if name.split('.')[-1] != drop}

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        self.method = str(init.get('method', 'prim'))
        if self.method != 'prim':
            raise NotImplementedError("method {!r}: the transfer-function normalisation needs growth factors of a cosmology engine; only 'prim' is available".format(self.method))
        self.mode = str(init.get('mode', 'b-p'))
        init.setdefault('ells', (0, 2))            # primordial_non_gaussianity.py:68
        init.setdefault('mu', 20)
        template = init.get('template', None)
        if template is None:
            from .power_template import FixedPowerSpectrumTemplate
            init['template'] = FixedPowerSpectrumTemplate()
        super(PNGTracerPowerSpectrumMultipoles, self).initialize()
        kt = self.template.k
        fid = self.template.fiducial
        pphi_prim = 9 / 25 * 2 * np.pi**2 / kt**3 * fid.pk_prim(kt) / fid.h**3      # primordial_non_gaussianity.py:85-86
        self.alpha_fid = 1. / (self.template.pk_dd_fid / pphi_prim)**0.5
        return self

    def _theory_spec(self):
        spec = super(PNGTracerPowerSpectrumMultipoles, self)._theory_spec()
        spec.update(png_alpha=self.alpha_fid, png_mode=np.array([{'bphi': 0, 'b-p': 1}[self.mode]], dtype='i4'))
        return spec

    def _input_map(self):
        toret = super(PNGTracerPowerSpectrumMultipoles, self)._input_map()
        nsX, nsY, nsC = (ns + '.' if ns else '' for ns in multitracer_namespace(self.tracers))
        for name in ['sigmapar', 'sigmaper']: toret.pop(name, None)
        toret.update(fnl_loc='fnl_loc', sigmas=nsX + 'sigmas', sigmasY=nsY + 'sigmas')
        if self.mode == 'bphi': toret.update(bphiX=nsX + 'bphi', bphiY=nsY + 'bphi')
        else: toret.update(pX=nsX + 'p', pY=nsY + 'p')
        return toret


class PNGTracerVelocityPowerSpectrumMultipoles(PNGTracerPowerSpectrumMultipoles):
    r"""
    Tracer-velocity cross power spectrum multipoles with the scale-dependent bias of local primordial non-Gaussianity (primordial_non_gaussianity.py:196-330; the
    reference models :math:`-i P`): odd multipoles (default 1, 3) of

    .. math:: P(k, \mu) = J \; \frac{\mathrm{sinc}(\sigma_u k')}{1 + \sigma_s^2 k'^2 \mu'^2 / 2} \; (b_1 + b_{f_\mathrm{NL}} \alpha(k') + f \mu'^2) \; \frac{100 \, b_v f \mu'}{(1 + z) k'} \; P_{dd}(k') .

    The reference integrates 81 trapezoid nodes in :math:`\mu \in [-1, 1]`; the integrand times an odd Legendre polynomial is even in :math:`\mu`, so the device
    evaluates the 41 nodes :math:`\mu \ge 0` with the weights of the mirror nodes added (same sum).  Parameters b1, bv, sigmas, sigmau, fnl_loc and p (``mode='b-p'``)
    or bphi (``'bphi'``); one tracer, no stochastic term."""
    _stochastic_bias_params = []
    _own_params = {**{name: conf for name, conf in PNGTracerPowerSpectrumMultipoles._own_params.items() if name != 'sn0'},
                   'bv': dict(value=1., prior=dict(limits=[0., 3.]), ref=dict(limits=[0.9, 1.1]), delta=0.05, latex='b_{v}'),
                   'sigmau': dict(value=0., prior=dict(limits=[0., 20.]), ref=dict(limits=[0., 5.]), delta=0.2, latex=r'\sigma_{u}')}

    def initialize(self):
        if self._initialized:
            return self
        self.init.setdefault('ells', (1, 3))       # primordial_non_gaussianity.py:243
        if self.init.get('tracers', None) is not None: raise NotImplementedError('one tracer (primordial_non_gaussianity.py:243-262)')
        super(PNGTracerVelocityPowerSpectrumMultipoles, self).initialize()
        if any(ell % 2 == 0 for ell in self.ells): raise ValueError('odd multipoles only: the integrand is odd in mu')
        # the reference's grid: np.linspace(-1, 1, 81), trapezoid weights over its length (utils.py:633-639), folded onto mu >= 0
        full = np.linspace(-1., 1., 81)
        weight = utils.weights_trapz(full) / (full[-1] - full[0])
        half = full >= -1e-12
        self.mu = np.abs(full[half])
        folded = np.where(self.mu > 0., 2., 1.) * weight[half]
        self.wmu = utils.multipole_weights(self.mu, folded, self.ells)
        return self

    def _theory_spec(self):
        spec = super(PNGTracerVelocityPowerSpectrumMultipoles, self)._theory_spec()
        spec.update(png_velocity=np.array([1], dtype='i4'), png_velfac=[100. / (1. + self.z)])
        return spec

    def _input_map(self):
        toret = super(PNGTracerVelocityPowerSpectrumMultipoles, self)._input_map()
        toret.pop('sn0', None)
        toret.update(bv='bv', sigmau='sigmau')
        return toret


# ----------------------------------------------------------------------------------------------------------------------
# TNS one-loop theory: the reference's own perturbation-theory producer (full_shape.py:688-1037)
# ----------------------------------------------------------------------------------------------------------------------
_NORM15 = dict(prior=dict(dist='norm', loc=0., scale=15.), ref=dict(dist='norm', loc=0., scale=0.5))
_TNS_BIAS = {**_B1, 'b2': dict(latex='b_{2}', **_NORM15), 'bs': dict(latex='b_{s}', fixed=True, **_NORM15), 'b3': dict(latex='b_{3}', fixed=True, **_NORM15)}
_SIGMAV = {'sigmav': dict(prior=dict(dist='norm', loc=0., scale=20., limits=[0., 10.]), ref=dict(dist='norm', loc=0., scale=0.5), latex=r'\sigma_{v}')}


class TNSTracerPowerSpectrumMultipoles(KaiserTracerPowerSpectrumMultipoles):
    r"""
    TNS (Taruya, Nishimichi & Saito 2010) tracer power spectrum multipoles with one-loop bias terms, as ``TNSPowerSpectrumMultipoles`` +
    ``TNSTracerPowerSpectrumMultipoles`` compute them (full_shape.py:836-971): the 29 one-loop tables (P22 / P13 of density and velocity, the b2 / bs / b3 terms,
    the A and B correction terms) on ``k11 = linspace(0.7 k[0], 1.3 k[-1], 1.6 len(k))`` from the template by direct integration (``tns_pt``, 749-833), cubic
    interpolation to the AP-distorted wavenumbers, Lorentzian or Gaussian finger-of-god damping, Legendre projection, bias combination.  On the device the
    integration is a batched fp64 MFMA GEMM against geometry tables built once per context (csrc/dl_tns.h).

    Parameters: b1, b2, bs, b3 (bs, b3 fixed by default; ``freedom='max'`` frees them, ``'min'`` fixes them to 0), sn0, sigmav.
    Options: ``nloop=1``, ``fog='lorentzian' | 'gaussian'``, ``freedom=None | 'max' | 'min'``.
    """
    _kind = 4  # DL_THEORY_TNS
    _klim = (1e-3, 2., 500)   # full_shape.py:855
    _deterministic_bias_params = ['b1', 'b2', 'bs', 'b3']
    _stochastic_bias_params = ['sn0']
    _own_params = {**_TNS_BIAS, **_SN0, **_SIGMAV}
    _nmu_loop = 10   # cosines of the loop integrals (full_shape.py:757)

    def initialize(self):
        if self._initialized:
            return self
        init = self.init
        self.nloop = int(init.get('nloop', 1))
        if self.nloop not in [1]:
            raise ValueError('nloop must be 1 (1-loop)')    # full_shape.py:860-861
        self.fog = init.get('fog', 'lorentzian')
        if self.fog not in ['lorentzian', 'gaussian']:
            raise ValueError('fog must be lorentzian or gaussian')   # full_shape.py:862-863
        tracers = init.get('tracers', None)
        if tracers is not None and not isinstance(tracers, str) and len(set(tracers)) > 1:
            raise ValueError('cross-correlations are not implemented for the TNS model')   # _with_cross = False (full_shape.py:64)
        freedom = init.get('freedom', None)
        fix = []
        if freedom == 'max':    # full_shape.py:946-955
            for param in init.params.select(basename=['b1', 'b2', 'bs', 'b3']): param.update(fixed=False)
            fix += ['alpha6']
        if freedom == 'min':
            fix += ['b3', 'bs']
        for param in init.params.select(basename=fix): param.update(value=0., fixed=True)
        super(TNSTracerPowerSpectrumMultipoles, self).initialize()
        self.k11 = np.linspace(self.k[0] * 0.7, self.k[-1] * 1.3, int(len(self.k) * 1.6 + 0.5))   # full_shape.py:875
        return self

    def _theory_spec(self):
        spec = super(TNSTracerPowerSpectrumMultipoles, self)._theory_spec()
        mus, wmus = utils.weights_mu(self._nmu_loop, method='leggauss')
        spec.update(tns_k11=self.k11, tns_mu=np.asarray(mus, dtype='f8'), tns_wmu=np.asarray(wmus, dtype='f8'), tns_fog=np.array([{'lorentzian': 0, 'gaussian': 1}[self.fog]], dtype='i4'))
        return spec

    def _input_map(self):
        toret = super(TNSTracerPowerSpectrumMultipoles, self)._input_map()
        nsX = multitracer_namespace(self.tracers)[0]
        nsX = nsX + '.' if nsX else ''
        for name in ['sigmapar', 'sigmaper']: toret.pop(name, None)
        toret.update(b2=nsX + 'b2', bs=nsX + 'bs', b3=nsX + 'b3', sigmav='sigmav')
        return toret


class EFTLikeTNSTracerPowerSpectrumMultipoles(TNSTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles):
    """TNS multipoles with the EFT-like counter terms ``ct{ell}_2`` (times the projected linear spectrum, monopole) and stochastic terms ``sn{ell}_2``
    (full_shape.py:996-1014 on the mixin 577-634).  As in the reference's parameter file, there is no ``sigmav`` here (the damping is 1)."""
    _deterministic_bias_params = ['b1', 'b2', 'bs', 'b3', 'ct0_2', 'ct2_2', 'ct4_2']
    _stochastic_bias_params = ['sn0', 'sn0_2', 'sn2_2', 'sn4_2']
    _own_params = {**_TNS_BIAS, **{name: conf for name, conf in EFTLikeKaiserTracerPowerSpectrumMultipoles._own_params.items() if name.startswith(('ct', 'sn'))}}

    def _input_map(self):
        toret = TNSTracerPowerSpectrumMultipoles._input_map(self)
        toret.pop('sigmav', None)
        nsX, nsY, nsC = (ns + '.' if ns else '' for ns in multitracer_namespace(self.tracers))
        toret['ct'] = [(nsX + name, nsY + name) for name in self.counterterm_params]
        toret['sn'] = [nsC + name for name in self.stochastic_params]
        return toret


class TNSTracerCorrelationFunctionMultipoles(_CorrelationFunctionFromPowerSpectrum, TNSTracerPowerSpectrumMultipoles):
    """TNS tracer correlation function multipoles (full_shape.py:974-993): the Hankel transform of the power spectrum multipoles, folded into the window."""
    _own_params = {**_TNS_BIAS, **_SIGMAV}


class EFTLikeTNSTracerCorrelationFunctionMultipoles(_CorrelationFunctionFromPowerSpectrum, EFTLikeTNSTracerPowerSpectrumMultipoles):
    """EFT-like TNS tracer correlation function multipoles (full_shape.py:1017-1037)."""
    _own_params = {name: conf for name, conf in EFTLikeTNSTracerPowerSpectrumMultipoles._own_params.items() if not name.startswith('sn')}


# ----------------------------------------------------------------------------------------------------------------------
# velocileptors-style tracers on top of an EMULATED perturbation-theory node
# ----------------------------------------------------------------------------------------------------------------------
def get_physical_stochastic_settings(tracer=None):
    """Preset satellite fraction / velocity dispersion per tracer (values of the reference, full_shape.py:1077-1091)."""
    if tracer is not None:
        tracer = str(tracer).upper()
        settings = {'BGS': {'fsat': 0.15, 'sigv': 150 * (10)**(1 / 3) * (1 + 0.2)**(1 / 2) / 70.},
                    'LRG': {'fsat': 0.15, 'sigv': 150 * (10)**(1 / 3) * (1 + 0.8)**(1 / 2) / 70.},
                    'ELG': {'fsat': 0.10, 'sigv': 150 * 2.1**(1 / 2) / 70.},
                    'QSO': {'fsat': 0.03, 'sigv': 150 * (10)**(0.7 / 3) * (2.4)**(1 / 2) / 70.}}
        try:
            return settings[tracer]
        except KeyError:
            raise ValueError('unknown tracer: {}, please use any of {}'.format(tracer, list(settings.keys())))
    return {'fsat': 0.1, 'sigv': 5.}


class _BaseVelocileptorsTracer(BaseCalculator):
    """Table-level velocileptors tracer (full_shape.py:1182-1186) fed by an :class:`desilike_amd.emulators.EmulatedCalculator` ``pt``
    (the PT engines themselves are external CPU codes: SURVEY.md section 2 row 7).

    Parameters: ``pt`` (EmulatedCalculator with 'pktable', 'sigma8', 'fsigma8'), ``k``, ``ells``, ``prior_basis`` ('physical' | 'standard'),
    ``tracer`` / ``fsat`` / ``sigv`` / ``shotnoise`` (full_shape.py:1154-1157), ``freedom`` (None, 'max', 'min': parameter presets, 1100-1117).
    """
    _kind = 3   # DL_THEORY_EMULATED
    _rept = False
    _names = ['b1', 'b2', 'bs', 'b3', 'alpha0', 'alpha2', 'alpha4', 'alpha6', 'sn0', 'sn2', 'sn4']

    @classmethod
    def _default_params(cls, prior_basis='physical', pt=None, **kwargs):
        params = {}
        if pt is not None:
            for name in pt.param_names:
                params[name] = dict(pt.param_specs.get(name, dict(value=None)))
        if prior_basis == 'physical':   # full_shape.py:1126-1134
            params['b1p'] = dict(prior=dict(dist='uniform', limits=[0., 3.]), ref=dict(dist='norm', loc=1., scale=0.1))
            for name in ['b2p', 'bsp']:
                params[name] = dict(prior=dict(dist='norm', loc=0., scale=5.), ref=dict(dist='norm', loc=0., scale=1.))
            params['b3p'] = dict(value=0., fixed=True, prior=dict(dist='norm', loc=0., scale=5.))
            for name in ['alpha0p', 'alpha2p', 'alpha4p']:
                params[name] = dict(prior=dict(dist='norm', loc=0., scale=12.5), ref=dict(dist='norm', loc=0., scale=1.))
            params['alpha6p'] = dict(value=0., fixed=True)
            for name in ['sn0p', 'sn2p', 'sn4p']:
                params[name] = dict(prior=dict(dist='norm', loc=0., scale=2. if name == 'sn0p' else 5.), ref=dict(dist='norm', loc=0., scale=1.))
        else:
            # the reference's own defaults (full_shape.yaml: LPT / REPT tracer, power spectrum / correlation function; no ``value``: the centre of ``ref``, parameter.py:811-819)
            norm = lambda scale, ref: dict(prior=dict(dist='norm', loc=0., scale=scale), ref=dict(dist='norm', loc=0., scale=ref))   # noqa: E731
            xi = getattr(cls, '_standalone_space', 'pk') == 'xi'
            params['b1'] = dict(prior=dict(limits=[0., 4.]), ref=dict(limits=[1.4, 1.6])) if cls._rept else dict(prior=dict(limits=[-1., 10.]), ref=dict(limits=[0.4, 0.6]))
            params['b2'] = norm(10., 0.5)
            params['bs'] = norm(10. if xi else 5., 0.5)
            params['b3'] = dict(fixed=True, **norm(10. if xi else 5., 0.5))
            params['alpha0'] = norm(30., 1.)
            params['alpha2'], params['alpha4'] = norm(50., 1.), norm(50., 1.)
            params['alpha6'] = dict(fixed=True, **norm(50., 1.))
            params['sn0'], params['sn2'], params['sn4'] = norm(4., 0.1), norm(100., 0.1), norm(500., 0.1)
        # ``freedom`` presets (full_shape.py:1100-1117): applied before the physical-basis priors, which then only keep the list of fixed parameters
        freedom = kwargs.get('freedom', None)
        suffix = 'p' if prior_basis == 'physical' else ''
        fix = []
        if freedom == 'max':
            fix += ['alpha6']
            if not suffix:
                for name in ['b1', 'b2', 'bs', 'b3']: params[name].update(fixed=False)
                for name in ['b2', 'bs', 'b3']: params[name].update(prior=dict(limits=[-15., 15.]))
                for name in cls._names[4:]: params[name].update(prior=None)
        elif freedom == 'min':
            fix += ['b3', 'bs', 'alpha6']
            if not suffix:
                params['b2'].update(prior=dict(dist='norm', loc=0., scale=10.))
                for name in cls._names[4:]: params[name].update(prior=None)
        elif freedom is not None:
            raise ValueError("freedom must be None, 'max' or 'min'")
        for name in fix:
            params[name + suffix].update(value=0., fixed=True)
        return params

    def initialize(self):
        if self._initialized:
            return self
        from scipy import interpolate
        init = self.init
        self.pt = init.get('pt', None)
        if self.pt is None:
            raise ValueError('provide pt=EmulatedCalculator(...): the perturbation-theory engines are external CPU codes (out of scope)')
        self.prior_basis = init.get('prior_basis', 'physical')
        self.is_physical_prior = self.prior_basis == 'physical'
        self.ells = tuple(init.get('ells', self.pt.ells))
        k = init.get('k', None)
        self.k = self.pt.k.copy() if k is None else np.array(k, dtype='f8')
        self.z = self.pt.z
        if getattr(self.pt, 'stacked', False):
            # the emulated node holds several redshifts (emulators/conversion.py:44-98); the REPT tracer names its own (full_shape.py:1556-1560), the node keeps the two
            # emulated redshifts that bracket it and blends them (full_shape.py:1416-1443): weights per emulated redshift, zero for all but two
            z = init.get('z', None)
            zgrid = np.atleast_1d(np.asarray(self.pt.z, dtype='f8'))
            if z is None:
                if zgrid.size != 1: raise ValueError('the emulated node holds redshifts {}: provide z'.format(zgrid))
                z = zgrid[0]
            self.z = float(z)
            if self.z < zgrid[0] or self.z > zgrid[-1]:
                raise ValueError('input z = {} is outside of the range of emulated z: {} - {}'.format(self.z, zgrid[0], zgrid[-1]))
            iz = int(np.searchsorted(zgrid, self.z, side='right')) - 1
            self._zweights = np.zeros(zgrid.size)
            if iz + 1 < zgrid.size:
                wz = self.z - zgrid[iz]         # (as the reference writes it, full_shape.py:1436: a difference of redshifts, not a fraction of the interval)
                self._zweights[iz], self._zweights[iz + 1] = 1. - wz, wz
            else: self._zweights[iz] = 1.     # the last emulated redshift itself (the reference's blend indexes one past the end there)
        shotnoise = float(init.get('shotnoise', 1e4))
        self.nd = 1e-4                                               # full_shape.py:1154
        self.fsat = self.snd = self.sigv = 1.
        if self.is_physical_prior:
            settings = get_physical_stochastic_settings(tracer=init.get('tracer', None))
            self.fsat = settings['fsat'] if init.get('fsat', None) is None else float(init['fsat'])
            self.sigv = settings['sigv'] if init.get('sigv', None) is None else float(init['sigv'])
            self.snd = shotnoise * self.nd                           # full_shape.py:1157
        for ell, fix in [(4, ['alpha4', 'alpha6', 'sn4']), (2, ['alpha2', 'sn2'])]:   # full_shape.py:1148-1152
            if ell not in self.ells:
                for name in fix:
                    name = name + 'p' if self.is_physical_prior else name
                    if name in self.init.params: self.init.params[name].update(value=0., fixed=True)
        # cubic interpolation pt.k -> k (full_shape.py:1312, 1598) as a constant matrix
        index = [self.pt.ells.index(ell) for ell in self.ells]
        self._ell_index = index
        if self.k.shape == self.pt.k.shape and np.array_equal(self.k, self.pt.k):
            self._interp = np.eye(self.k.size)
        else:
            self._interp = interpolate.interp1d(self.pt.k, np.eye(self.pt.k.size), kind='cubic', axis=0, fill_value='extrapolate')(self.k)
        self._initialized = True
        return self

    def _stacked_groups(self):
        """The stacked layout (emulators/conversion.py:44-98) as the device takes it (include/desilike_amd.h, ``emu0.type = 2``): per engine ('11', 'loop', 'ct', 'st': one group
        of bias monomials each) the networks (z, ell) that reach this tracer -- nonzero redshift weight, a multipole it uses -- their hidden layers, and the constant operator
        ``[n_ell * n_k, K_g * n_m]`` (final layers x y-scalers x redshift blend x k-interpolation), K_g = n_networks * H + 1."""
        from ...emulators import STACKED_COMPONENTS
        pt, nk = self.pt, self.pt.k.size
        groups, scale, weights, blocks = [], [], [], []
        ntrunk, m0 = 0, 0
        for name, nm in STACKED_COMPONENTS:
            engine = pt.engines[name]
            kl, bl = engine.layers[-1]                                                  # [n_z, n_ell, H, n_m * n_k], [n_z, n_ell, n_m * n_k]
            lo, rng = engine.ylimits[..., 0], engine.ylimits[..., 1] - engine.ylimits[..., 0]    # [n_z, n_ell, n_m, n_k]
            factor = 1. if engine.amplitude is None else engine.amplitude[1]**engine.amplitude[2]
            H = kl.shape[-2]
            basis, used = [], []
            const = np.zeros((len(self.ells), nk, nm))
            for iz, wz in enumerate(self._zweights):
                if wz == 0.: continue
                for ill, ipt in enumerate(self._ell_index):
                    const[ill] += (wz * factor * (lo[iz, ipt] + rng[iz, ipt] * bl[iz, ipt].reshape(nm, nk))).T
                    jac = wz * factor * rng[iz, ipt][None] * kl[iz, ipt].reshape(H, nm, nk)       # d table[ell, m, k] / d (amplitude x hidden unit h)
                    if not np.any(jac != 0.): continue
                    table = np.zeros((H, len(self.ells), nk, nm))
                    table[:, ill] = jac.transpose(0, 2, 1)
                    basis.append(table); used.append((iz, ipt))
            basis = np.concatenate(basis + [const[None]], axis=0)                       # [K_g, ell, kpt, m]
            blocks.append(np.einsum('kq,hlqm->lkhm', self._interp, basis).reshape(len(self.ells) * self.k.size, -1))
            groups.append([ntrunk, ntrunk + len(used), m0, m0 + nm])
            row = np.zeros(len(pt.param_names) + 1)
            if engine.amplitude is not None: row[pt.param_names.index(engine.amplitude[0])] = engine.amplitude[2]
            scale.append(row)
            for iz, ipt in used:
                weights.append(np.concatenate([np.concatenate([kernel[iz, ipt].ravel(), bias[iz, ipt].ravel()]) for kernel, bias in engine.layers[:-1]]))
            ntrunk += len(used); m0 += nm
        return groups, scale, weights, blocks

    def _fold(self):
        """Theory vector P_ell(k) [n_ell * n_k] = fold . phi, phi[(h, m)] = basis_h mono_m."""
        if getattr(self.pt, 'stacked', False):
            return np.hstack(self._stacked_groups()[3])
        engine = self.pt.engines[self.pt.table_name]
        nb = engine.n_basis
        nellpt, nkpt = len(self.pt.ells), self.pt.k.size
        if self.pt.table_name == 'pktable':
            table = engine.basis_matrix().reshape(nb, nellpt, nkpt, 19)[:, self._ell_index]          # [h, ell, kpt, m]
            fold = np.einsum('kq,hlqm->lkhm', self._interp, table)
            return fold.reshape(len(self.ells) * self.k.size, nb * 19)
        table = engine.basis_matrix().reshape(nb, nellpt, nkpt)[:, self._ell_index]
        return np.einsum('kq,hlq->lkh', self._interp, table).reshape(len(self.ells) * self.k.size, nb)

    def _theory_spec(self):
        self.initialize()
        if self.pt.table_name in ('pktable', 'stacked'):
            mode = {(True, False): 1, (True, True): 2, (False, False): 3, (False, True): 4}[(self.is_physical_prior, self._rept)]
        else:
            mode = 0
        spec = dict(theory=np.array([self._kind], dtype='i4'), mono_mode=np.array([mode], dtype='i4'), vconst=[self.snd, self.fsat, self.sigv, self.nd])
        spec.update(self.pt.engine_specs())
        if getattr(self.pt, 'stacked', False):
            from ...emulators import ACTIVATIONS, STACKED_COMPONENTS
            first = self.pt.engines[STACKED_COMPONENTS[0][0]]
            groups, scale, weights, blocks = self._stacked_groups()
            spec['emu0'] = dict(type=np.array([2], dtype='i4'), widths=np.array([first.xlimits.shape[0]] + first.hidden, dtype='i4'), act=np.array([ACTIVATIONS[first.activation]], dtype='i4'),
                                xlimits=first.xlimits, weights=np.concatenate(weights) if weights else np.zeros(1), groups=np.array(groups, dtype='i4'), scale=np.array(scale, dtype='f8'))
        return spec

    def _input_map(self):
        suffix = 'p' if self.is_physical_prior else ''
        toret = {'x': list(self.pt.param_names)}
        if self.pt.table_name in ('pktable', 'stacked'):
            toret['vp'] = [name + suffix for name in self._names]
        return toret

    def _all_params(self):
        self.initialize()
        return self.params.copy()

    _standalone_space = 'pk'

    def _standalone_pipeline(self):
        return self._standalone_theory_pipeline()

    def _standalone_products(self, likelihood):
        """``power [n_ell, n_k]`` (``corr [n_ell, n_s]`` for the correlation function classes) at the last call (full_shape.py:502-510, tgc/base.py:127-136)."""
        flat = np.array(likelihood.observable_flattheory(0))
        if self._standalone_space == 'xi': self.corr = flat.reshape(len(self.ells), -1)
        else: self.power = flat.reshape(len(self.ells), -1)


class LPTVelocileptorsTracerPowerSpectrumMultipoles(_BaseVelocileptorsTracer):
    """Velocileptors LPT tracer multipoles (full_shape.py:1225-1313) from emulated tables."""
    _rept = False


class REPTVelocileptorsTracerPowerSpectrumMultipoles(_BaseVelocileptorsTracer):
    """Velocileptors REPT tracer multipoles (full_shape.py:1496-1599, co-evolution shift 1479-1488) from emulated tables."""
    _rept = True


class _VelocileptorsCorrelationFunction(object):
    r"""Velocileptors tracer correlation function multipoles as Hankel transforms of the power spectrum multipoles (full_shape.py:1317-1343, 1603-1629 on top of
    tgc/base.py:46-139): P_\ell is evaluated on ``kin = geomspace(1e-4, 0.6, 300)`` (cubic interpolation / extrapolation from the emulated node's own k), the
    Hankel operator (built by the device FFTLog) multiplies the folded table operator; no stochastic parameters (``_stochastic_bias_params = []``)."""
    _standalone_space = 'xi'

    @classmethod
    def _default_params(cls, **kwargs):
        params = super(_VelocileptorsCorrelationFunction, cls)._default_params(**kwargs)
        return {name: conf for name, conf in params.items() if not name.startswith('sn')}

    def initialize(self):
        if self._initialized:
            return self
        s = self.init.get('s', None)
        if s is None: s = np.linspace(20., 200, 101)
        self.s = np.array(s, dtype='f8')
        self.interp_order = {'linear': 1, 'cubic': 3}.get(self.init.get('interp_order', 1), self.init.get('interp_order', 1))
        if self.interp_order not in (1, 3):
            raise ValueError('interp_order must be one of [1, 3]')    # tgc/base.py:54-57
        self._kfft, self._hankel = np.logspace(-4., 3., 2048), None
        if self.init.get('k', None) is None:
            self.init['k'] = np.geomspace(self._kfft[0], 0.6, int(300. / self.interp_order + 0.5))     # tgc/base.py:66
        super(_VelocileptorsCorrelationFunction, self).initialize()
        self.kin = self.k
        return self

    @property
    def hankel(self):
        if self._hankel is None:
            from ...fftlog import hankel_operator
            self._hankel = hankel_operator(self.kin, self.s, self.ells, k=self._kfft, engine='hip', interp_order=self.interp_order)
        return self._hankel

    def _fold(self):
        from scipy import linalg
        return linalg.block_diag(*self.hankel).dot(super(_VelocileptorsCorrelationFunction, self)._fold())


class LPTVelocileptorsTracerCorrelationFunctionMultipoles(_VelocileptorsCorrelationFunction, LPTVelocileptorsTracerPowerSpectrumMultipoles):
    """Velocileptors LPT tracer correlation function multipoles (full_shape.py:1317-1343) from emulated tables."""


class REPTVelocileptorsTracerCorrelationFunctionMultipoles(_VelocileptorsCorrelationFunction, REPTVelocileptorsTracerPowerSpectrumMultipoles):
    """Velocileptors REPT tracer correlation function multipoles (full_shape.py:1603-1629) from emulated tables."""


class EmulatedTracerPowerSpectrumMultipoles(_BaseVelocileptorsTracer):
    """Any tracer theory emulated as a whole: ``pt`` emulates the array 'power' [n_ell, n_k] (EmulatedCalculator.calculate, emulators/__init__.py:408-409)."""

    @classmethod
    def _default_params(cls, pt=None, **kwargs):
        return {name: dict(pt.param_specs.get(name, dict(value=None))) for name in (pt.param_names if pt is not None else [])}
