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
        if k is None: k = np.linspace(0.01, 0.2, 101)
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
        kin = np.geomspace(min(self._klim[0], self.k[0] / 2, tk[0] if tk is not None else 1.), max(self._klim[1], self.k[-1] * 2, tk[0] if tk is not None else 0.), self._klim[2])
        template.init.update(k=kin)
        if init.get('z', None) is not None:
            template.init.update(z=init['z'])
        template.initialize()
        self.z = template.z
        self._set_eft()
        self._initialized = True  # after template.init.update, which invalidates dependents
        return self

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
        spec.update(self.template._template_spec())
        return spec

    def _input_map(self):
        """kernel input name -> parameter name (+ EFT lists)."""
        toret = {name: name for name in ['qpar', 'qper', 'qiso', 'qap', 'df', 'dm', 'dn', 'sigmapar', 'sigmaper']}
        toret.update(self._bias_names())
        return toret

    def _all_params(self):
        """Parameters in the reference's pipeline order: APEffect, template, theory."""
        from ...parameter import ParameterCollection
        self.initialize()
        ap = [param for param in self.template.params if param.basename in ('qpar', 'qper', 'qiso', 'qap')]
        others = [param for param in self.template.params if param not in ap]
        return ParameterCollection(ap + others) + self.params


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
