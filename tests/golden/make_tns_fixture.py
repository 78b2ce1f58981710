"""Golden vectors of the reference's TNS one-loop theory (full_shape.py:688-971), run here with the reference's own code.

    python tests/golden/make_tns_fixture.py        (build container only; writes tests/golden/tns*.npz)

``tns_pt`` is written for jax: under the reference's numpy fallback (jax is absent here) two HARNESS calls do not run --
``utils.weights_trapz`` (``jnp.insert`` with an index one past the end: jax clamps it, numpy raises) and ``jax.vmap`` (the fallback is
``numpy.vectorize``, which cannot return the stacked arrays).  They are replaced below by their plain meaning (trapezoidal weights; a loop over the
ten cosine nodes, stacked); every line of the reference's arithmetic -- kernels, integrands, interpolation, the tracer combination -- runs unmodified.
The synthetic spectrum of the stand-in cosmology is scaled up (REFSTUB_PK_SCALE) so that the loop terms have a realistic relative size
(P(k = 0.1) ~ 1e4: one-loop corrections of several per cent at k = 0.2).
"""
import os
import sys

os.environ.setdefault('REFSTUB_PK_SCALE', '300')

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402  (sets the import paths of the reference and its stand-ins)

from desilike import utils, jax as djax   # noqa: E402
from desilike.theories.galaxy_clustering import (ShapeFitPowerSpectrumTemplate, StandardPowerSpectrumTemplate, TNSTracerPowerSpectrumMultipoles,
                                                 EFTLikeTNSTracerPowerSpectrumMultipoles)   # noqa: E402
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable   # noqa: E402
from desilike.likelihoods import ObservablesGaussianLikelihood   # noqa: E402


def _weights_trapz(x):
    x = np.asarray(x)
    return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.


def _vmap(fun):
    def wrapper(*args):
        return np.stack([fun(*a) for a in zip(*args)], axis=0)
    return wrapper


utils.weights_trapz = _weights_trapz
djax.vmap = _vmap

TNS_NAMES = ['pk11', 'pk_dd', 'pk_b2d', 'pk_bs2d', 'pk_sig3sq', 'pk_b22', 'pk_b2s2', 'pk_bs22', 'pk_dt', 'pk_b2t', 'pk_bs2t', 'pk_tt']


def dump(name, eft=False, fog='lorentzian', template='shapefit', resolution=3, size=24, seed=42, free=('bs', 'b3'), ells=(0, 2, 4)):
    tmpl = ShapeFitPowerSpectrumTemplate(z=0.5) if template == 'shapefit' else StandardPowerSpectrumTemplate(z=0.5)
    cls = EFTLikeTNSTracerPowerSpectrumMultipoles if eft else TNSTracerPowerSpectrumMultipoles
    theory = cls(template=tmpl, fog=fog)
    for pname in free: theory.init.params[pname].update(fixed=False)
    kedges = np.linspace(0., 0.2, 41)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'b2': 0.5, **({} if eft else {'sigmav': 3.})}, kedges=kedges, ells=ells, wmatrix={'resolution': resolution}, theory=theory, shotnoise=1e4)
    n = len(ells) * 40
    likelihood = ObservablesGaussianLikelihood(observables=[obs], covariance=mg.spd_covariance(n, seed=3, diag=2e5, amp=150.))
    likelihood()
    names = likelihood.varied_params.names()
    theta = mg.sample_theta(likelihood, size, seed)
    rng = np.random.RandomState(seed + 1)
    if 'sigmav' in names: theta[:, names.index('sigmav')] = rng.uniform(0., 6., size)
    for pname in ('bs', 'b3'):
        if pname in names: theta[:, names.index(pname)] = rng.normal(0., 0.5, size)
    theta = mg.special_rows(theta, names)
    vlike = mg.vmap(likelihood, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({pname: theta[:, i] for i, pname in enumerate(names)})
    out = dict(theta=theta, names=np.array(names), logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[likelihood._param_loglikelihood]),
               logprior=np.asarray(derived[likelihood._param_logprior]), nerrors=len(errors), fog=fog, eft=eft)
    c = mg.extract_observable(obs)
    pt = theory.pt
    out['k11_table'] = np.linspace(pt.k[0] * 0.7, pt.k[-1] * 1.3, int(len(pt.k) * 1.6 + 0.5))   # full_shape.py:875 (a local of calculate)
    out['priors'] = np.array([[{'uniform': 0, 'norm': 1}[likelihood.all_params[pname].prior.dist]] + [mg.prior_spec(likelihood.all_params[pname])[key] for key in ['lo', 'hi', 'loc', 'scale']]
                              for pname in names], dtype='f8')
    out['precision'] = np.asarray(likelihood.precision)
    nint = 6
    inter = {key: [] for key in ['pk_dd_template', 'f', 'power', 'flattheory', 'tables', 'poles']}
    from desilike.theories.galaxy_clustering.full_shape import tns_pt
    q = np.asarray(pt.template.k)
    wq = _weights_trapz(q)
    for row in theta[:nint]:
        if not np.all(np.isfinite(row)): row = theta[0]
        likelihood(**dict(zip(names, row)))
        inter['pk_dd_template'].append(np.asarray(pt.template.pk_dd))
        inter['f'].append(float(pt.template.f))
        inter['power'].append(np.asarray(theory.power))
        inter['flattheory'].append(np.asarray(likelihood.flattheory))
        raw = tns_pt(out['k11_table'], q, wq, np.asarray(pt.template.pk_dd), *pt.kernels)    # the 29 tables on k11 before AP / damping / projection
        inter['tables'].append(np.concatenate([np.array(raw[:12]), np.asarray(raw[12]), np.array(raw[13])], axis=0))
        inter['poles'].append(np.concatenate([np.array([pt.pktable[key] for key in TNS_NAMES]), np.asarray(pt.pktable['A']), np.asarray(pt.pktable['B'])], axis=0))
    for key, value in inter.items(): out['int_' + key] = np.array(value)
    out['kernel_rows'] = np.arange(0, len(out['k11_table']), 16)   # the P13 / A kernels of every 16th table wavenumber (the full arrays are 4 MB)
    out['kernel13_d'], out['kernel13_t'], out['kernel_a'] = (np.asarray(kk)[..., out['kernel_rows'], :] for kk in pt.kernels)
    mg.save(name, c=c, **out)
    print(name, names, 'logL range', np.nanmin(out['loglikelihood'][np.isfinite(out['loglikelihood'])]), np.nanmax(out['loglikelihood'][np.isfinite(out['loglikelihood'])]), 'errors', len(errors))


def boundary():
    """Reference-side binding on real TNS likelihoods (integration/desilike_mi355x.py::extract_config): the key sets + the reference's own outputs, as
    tests/golden/make_boundary_fixture.py does for the other theories (replayed by tests/test_gpu_boundary.py)."""
    import make_boundary_fixture as mb
    for name, cls, fog in [('tns', TNSTracerPowerSpectrumMultipoles, 'gaussian'), ('tns_eft', EFTLikeTNSTracerPowerSpectrumMultipoles, 'lorentzian')]:
        theory = cls(template=ShapeFitPowerSpectrumTemplate(z=0.5), fog=fog)
        if name == 'tns': theory.init.params['bs'].update(fixed=False)
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'b2': 0.5}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 1}, theory=theory, shotnoise=1e4)
        likelihood = ObservablesGaussianLikelihood(observables=[obs], covariance=mg.spd_covariance(120, seed=5, diag=2e5, amp=150.))
        mb.dump(name, likelihood, size=16, seed=9)


if __name__ == '__main__':
    boundary()
    dump('tns')
    dump('tns_eft', eft=True, resolution=1, size=12, free=())
    dump('tns_standard_gaussian', template='standard', fog='gaussian', resolution=1, size=12, ells=(0, 2))
