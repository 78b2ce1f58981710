"""Golden vectors of the reference's PNG theory (scale-dependent bias from local primordial non-Gaussianity, primordial_non_gaussianity.py:12-116), run here with the
reference's own code (method 'prim'; the other method needs ``growth_factor`` / ``Omega0_m`` of a real cosmology engine):

    python tests/golden/make_png_fixture.py        (build container only; writes tests/golden/png*.npz and boundary_png.npz)
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg   # noqa: E402

from desilike.theories.galaxy_clustering import PNGTracerPowerSpectrumMultipoles, ShapeFitPowerSpectrumTemplate, FixedPowerSpectrumTemplate   # noqa: E402
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable   # noqa: E402
from desilike.likelihoods import ObservablesGaussianLikelihood   # noqa: E402


def build(mode='b-p', template='fixed'):
    tmpl = ShapeFitPowerSpectrumTemplate(z=0.5) if template == 'shapefit' else FixedPowerSpectrumTemplate(z=0.5)
    theory = PNGTracerPowerSpectrumMultipoles(template=tmpl, mode=mode)
    data = {'b1': 2., 'fnl_loc': 20.} if mode != 'bfnl' else {'b1': 2., 'bfnl_loc': 30.}
    obs = TracerPowerSpectrumMultipolesObservable(data=data, kedges=np.linspace(0.002, 0.102, 26), ells=(0, 2), wmatrix={'resolution': 2}, theory=theory, shotnoise=1e4)
    likelihood = ObservablesGaussianLikelihood(observables=[obs], covariance=mg.spd_covariance(50, seed=6, diag=4e6, amp=400.))
    return likelihood, obs, theory


def dump(name, mode='b-p', template='fixed', size=24, seed=42):
    likelihood, obs, theory = build(mode, template)
    likelihood()
    names = likelihood.varied_params.names()
    theta = mg.sample_theta(likelihood, size, seed)
    rng = np.random.RandomState(seed + 1)
    for pname, (lo, hi) in {'fnl_loc': (-60., 60.), 'bfnl_loc': (-80., 80.), 'sigmas': (0., 5.), 'p': (0.8, 1.7), 'bphi': (0.5, 3.)}.items():
        if pname in names: theta[:, names.index(pname)] = rng.uniform(lo, hi, size)
    theta[-2, names.index('b1')] = -0.5      # outside the prior [0, 4]
    theta[-3, names.index('b1')] = 4.        # on the (closed) upper limit: finite
    vlike = mg.vmap(likelihood, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({pname: theta[:, i] for i, pname in enumerate(names)})
    out = dict(theta=theta, names=np.array(names), logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[likelihood._param_loglikelihood]),
               logprior=np.asarray(derived[likelihood._param_logprior]), nerrors=len(errors), mode=mode)
    wm, tmpl = obs.wmatrix, theory.template
    c = dict(ells=np.array(wm.ells), ellsin=np.array(wm.ellsin), kin=np.asarray(theory.k), kout=np.concatenate(wm.k), mu=np.asarray(theory.mu), wmu_ell=np.asarray(theory.wmu),
             k11=np.asarray(tmpl.k), pk_dd_fid=np.asarray(tmpl.pk_dd_fid), f_fid=float(tmpl.f_fid), template=tmpl.__class__.__name__, nd=theory.nd,
             matrix_full=np.asarray(wm.matrix_full), shotnoisein=np.asarray(wm.shotnoisein), shotnoiseout=np.asarray(wm.shotnoiseout), flatdata=np.asarray(obs.flatdata))
    if hasattr(tmpl, 'kp'): c['kp'], c['a'] = tmpl.kp, tmpl.a
    cosmo = tmpl.cosmo
    out['pk_prim'] = np.asarray(cosmo.get_primordial(mode='scalar').pk_interpolator()(np.asarray(tmpl.k)))
    out['h'] = float(cosmo.h)
    out['priors'] = np.array([[{'uniform': 0, 'norm': 1}[likelihood.all_params[pname].prior.dist]] + [mg.prior_spec(likelihood.all_params[pname])[key] for key in ['lo', 'hi', 'loc', 'scale']]
                              for pname in names], dtype='f8')
    out['precision'] = np.asarray(likelihood.precision)
    inter = {key: [] for key in ['power', 'flattheory']}
    for row in theta[:6]:
        if not np.all(np.isfinite(row)): row = theta[0]
        likelihood(**dict(zip(names, row)))
        inter['power'].append(np.asarray(theory.power))
        inter['flattheory'].append(np.asarray(likelihood.flattheory))
    for key, value in inter.items(): out['int_' + key] = np.array(value)
    mg.save(name, c=c, **out)
    ok = np.isfinite(out['loglikelihood'])
    print(name, names, 'logL range', out['loglikelihood'][ok].min(), out['loglikelihood'][ok].max(), 'errors', len(errors))


def boundary():
    import make_boundary_fixture as mb
    likelihood, obs, theory = build('b-p', 'shapefit')
    mb.dump('png', likelihood, size=16, seed=4)


if __name__ == '__main__':
    dump('png_bp_fixed', mode='b-p', template='fixed')
    dump('png_bphi_shapefit', mode='bphi', template='shapefit')
    if '--boundary' in sys.argv: boundary()
