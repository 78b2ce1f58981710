"""Generate golden vectors by running the *reference* (cosmodesi/desilike at /root/reference) here.

Run from the repo root, in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

The reference is imported through ``tests/golden/refstub`` (our own stand-in for its missing
third-party deps cosmoprimo / lsstypes: a synthetic analytic cosmology).
desilike's own arithmetic downstream of ``pk_dd_fid[k]``, ``f_fid`` runs unmodified (numpy backend:
scipy not-a-knot cubic ``interp1d``).  Outputs: small ``.npz`` fixtures in this directory holding
inputs (constants extracted from the *initialised reference calculators*), the theta batch, and every
intermediate + loglikelihood / logprior returned by the reference.
"""
import os
import sys
import time
import warnings

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
warnings.filterwarnings('ignore')

from desilike.theories.galaxy_clustering import (ShapeFitPowerSpectrumTemplate, FixedPowerSpectrumTemplate, StandardPowerSpectrumTemplate,
                                                 KaiserTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles)
from desilike.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
from desilike.likelihoods import ObservablesGaussianLikelihood
from desilike.base import vmap


def prior_spec(param):
    prior = param.prior
    spec = dict(dist=prior.dist, lo=float(prior.limits[0]), hi=float(prior.limits[1]), loc=0., scale=1.)
    if prior.dist == 'norm':
        spec.update(loc=float(prior.attrs['loc']), scale=float(prior.attrs['scale']))
    return spec


def extract_observable(obs):
    """Constants of one TracerPowerSpectrumMultipolesObservable, read off the initialised reference objects."""
    wm = obs.wmatrix
    theory = wm.theory
    pt = theory.pt
    template = pt.template
    c = {}
    c['ells'] = np.array(wm.ells)
    c['ellsin'] = np.array(wm.ellsin)
    c['kin'] = np.asarray(pt.k)
    c['kout'] = np.concatenate(wm.k)
    c['mu'], c['wmu_ell'] = np.asarray(pt.mu), np.asarray(pt.wmu)
    c['k11'] = np.asarray(template.k)
    c['pk_dd_fid'] = np.asarray(template.pk_dd_fid)
    if getattr(template, 'with_now', False):
        c['pknow_dd_fid'] = np.asarray(template.pknow_dd_fid)
    c['f_fid'] = float(template.f_fid)
    c['template'] = template.__class__.__name__
    if hasattr(template, 'kp') and hasattr(template, 'a'):      # ShapeFit (the band template's ``kp`` are its pivots)
        c['kp'], c['a'] = template.kp, template.a
    c['nd'] = theory.nd
    if wm.matrix_full is not None: c['matrix_full'] = np.asarray(wm.matrix_full)
    if wm.offset is not None: c['offset'] = np.asarray(wm.offset)
    if wm.kmask is not None: c['kmask'] = np.asarray(wm.kmask)
    c['shotnoisein'], c['shotnoiseout'] = np.asarray(wm.shotnoisein), np.asarray(wm.shotnoiseout)
    c['flatdata'] = np.asarray(obs.flatdata)
    if hasattr(theory, 'counterterm_matrix'):
        c['ct_matrix'] = np.asarray(theory.counterterm_matrix)
        c['sn_matrix'] = np.asarray(theory.stochastic_matrix)
        c['ct_params'] = np.array(theory.counterterm_params)
        c['sn_params'] = np.array(theory.stochastic_params)
    return c


def run_batch(likelihood, observables, theta, names, nint=8):
    """Reference evaluation: vmap over the batch (base.py:232-258) + per-point intermediates for the first nint rows."""
    vlike = vmap(likelihood, backend=None, errors='return', return_derived=True)
    t0 = time.time()
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    dt = time.time() - t0
    out = {'logposterior': np.asarray(logpost),
           'loglikelihood': np.asarray(derived[likelihood._param_loglikelihood]),
           'logprior': np.asarray(derived[likelihood._param_logprior]),
           'ref_seconds_per_eval': dt / len(theta), 'nerrors': len(errors)}
    inter = {name: [] for name in ['pk_dd_template', 'pk_dd', 'pk_dt', 'pk_tt', 'power', 'flatpower', 'flatdiff']}
    flattheory = []
    for row in theta:
        likelihood(**dict(zip(names, row)))
        flattheory.append(np.asarray(likelihood.flattheory))
    for row in theta[:nint]:
        likelihood(**dict(zip(names, row)))
        for iobs, obs in enumerate(observables):
            pt = obs.wmatrix.theory.pt
            inter['pk_dd_template'].append(np.asarray(pt.template.pk_dd))
            for name in ['pk_dd', 'pk_dt', 'pk_tt']:
                inter[name].append(np.asarray(pt.pktable[name]))
            inter['power'].append(np.asarray(obs.wmatrix.theory.power))
            inter['flatpower'].append(np.asarray(obs.wmatrix.flatpower))
        inter['flatdiff'].append(np.asarray(likelihood.flatdiff))
    out['flattheory'] = np.array(flattheory)
    nobs = len(observables)
    for name, value in inter.items():
        value = np.array(value)
        if nint == 0: continue
        if name != 'flatdiff':
            value = value.reshape((min(nint, len(theta)), nobs) + value.shape[1:])
        out['int_' + name] = value
    return out


def sample_theta(likelihood, size, seed):
    """theta ~ Parameter.ref (samplers/base.py:222-230), RandomState(seed)."""
    rng = np.random.RandomState(seed)
    cols = []
    for param in likelihood.varied_params:
        if param.ref.is_proper():
            cols.append(param.ref.sample(size=size, random_state=rng))
        else:
            cols.append(np.full(size, param.value))
    return np.column_stack(cols)


def save(name, **arrays):
    flat = {}
    for key, value in arrays.items():
        if isinstance(value, dict):
            for k, v in value.items():
                flat['{}.{}'.format(key, k)] = v
        else:
            flat[key] = value
    fn = os.path.join(here, name + '.npz')
    np.savez_compressed(fn, **flat)
    print('saved', fn, '{:.1f} kB'.format(os.path.getsize(fn) / 1e3))


def dense_window(kedges, ells, resolution=10, seed=7):
    """Synthetic survey-like dense window: bininteg (x) smooth Gaussian mixing kernel + 5 % ell-leakage (SURVEY 8d cfg 2)."""
    from oracle.np_oracle import window_matrix_bininteg
    edges = np.column_stack([kedges[:-1], kedges[1:]])
    kin, binmat = window_matrix_bininteg([edges] * len(ells), resolution=resolution)
    binmat = binmat.T  # [n_out, n_in]
    nin = kin.size
    dk = kin[:, None] - kin[None, :]
    smooth = np.exp(-0.5 * (dk / 0.004)**2)
    smooth /= smooth.sum(axis=1)[:, None]
    nl = len(ells)
    mix = np.zeros((nl * nin, nl * nin))
    for i in range(nl):
        for j in range(nl):
            mix[i * nin:(i + 1) * nin, j * nin:(j + 1) * nin] = smooth * (1. if i == j else 0.05 / (1 + abs(i - j)))
    rng = np.random.RandomState(seed)
    wmat = binmat.dot(mix) * (1. + 0.01 * rng.standard_normal((binmat.shape[0], mix.shape[1])))
    return kin, wmat


def spd_covariance(n, seed=0, diag=1e4, amp=30.):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal((n, n)) * amp
    return A.dot(A.T) + diag * np.eye(n)


def special_rows(theta, names):
    """Append rows outside the prior and with NaN (SURVEY 8a row a14 conventions)."""
    theta = theta.copy()
    iq, ib = names.index('qpar'), names.index('b1')
    theta[-1, iq] = 1.3       # outside uniform prior [0.8, 1.2] -> logprior = -inf
    theta[-2, ib] = -0.5      # outside [0, 4]
    theta[-3, ib] = 4.        # exactly on the (closed) upper limit -> finite
    return theta


def cfg1():
    """BASELINE config 1: Kaiser ell=(0, 2), 40 k-bins, no window, Gaussian likelihood, single evals."""
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2), theory=theory, shotnoise=1e4)
    cov = spd_covariance(80, seed=0)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    assert abs(like(b1=2.)) < 1e-20
    names = like.varied_params.names()
    theta = special_rows(sample_theta(like, 16, seed=42), names)
    out = run_batch(like, [obs], theta, names)
    save('cfg1_kaiser_nowindow', names=np.array(names), theta=theta, obs0=extract_observable(obs), precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]), **out)


def cfg2(dense=False):
    """BASELINE config 2: ShapeFit + Kaiser ell=(0, 2, 4), 40 k-bins, window (binning res=10, or dense synthetic), 64 points."""
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    kedges = np.linspace(0., 0.2, 41)
    if dense:
        kin, wmat = dense_window(kedges, (0, 2, 4))
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix=wmat, kin=kin, ellsin=(0, 2, 4), theory=theory, shotnoise=1e4)
    else:
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=1e4)
    cov = spd_covariance(120, seed=1)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    assert abs(like(b1=2.)) < 1e-18, like(b1=2.)
    names = like.varied_params.names()
    theta = special_rows(sample_theta(like, 64, seed=42), names)
    out = run_batch(like, [obs], theta, names)
    print('cfg2 dense={} reference: {:.1f} evals/s'.format(dense, 1. / out['ref_seconds_per_eval']))
    save('cfg2_shapefit_window' + ('_dense' if dense else ''), names=np.array(names), theta=theta, obs0=extract_observable(obs), precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]), **out)


def cfg2_variants():
    """Damping on (sigmapar, sigmaper varied), dn varied, diagonal precision, qisoqap AP mode, EFT-like terms."""
    template = ShapeFitPowerSpectrumTemplate(z=0.5, apmode='qisoqap')
    template.init.params['dn'].update(fixed=False)
    theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=template)
    for name in ['sigmapar', 'sigmaper']:
        theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=4., scale=0.5))
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmapar': 4., 'sigmaper': 3.}, kedges=np.linspace(0.01, 0.2, 39), ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=np.diag(1. / np.linspace(1e-4, 3e-4, 114)))
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 24, seed=3)
    out = run_batch(like, [obs], theta, names)
    save('cfg2v_eft_damping_qisoqap', names=np.array(names), theta=theta, obs0=extract_observable(obs), precision=np.asarray(like.precision),
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]), **out)




def marg_grid():
    """Pin for analytic marginalisation (the reference's own implementation needs jax: not runnable here).

    The reference's NON-marginalised likelihood + prior is evaluated on a fine grid of the linear parameter sn0
    (all other parameters fixed at a few points): the Gaussian integral over sn0 must equal the analytically
    marginalised posterior up to the (2 pi)^(1/2) the reference drops (likelihoods/base.py:396-401).
    Also stores d(flattheory)/d(sn0) by finite difference of the reference (exactly linear in sn0)."""
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    theory.init.params['sn0'].update(prior=dict(dist='norm', loc=0.2, scale=1.5))
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sn0': 0.4}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=1e4)
    cov = spd_covariance(120, seed=1)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 6, seed=11)
    isn0 = names.index('sn0')
    grid = np.linspace(-4., 4.8, 881)
    logpost = np.empty((len(theta), grid.size))
    flat0, flat1 = [], []
    for ip, row in enumerate(theta):
        for ig, sn0 in enumerate(grid):
            row2 = row.copy(); row2[isn0] = sn0
            logpost[ip, ig] = like(**dict(zip(names, row2)))
        row2 = row.copy(); row2[isn0] = 0.
        like(**dict(zip(names, row2))); flat0.append(np.asarray(like.flattheory).copy())
        row2[isn0] = 1.
        like(**dict(zip(names, row2))); flat1.append(np.asarray(like.flattheory).copy())
    # logprior of the non-solved parameters only (to compare with the marginalised likelihood's own logprior)
    logprior_others = np.array([like.all_params.prior(**{name: value for name, value in zip(names, row) if name != 'sn0'}) for row in theta])
    save('marg_sn0_grid', names=np.array(names), theta=theta, grid=grid, logposterior_grid=logpost, flattheory_sn0_0=np.array(flat0), flattheory_sn0_1=np.array(flat1),
         logprior_others=logprior_others, sn0_prior=np.array([0.2, 1.5]), obs0=extract_observable(obs), precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]))


def marg_multi():
    """Multi-parameter pin of analytic marginalisation (likelihoods/base.py:314-413; the reference's own implementation needs jax): the reference's NON-marginalised
    log-posterior is an exact quadratic form of the linear parameters x,  log p(x) = c + g.(x - x0) + 1/2 (x - x0)^T H (x - x0)  (H includes the Gaussian priors of x).
    It is evaluated by the reference on the stencil {x0, x0 +- d_i e_i, x0 + d_i e_i + d_j e_j}; (c, g, H) follow exactly and with them the closed forms
        x* = x0 - H^-1 g,   logposterior(.best) = c - 1/2 g H^-1 g,   logposterior(.marg over M) = that - 1/2 logdet(-H[M, M])      (reference convention 394-404: no 2 pi).
    Cases: (a) EFT-like Kaiser with two counter terms (derivative rows depend on the point) and a stochastic term, (b) two tracers with both shot-noise terms."""
    def quadratic_form(like, names, row, solved, x0, steps):
        base = dict(zip(names, row))

        def logpost(x):
            return like(**{**base, **dict(zip(solved, x))})

        ns = len(solved)
        f0 = logpost(x0)
        g, H = np.zeros(ns), np.zeros((ns, ns))
        fp, fm = np.zeros(ns), np.zeros(ns)
        for i in range(ns):
            e = np.zeros(ns); e[i] = steps[i]
            fp[i], fm[i] = logpost(x0 + e), logpost(x0 - e)
            g[i] = (fp[i] - fm[i]) / (2. * steps[i])
            H[i, i] = (fp[i] - 2. * f0 + fm[i]) / steps[i]**2
        for i in range(ns):
            for j in range(i + 1, ns):
                e = np.zeros(ns); e[i], e[j] = steps[i], steps[j]
                fpp = logpost(x0 + e)
                H[i, j] = H[j, i] = (fpp - fp[i] - fp[j] + f0) / (steps[i] * steps[j])
        # exactness check: an independent point
        e = 0.37 * np.asarray(steps) * (1. + np.arange(ns))
        assert abs(logpost(x0 + e) - (f0 + g.dot(e) + 0.5 * e.dot(H).dot(e))) < 1e-8 * max(1., abs(f0)), 'the reference posterior is not quadratic in the solved parameters'
        return f0, g, H

    out = {}
    # (a) EFT-like Kaiser: ct0_2, ct2_2 (point-dependent derivative rows), sn0_2
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = EFTLikeKaiserTracerPowerSpectrumMultipoles(template=template)
    theory.init.params['ct0_2'].update(prior=dict(dist='norm', loc=0., scale=30.))
    theory.init.params['ct2_2'].update(prior=dict(dist='norm', loc=1., scale=50.))
    theory.init.params['sn0_2'].update(prior=dict(dist='norm', loc=0., scale=5.))
    for name in ['ct4_2', 'sn2_2', 'sn4_2']:
        theory.init.params[name].update(fixed=True, value=0.)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 1.9, 'ct0_2': 3., 'sn0_2': 0.5}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    cov = spd_covariance(120, seed=21)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    solved = ['ct0_2', 'ct2_2', 'sn0_2']
    others = [name for name in names if name not in solved]
    theta = sample_theta(like, 5, seed=23)
    x0 = np.array([like.all_params[name].value for name in solved])
    rows = []
    for row in theta:
        rows.append(quadratic_form(like, names, row, solved, x0, steps=[10., 15., 2.]))
    out['a'] = dict(names=np.array(others), solved=np.array(solved), theta=theta[:, [names.index(name) for name in others]], x0=x0,
                    c=np.array([r[0] for r in rows]), g=np.array([r[1] for r in rows]), H=np.array([r[2] for r in rows]),
                    prior=np.array([[0., 30.], [1., 50.], [0., 5.]]), flatdata=np.asarray(obs.flatdata), covariance=cov)
    # (b) two tracers, LRG.sn0 and ELG.sn0
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    observables = []
    for tracer, b1, kmax in [('LRG', 2., 0.2), ('ELG', 1.3, 0.15)]:
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        theory.init.params[tracer + '.sn0'].update(prior=dict(dist='norm', loc=0.1, scale=2.))
        nk = int(round(kmax / 0.005))
        observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1, tracer + '.sn0': 0.3}, kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 4},
                                                                   theory=theory, shotnoise=1e4 if tracer == 'LRG' else 4e3))
    cov = spd_covariance(210, seed=22)
    like = ObservablesGaussianLikelihood(observables=observables, covariance=cov)
    like()
    names = like.varied_params.names()
    solved = ['LRG.sn0', 'ELG.sn0']
    others = [name for name in names if name not in solved]
    theta = sample_theta(like, 5, seed=24)
    x0 = np.array([like.all_params[name].value for name in solved])
    rows = [quadratic_form(like, names, row, solved, x0, steps=[1.5, 1.5]) for row in theta]
    out['b'] = dict(names=np.array(others), solved=np.array(solved), theta=theta[:, [names.index(name) for name in others]], x0=x0,
                    c=np.array([r[0] for r in rows]), g=np.array([r[1] for r in rows]), H=np.array([r[2] for r in rows]), prior=np.array([[0.1, 2.], [0.1, 2.]]),
                    flatdata0=np.asarray(observables[0].flatdata), flatdata1=np.asarray(observables[1].flatdata), covariance=cov)
    save('marg_multi', **out)


def cfg5():
    """BASELINE config 5 (one walker batch): two tracers, one ObservablesGaussianLikelihood with a joint covariance, shared ShapeFit template,
    per-tracer b1 / sn0 namespaces (full_shape.py:59-133; likelihoods/base.py:567, 662-664)."""
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    observables = []
    for tracer, b1, kmax in [('LRG', 2., 0.2), ('ELG', 1.3, 0.15)]:
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        nk = int(round(kmax / 0.005))
        observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1}, kedges=np.linspace(0., kmax, nk + 1), ells=(0, 2, 4), wmatrix={'resolution': 4},
                                                                   theory=theory, shotnoise=1e4 if tracer == 'LRG' else 4e3))
    sizes = [3 * 40, 3 * 30]
    cov = spd_covariance(sum(sizes), seed=5)
    like = ObservablesGaussianLikelihood(observables=observables, covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 32, seed=21)
    out = run_batch(like, observables, theta, names, nint=0)
    obs = {'obs{:d}'.format(i): {**extract_observable(o), 'tracer': ['LRG', 'ELG'][i]} for i, o in enumerate(observables)}
    save('cfg5_two_tracers', names=np.array(names), theta=theta, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         **obs, **{k: v for k, v in out.items() if not k.startswith('int_')})


def cfg5_bench():
    """The shape bench.py times for BASELINE configs[4] (``bench.make_likelihood_config5``): two config-2 tracers, each with the dense synthetic 120 x 1200 window
    (seeds 7 / 8), shared ShapeFit template, per-tracer b1 / sn0 namespaces, block-diagonal joint covariance (n = 240, 8 varied parameters), run by the REFERENCE
    (full_shape.py:59-133; likelihoods/base.py:567, 617-619, 662-664).  The big inputs (windows 2 x 1.15 MB, covariance 0.46 MB) are regenerated by the test from the
    same seeds: the fixture keeps their checksums (sum, sum of squares, a strided sample) so that a drift of the generators is caught, plus the template tables, flatdata and
    every output of the reference.  theta: 64 draws of Parameter.ref (what the sampler starts from) + 61 draws of a five times wider cloud (where walkers wander) + the
    special rows (outside the prior, on the closed limit)."""
    from scipy import linalg
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    kedges = np.linspace(0., 0.2, 41)
    observables, windows = [], []
    for tracer, b1, shotnoise, seed in [('LRG', 2., 1e4, 7), ('ELG', 1.3, 4e3, 8)]:
        kin, wmat = dense_window(kedges, (0, 2, 4), seed=seed)
        windows.append(wmat)
        theory = KaiserTracerPowerSpectrumMultipoles(template=template, tracers=tracer)
        observables.append(TracerPowerSpectrumMultipolesObservable(data={tracer + '.b1': b1}, kedges=kedges, ells=(0, 2, 4), wmatrix=wmat, kin=kin, ellsin=(0, 2, 4), theory=theory, shotnoise=shotnoise))
    cov = linalg.block_diag(spd_covariance(120, seed=1), spd_covariance(120, seed=2))
    like = ObservablesGaussianLikelihood(observables=observables, covariance=cov)
    assert abs(like(**{'LRG.b1': 2., 'ELG.b1': 1.3})) < 1e-16
    names = like.varied_params.names()
    assert sorted(names) == sorted(['qpar', 'qper', 'dm', 'df', 'LRG.b1', 'LRG.sn0', 'ELG.b1', 'ELG.sn0']), names   # (the reference's own order: the test maps columns by name)
    narrow = sample_theta(like, 64, seed=42)
    centre = np.array([param.value for param in like.varied_params])
    wide = centre + 5. * (sample_theta(like, 64, seed=43) - centre)
    theta = np.concatenate([narrow, wide])
    iq, ib = names.index('qpar'), names.index('ELG.b1')
    theta[-1, iq] = 1.3; theta[-2, ib] = -0.5; theta[-3, ib] = 4.
    out = run_batch(like, observables, theta, names, nint=0)
    print('cfg5_bench reference: {:.1f} evals/s'.format(1. / out['ref_seconds_per_eval']))

    def checksum(a):
        a = np.asarray(a, dtype='f8')
        return np.concatenate([[a.sum(), (a**2).sum()], a.ravel()[::997][:64]])

    tmpl = observables[0].wmatrix.theory.pt.template
    precision = np.asarray(like.precision)
    save('cfg5_bench', names=np.array(names), theta=theta, k11=np.asarray(tmpl.k), pk_dd_fid=np.asarray(tmpl.pk_dd_fid), f_fid=np.array(float(tmpl.f_fid)),
         flatdata=np.concatenate([np.asarray(o.flatdata) for o in observables]), window_checksum=np.array([checksum(w) for w in windows]), covariance_checksum=checksum(cov),
         precision_checksum=checksum(precision), window_seeds=np.array([7, 8]), covariance_seeds=np.array([1, 2]),
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         **{k: v for k, v in out.items() if not k.startswith('int_')})


def extract_bao_observable(obs, space='xi'):
    wm = obs.wmatrix
    theory = wm.theory
    pt = theory.pt if space == 'pk' else theory.power
    template = pt.template
    c = {'ells': np.array(wm.ells), 'ellsin': np.array(wm.ellsin), 'kin': np.asarray(pt.k), 'mu': np.asarray(pt.mu), 'wmu_ell': np.asarray(pt.wmu),
         'k11': np.asarray(template.k), 'pk_dd_fid': np.asarray(template.pk_dd_fid), 'pknow_dd_fid': np.asarray(template.pknow_dd_fid), 'f_fid': float(template.f_fid),
         'mode': pt.mode, 'model': pt.model, 'smoothing_radius': pt.smoothing_radius, 'flatdata': np.asarray(obs.flatdata), 'broadband': theory.broadband}
    names = [name for ell in theory.ells for name in theory.broadband_orders[ell]]
    c['broadband_params'] = np.array(names)
    if space == 'xi':
        c['s'] = np.asarray(theory.s); c['sp'] = theory.sp; c['sout'] = np.concatenate(wm.s)
        bb = np.zeros((len(theory.ells), len(theory.s), len(names)))
    else:
        c['kp'] = theory.kp; c['kout'] = np.concatenate(wm.k)
        bb = np.zeros((len(theory.ells), len(theory.k), len(names)))
        c['shotnoisein'], c['shotnoiseout'] = np.asarray(wm.shotnoisein), np.asarray(wm.shotnoiseout)
    for ill, ell in enumerate(theory.ells):
        for name, row in zip(theory.broadband_orders[ell], np.asarray(theory.broadband_matrix[ell])):
            bb[ill, :, names.index(name)] = row
    c['broadband_matrix'] = bb
    if wm.matrix_full is not None: c['matrix_full'] = np.asarray(wm.matrix_full)
    if getattr(wm, 'smask', None) is not None: c['smask'] = np.asarray(wm.smask)
    return c


def cfg4(space='xi'):
    """BASELINE config 4: Damped-BAO xi_ell (ell = 0, 2; 30 s-bins) via FFTLog + Gaussian likelihood; space='pk': the P_ell version with a binning window."""
    from desilike.theories.galaxy_clustering import BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles, DampedBAOWigglesTracerPowerSpectrumMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    template = BAOPowerSpectrumTemplate(z=0.5)
    if space == 'xi':
        theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='reciso')
        obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
        n, scale = 60, 3e-4
    else:
        theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template)
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
        n, scale = 112, 30.
    for name in ['sigmapar', 'sigmaper']:
        theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
    rng = np.random.RandomState(4)
    A = rng.standard_normal((n, n)) * scale
    cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 24, seed=9)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    t0 = time.time()
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    dt = (time.time() - t0) / len(theta)
    print('cfg4 {} reference: {:.1f} evals/s'.format(space, 1. / dt))
    power, corr, flat = [], [], []
    for row in theta:
        like(**dict(zip(names, row)))
        pt = theory.pt if space == 'pk' else theory.power
        power.append(np.asarray(pt.power).copy())
        corr.append(np.asarray(theory.corr if space == 'xi' else theory.power).copy())
        flat.append(np.asarray(like.flattheory).copy())
    save('cfg4_bao_' + space, names=np.array(names), theta=theta, obs0=extract_bao_observable(obs, space=space), precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         wiggle_power=np.array(power), theory=np.array(corr), flattheory=np.array(flat), ref_seconds_per_eval=dt)


def cfg3_table():
    """Velocileptors-style table combination (full_shape.py:1182-1186, 1573-1599, 1479-1488) run by the REFERENCE on a stand-in PT node.

    velocileptors itself is an external CPU code (absent, out of scope): the PT calculator is replaced by a subclass that keeps the reference's
    ``combine_bias_terms_poles`` but produces a synthetic ``pktable`` that is an exact quadratic polynomial of the template parameters
    (so that a second-order Taylor emulator reproduces it exactly), with sigma8, fsigma8 varying too."""
    from desilike.theories.galaxy_clustering.full_shape import REPTVelocileptorsPowerSpectrumMultipoles, REPTVelocileptorsTracerPowerSpectrumMultipoles

    ells = (0, 2, 4)

    def make_tables(kpt):
        base = 2e4 * (kpt / 0.05)**0.96 / (1. + (kpt / 0.02)**2.5)
        tables = np.array([[[base * amp * (1. + 0.3 * np.sin(3. * m + ell + 20. * kpt * (1 + t))) for m in range(19)] for ell in ells] for t, amp in enumerate([1., 0.3, -0.2, 0.15, 0.1, 0.05])])
        tables = np.moveaxis(tables, 2, -1)                        # [6, n_ell, n_kpt, 19]
        tables[..., 16:] = 0.
        for ill in range(3): tables[0, ill, :, 16 + ill] = kpt**(2 * ill) if ill else 1.   # stochastic monomials: 1, k^2-like, k^4-like
        return tables

    class FakePT(REPTVelocileptorsPowerSpectrumMultipoles):

        _params = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
                   'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}

        def initialize(self, k=None, ells=(0, 2, 4), **kwargs):
            self.k = np.linspace(0.01, 0.2, 101) if k is None else np.asarray(k, dtype='f8')
            self.ells = tuple(ells)
            self.tables = make_tables(self.k)
            self.z = np.array(0.8)
            self.options = {}

        def calculate(self, qpar=1., qper=1., dm=0.):
            x = [qpar - 1., qper - 1., dm]
            tables = self.tables
            self.pktable = tables[0] + x[0] * tables[1] + x[1] * tables[2] + x[2] * tables[3] + x[0] * x[2] * tables[4] + x[1]**2 * tables[5]
            self.sigma8 = 0.8 * (1. + 0.2 * dm + 0.1 * x[0]**2)
            self.fsigma8 = 0.45 * (1. + 0.3 * x[1] - 0.1 * dm)

    pt = FakePT()
    theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='LRG')
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1p': 1.6, 'b2p': 0.3, 'alpha0p': 2.}, kedges=np.linspace(0.02, 0.2, 37), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory, shotnoise=8e3)
    cov = spd_covariance(108, seed=8, diag=4e4, amp=40.)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 24, seed=17)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    assert not errors, errors
    power, flat = [], []
    for row in theta:
        like(**dict(zip(names, row)))
        power.append(np.asarray(theory.power).copy()); flat.append(np.asarray(like.flattheory).copy())
    wm = obs.wmatrix
    obs0 = dict(kpt=np.asarray(pt.k), k=np.asarray(theory.k), ells=np.array(theory.ells), tables=pt.tables, nd=theory.nd, snd=theory.snd, fsat=theory.fsat, sigv=theory.options['sigv'],
                matrix_full=np.asarray(wm.matrix_full), shotnoisein=np.asarray(wm.shotnoisein), shotnoiseout=np.asarray(wm.shotnoiseout), flatdata=np.asarray(obs.flatdata))
    save('cfg3_velocileptors_table', names=np.array(names), theta=theta, obs0=obs0, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         power=np.array(power), flattheory=np.array(flat))
    print(names)


def cfg3_full():
    """BASELINE configs[2] at the size SURVEY.md section 8d states (MLP in = 6, 4 x 64 silu, 3 * 128 * 19 = 7296 outputs; 19-monomial combination; cubic
    interpolation to n_kin = 400; binning window 120 x 1200): the REFERENCE's REPT velocileptors tracer + window + Gaussian likelihood on a stand-in PT node whose
    ``pktable`` / ``sigma8`` / ``fsigma8`` come from the MLPs of tests/emulator_utils.cfg3_full_engines (forward pass = the oracle's restatement of
    emulators/conversion.py:20-96; the engine itself is third-party and absent: unpinned).  Pins table combination -> interpolation -> window -> chi2 at full size."""
    from desilike.theories.galaxy_clustering.full_shape import REPTVelocileptorsPowerSpectrumMultipoles, REPTVelocileptorsTracerPowerSpectrumMultipoles
    sys.path.insert(0, os.path.join(os.path.dirname(here)))
    from emulator_utils import CFG3_PARAMS, CFG3_SPECS, cfg3_full_kpt, cfg3_full_engines
    from oracle import np_oracle as orc
    engines = cfg3_full_engines()

    def predict(name, x):
        e = engines[name]
        return orc.mlp_predict(x, e['xlimits'], e['layers'], 'silu', e['ylimits']).reshape(e['yshape'])

    class FakePT(REPTVelocileptorsPowerSpectrumMultipoles):

        _params = {name: dict(spec) for name, spec in CFG3_SPECS.items()}

        def initialize(self, k=None, ells=(0, 2, 4), **kwargs):
            self.k = cfg3_full_kpt()
            self.ells = tuple(ells)
            self.z = np.array(0.8)
            self.options = {}

        def calculate(self, **params):
            x = np.array([params[name] for name in CFG3_PARAMS])
            self.pktable = predict('pktable', x)
            self.sigma8, self.fsigma8 = float(predict('sigma8', x)[0]), float(predict('fsigma8', x)[0])

    pt = FakePT()
    theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='LRG')
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1p': 1.6, 'b2p': 0.3, 'alpha0p': 2.}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=8e3)
    cov = spd_covariance(120, seed=9, diag=4e4, amp=40.)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 16, seed=19)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    assert not errors, errors
    power, flat = [], []
    for row in theta[:4]:
        like(**dict(zip(names, row)))
        power.append(np.asarray(theory.power).copy()); flat.append(np.asarray(like.flattheory).copy())
    wm = obs.wmatrix
    assert wm.matrix_full.shape == (120, 1200)
    obs0 = dict(k=np.asarray(theory.k), ells=np.array(theory.ells), nd=theory.nd, snd=theory.snd, fsat=theory.fsat, sigv=theory.options['sigv'],
                shotnoisein=np.asarray(wm.shotnoisein), shotnoiseout=np.asarray(wm.shotnoiseout), flatdata=np.asarray(obs.flatdata))
    save('cfg3_full', names=np.array(names), theta=theta, obs0=obs0, cov_seed=np.array([9]),
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         power=np.array(power), flattheory=np.array(flat))
    print(names)


def cfg3_table_xi():
    """REPT velocileptors correlation function multipoles (full_shape.py:1603-1629): the table combination of cfg3_table on the 300-point log grid of get_corr,
    Hankel-transformed (tgc/base.py:127-136), run by the reference on the same kind of stand-in PT node."""
    from desilike.theories.galaxy_clustering.full_shape import REPTVelocileptorsPowerSpectrumMultipoles, REPTVelocileptorsTracerCorrelationFunctionMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable

    ells = (0, 2, 4)

    def make_tables(kpt):
        base = 2e4 * (kpt / 0.05)**0.96 / (1. + (kpt / 0.02)**2.5) * np.exp(-(kpt / 0.4)**2)
        tables = np.array([[[base * amp * (1. + 0.3 * np.sin(3. * m + ell + 20. * np.minimum(kpt, 0.3) * (1 + t))) for m in range(19)] for ell in ells] for t, amp in enumerate([1., 0.3, -0.2, 0.15, 0.1, 0.05])])
        tables = np.moveaxis(tables, 2, -1)                        # [6, n_ell, n_kpt, 19]
        tables[..., 16:] = 0.
        return tables

    class FakePT(REPTVelocileptorsPowerSpectrumMultipoles):

        _params = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
                   'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}

        def initialize(self, k=None, ells=(0, 2, 4), **kwargs):
            self.k = np.linspace(0.01, 0.2, 101) if k is None else np.asarray(k, dtype='f8')
            self.ells = tuple(ells)
            self.tables = make_tables(self.k)
            self.z = np.array(0.8)
            self.options = {}

        def calculate(self, qpar=1., qper=1., dm=0.):
            x = [qpar - 1., qper - 1., dm]
            tables = self.tables
            self.pktable = tables[0] + x[0] * tables[1] + x[1] * tables[2] + x[2] * tables[3] + x[0] * x[2] * tables[4] + x[1]**2 * tables[5]
            self.sigma8 = 0.8 * (1. + 0.2 * dm + 0.1 * x[0]**2)
            self.fsigma8 = 0.45 * (1. + 0.3 * x[1] - 0.1 * dm)

    pt = FakePT()
    theory = REPTVelocileptorsTracerCorrelationFunctionMultipoles(pt=pt, tracer='LRG')
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1p': 1.6, 'b2p': 0.3, 'alpha0p': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory)
    n, scale = 90, 3e-4
    rng = np.random.RandomState(24)
    A = rng.standard_normal((n, n)) * scale
    cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 16, seed=27)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    assert not errors, errors
    power, corr, flat = [], [], []
    for row in theta:
        like(**dict(zip(names, row)))
        power.append(np.asarray(theory.power.power).copy()); corr.append(np.asarray(theory.corr).copy()); flat.append(np.asarray(like.flattheory).copy())
    pw = theory.power
    obs0 = dict(kpt=np.asarray(pt.k), k=np.asarray(pw.k), ells=np.array(pw.ells), tables=pt.tables, nd=pw.nd, snd=pw.snd, fsat=pw.fsat, sigv=pw.options['sigv'], s=np.asarray(theory.s),
                flatdata=np.asarray(obs.flatdata))
    save('cfg3_velocileptors_table_xi', names=np.array(names), theta=theta, obs0=obs0, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         power=np.array(power), theory=np.array(corr), flattheory=np.array(flat))
    print(names)


def syst_template_0(ell, k):
    return 1e3 * (ell == 0) / (1. + (k / 0.02)**2)


def cfg4_kernel_broadband(space='xi'):
    """Damped BAO with the kernel broadbands (bao.py:468-523, 833-905): 'pcs' for P_ell (cubic B-spline nodes every kp, scaled by the no-wiggle power), 'pcs2' for
    xi_ell (the same kernels Hankel-transformed with the multipoles + powers of s: bl*)."""
    from desilike.theories.galaxy_clustering import BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles, DampedBAOWigglesTracerPowerSpectrumMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    template = BAOPowerSpectrumTemplate(z=0.5)
    if space == 'xi':
        theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode='recsym', broadband='pcs2')
        obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
        n, scale = 60, 3e-4
    else:
        theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template, broadband='pcs')
        obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
        n, scale = 112, 30.
    rng = np.random.RandomState(4)
    A = rng.standard_normal((n, n)) * scale
    cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 16, seed=39)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    assert not errors
    power, corr, flat = [], [], []
    for row in theta:
        like(**dict(zip(names, row)))
        pt = theory.pt
        power.append(np.asarray(pt.power).copy())
        corr.append(np.asarray(theory.corr if space == 'xi' else theory.power).copy())
        flat.append(np.asarray(like.flattheory).copy())
    ptheory = theory.power if space == 'xi' else theory      # the tracer power spectrum class carrying the Fourier-space kernels
    knames = [name for ell in ptheory.ells for name in ptheory.broadband_orders[ell]]
    kmat = np.zeros((len(ptheory.ells), len(ptheory.k), len(knames)))
    for ill, ell in enumerate(ptheory.ells):
        for name, row in zip(ptheory.broadband_orders[ell], np.asarray(ptheory.broadband_matrix[ell])):
            kmat[ill, :, knames.index(name)] = row
    c = {'flatdata': np.asarray(obs.flatdata), 'kin': np.asarray(ptheory.k), 'kp': ptheory.kp, 'kernel_params': np.array(knames), 'kernel_matrix': kmat}
    if space == 'xi':
        snames = [name for ell in theory.ells for name in theory.broadband_orders[ell]]
        smat = np.zeros((len(theory.ells), len(theory.s), len(snames)))
        for ill, ell in enumerate(theory.ells):
            for name, row in zip(theory.broadband_orders[ell], np.asarray(theory.broadband_matrix[ell])):
                smat[ill, :, snames.index(name)] = row
        c.update(s=np.asarray(theory.s), sp=theory.sp, s_params=np.array(snames), s_matrix=smat)
    else:
        c.update(matrix_full=np.asarray(obs.wmatrix.matrix_full), shotnoisein=np.asarray(obs.wmatrix.shotnoisein), shotnoiseout=np.asarray(obs.wmatrix.shotnoiseout))
    save('cfg4_bao_' + space + '_pcs', names=np.array(names), theta=theta, obs0=c, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         wiggle_power=np.array(power), theory=np.array(corr), flattheory=np.array(flat))
    print(names)


def cfg4_models():
    """The non-standard wiggle models of DampedBAOWigglesPowerSpectrumMultipoles (bao.py:137-150): Beutler 2016 ('fog-damping_move-all') for P_ell, 'fix-damping' with
    reciso for xi_ell, and the Howlett 2023 form ('move-all')."""
    from desilike.theories.galaxy_clustering import BAOPowerSpectrumTemplate, DampedBAOWigglesTracerCorrelationFunctionMultipoles, DampedBAOWigglesTracerPowerSpectrumMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    out = {}
    for tag, space, model, mode in [('a', 'pk', 'fog-damping_move-all', ''), ('b', 'xi', 'fix-damping', 'reciso'), ('c', 'pk', 'move-all', 'reciso')]:
        template = BAOPowerSpectrumTemplate(z=0.5)
        if space == 'xi':
            theory = DampedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode=mode, model=model)
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        else:
            theory = DampedBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode=mode, model=model)
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
            n, scale = 112, 30.
        for name in ['sigmapar', 'sigmaper']:
            theory.init.params[name].update(fixed=False, ref=dict(dist='norm', loc=8., scale=0.5))
        for param in theory.init.params.select(basename='al*'):
            param.update(fixed=True)
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
        like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
        like()
        names = like.varied_params.names()
        theta = sample_theta(like, 12, seed=49)
        (logpost, derived), errors = vmap(like, backend=None, errors='return', return_derived=True)({name: theta[:, i] for i, name in enumerate(names)})
        assert not errors
        power = []
        for row in theta:
            like(**dict(zip(names, row)))
            power.append(np.asarray(theory.pt.power).copy())
        pt = theory.pt
        tmpl = pt.template
        if tag == 'a':
            out.update(kin_pk=np.asarray(pt.k), mu=np.asarray(pt.mu), wmu_ell=np.asarray(pt.wmu), k11=np.asarray(tmpl.k), pk_dd_fid=np.asarray(tmpl.pk_dd_fid),
                       pknow_dd_fid=np.asarray(tmpl.pknow_dd_fid), f_fid=float(tmpl.f_fid))
        if space == 'xi': out['kin_xi'] = np.asarray(pt.k)
        out.update({tag + '_names': np.array(names), tag + '_theta': theta, tag + '_flatdata': np.asarray(obs.flatdata), tag + '_covariance': cov, tag + '_model': model, tag + '_mode': mode,
                    tag + '_space': space, tag + '_wiggle_power': np.array(power), tag + '_loglikelihood': np.asarray(derived[like._param_loglikelihood]),
                    tag + '_logprior': np.asarray(derived[like._param_logprior])})
    save('cfg4_bao_models', **out)


def cfg4_resummed():
    """Resummed BAO wiggles (bao.py:165-266, 670-717, 1051-1096): P_ell with reciso reconstruction and shot noise, xi_ell with recsym and the Beutler-like smooth part."""
    from desilike.theories.galaxy_clustering import BAOPowerSpectrumTemplate, ResummedBAOWigglesTracerCorrelationFunctionMultipoles, ResummedBAOWigglesTracerPowerSpectrumMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    out = {}
    for tag, space, model, mode in [('a', 'pk', 'standard', 'reciso'), ('b', 'xi', 'fog-damping_move-all', 'recsym'), ('c', 'pk', 'move-all', '')]:
        template = BAOPowerSpectrumTemplate(z=0.5)
        if space == 'xi':
            theory = ResummedBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode=mode, model=model)
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        else:
            theory = ResummedBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode=mode, model=model)
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmas': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=3e3)
            n, scale = 112, 30.
        if space == 'pk': theory.init.params['d'].update(fixed=False)   # (the correlation function class has no 'd' parameter: bao.yaml)
        for param in theory.init.params.select(basename='al*'):
            param.update(fixed=True)
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
        like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
        like()
        names = like.varied_params.names()
        theta = sample_theta(like, 12, seed=59)
        (logpost, derived), errors = vmap(like, backend=None, errors='return', return_derived=True)({name: theta[:, i] for i, name in enumerate(names)})
        assert not errors
        power = []
        for row in theta:
            like(**dict(zip(names, row)))
            power.append(np.asarray(theory.pt.power).copy())
        pt = theory.pt
        tmpl, wig = pt.template, pt.wiggles
        if tag == 'a':
            out.update(mu=np.asarray(pt.mu), wmu_ell=np.asarray(pt.wmu), f_fid=float(tmpl.f_fid), rs_drag=float(tmpl.cosmo.rs_drag))
        out.update({tag + '_kin': np.asarray(pt.k), tag + '_k11': np.asarray(tmpl.k), tag + '_pk_dd_fid': np.asarray(tmpl.pk_dd_fid), tag + '_pknow_dd_fid': np.asarray(tmpl.pknow_dd_fid)})
        out.update({tag + '_names': np.array(names), tag + '_theta': theta, tag + '_flatdata': np.asarray(obs.flatdata), tag + '_covariance': cov, tag + '_model': model, tag + '_mode': mode,
                    tag + '_space': space, tag + '_wiggle_power': np.array(power), tag + '_loglikelihood': np.asarray(derived[like._param_loglikelihood]),
                    tag + '_logprior': np.asarray(derived[like._param_logprior]), tag + '_shotnoise': float(wig.shotnoise),
                    tag + '_sigmas2': np.array([wig.sigma_dd2, wig.sigma_nl2, getattr(wig, 'sigma_x2', 0.), wig.sigma_sn2])})
    save('cfg4_bao_resummed', **out)


def cfg4_flexible():
    """Flexible BAO wiggles (bao.py:269-391, 719-763, 1099-1144): multiplicative terms ml{ell}_{i} on the wiggles ('pcs' nodes or powers of k), no damping."""
    from desilike.theories.galaxy_clustering import BAOPowerSpectrumTemplate, FlexibleBAOWigglesTracerCorrelationFunctionMultipoles, FlexibleBAOWigglesTracerPowerSpectrumMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    out = {}
    for tag, space, model, mode, wiggles in [('a', 'pk', 'standard', 'reciso', 'pcs'), ('b', 'xi', 'move-all', '', 'pcs'), ('c', 'pk', 'standard', '', 'power')]:
        template = BAOPowerSpectrumTemplate(z=0.5)
        if space == 'xi':
            theory = FlexibleBAOWigglesTracerCorrelationFunctionMultipoles(template=template, mode=mode, model=model, wiggles=wiggles)
            obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2), theory=theory)
            n, scale = 60, 3e-4
        else:
            theory = FlexibleBAOWigglesTracerPowerSpectrumMultipoles(template=template, mode=mode, model=model, wiggles=wiggles)
            obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=np.linspace(0.02, 0.3, 57), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory)
            n, scale = 112, 30.
        rng = np.random.RandomState(4)
        A = rng.standard_normal((n, n)) * scale
        cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
        like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
        for param in like.all_params.select(basename='al*'):
            param.update(fixed=True)
        for param in like.all_params.select(basename='ml*'):
            param.update(ref=dict(limits=[-0.3, 0.3]))
        like()
        names = like.varied_params.names()
        theta = sample_theta(like, 10, seed=69)
        (logpost, derived), errors = vmap(like, backend=None, errors='return', return_derived=True)({name: theta[:, i] for i, name in enumerate(names)})
        assert not errors
        power = []
        for row in theta:
            like(**dict(zip(names, row)))
            power.append(np.asarray(theory.pt.power).copy())
        pt = theory.pt
        tmpl = pt.template
        mlnames = [name for ell in pt.ells for name in pt.wiggles_orders[ell]]
        mlmat = np.zeros((len(pt.ells), len(pt.k), len(mlnames)))
        for ill, ell in enumerate(pt.ells):
            for name, row in zip(pt.wiggles_orders[ell], np.asarray(pt.wiggles_matrix[ell])):
                mlmat[ill, :, mlnames.index(name)] = row
        if tag == 'a': out.update(mu=np.asarray(pt.mu), wmu_ell=np.asarray(pt.wmu), f_fid=float(tmpl.f_fid))
        out.update({tag + '_kin': np.asarray(pt.k), tag + '_k11': np.asarray(tmpl.k), tag + '_pk_dd_fid': np.asarray(tmpl.pk_dd_fid), tag + '_pknow_dd_fid': np.asarray(tmpl.pknow_dd_fid),
                    tag + '_names': np.array(names), tag + '_theta': theta, tag + '_flatdata': np.asarray(obs.flatdata), tag + '_covariance': cov, tag + '_model': model, tag + '_mode': mode,
                    tag + '_space': space, tag + '_wiggles': wiggles, tag + '_kp': float(pt.kp), tag + '_ml_names': np.array(mlnames), tag + '_ml_matrix': mlmat,
                    tag + '_wiggle_power': np.array(power), tag + '_loglikelihood': np.asarray(derived[like._param_loglikelihood]), tag + '_logprior': np.asarray(derived[like._param_logprior])})
        print(tag, names)
    save('cfg4_bao_flexible', **out)


def cfg2_fc_syst():
    """Window extras (row a6): top-hat fiber collisions folded into the binning matrix (window.py:428-438, 972-1049) and two systematic templates
    (window.py:439-443, 472-473, 1253-1309), klim row selection on top."""
    from desilike.observables.galaxy_clustering import TopHatFiberCollisionsPowerSpectrumMultipoles
    import desilike.utils as ref_utils

    def weights_trapz(x):
        # the reference's utils.weights_trapz (utils.py:614-622) relies on jnp.insert clamping an out-of-range index (jax is absent here and numpy raises):
        # same weights, written for numpy -- the ONLY line of the reference replaced for this fixture
        x = np.asarray(x)
        return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.

    ref_utils.weights_trapz = weights_trapz
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    theory = KaiserTracerPowerSpectrumMultipoles(template=template)
    kedges = np.linspace(0., 0.2, 41)
    kc = (kedges[:-1] + kedges[1:]) / 2.
    klim = {0: (0.02, 0.2, 0.005), 2: (0.02, 0.18, 0.005), 4: (0.03, 0.15, 0.005)}
    nout = [int(((kc >= lo) & (kc <= hi)).sum()) for lo, hi, step in klim.values()]
    rng = np.random.RandomState(21)
    template1 = 50. * rng.standard_normal(sum(nout))
    fiber = TopHatFiberCollisionsPowerSpectrumMultipoles(fs=0.6, Dfc=2.5)
    # data: the same model without the row selection, evaluated at b1 = 2, then cut by the observable itself (power_spectrum.py:108)
    obs_full = TracerPowerSpectrumMultipolesObservable(data={'b1': 2.}, kedges=kedges, ells=(0, 2, 4), wmatrix={'resolution': 4}, shotnoise=1e4,
                                                       theory=KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5)),
                                                       fiber_collisions=TopHatFiberCollisionsPowerSpectrumMultipoles(fs=0.6, Dfc=2.5))
    ObservablesGaussianLikelihood(observables=[obs_full], covariance=np.eye(120))()
    obs = TracerPowerSpectrumMultipolesObservable(data=np.asarray(obs_full.flatdata), k=kc, klim=klim, ells=(0, 2, 4), wmatrix={'resolution': 4}, theory=theory, shotnoise=1e4,
                                                  fiber_collisions=fiber, systematic_templates=[syst_template_0, template1])
    n = sum(nout)
    cov = spd_covariance(n, seed=3)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    for param in like.all_params.select(basename='syst_*'):
        param.update(prior=dict(dist='norm', loc=0., scale=2.), ref=dict(dist='norm', loc=0., scale=0.5))
    like()
    names = like.varied_params.names()
    theta = special_rows(sample_theta(like, 24, seed=23), names)
    out = run_batch(like, [obs], theta, names, nint=4)
    c = extract_observable(obs)
    c['templates'] = np.array(list(obs.wmatrix.systematic_templates.templates.values()))
    c['template_names'] = np.array(list(obs.wmatrix.systematic_templates.templates))
    c['kernel_correlated'] = np.asarray(fiber.kernel_correlated)
    c['kernel_uncorrelated'] = np.asarray(fiber.kernel_uncorrelated)
    c['fs'], c['Dfc'] = fiber.fs, fiber.Dfc
    c['template1'] = template1
    save('cfg2_fc_syst', names=np.array(names), theta=theta, obs0=c, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]), **out)
    print(names)


def simple_tracer():
    """SimpleTracerPowerSpectrumMultipoles (full_shape.py:367-414): damping at the fiducial (k, mu), sn0 / nd added before the projection; Standard template, qisoqap."""
    from desilike.theories.galaxy_clustering import SimpleTracerPowerSpectrumMultipoles
    template = StandardPowerSpectrumTemplate(z=0.5, apmode='qisoqap')
    theory = SimpleTracerPowerSpectrumMultipoles(template=template)
    for name in ['sigmapar', 'sigmaper']:
        theory.init.params[name].update(fixed=False)
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1': 2., 'sigmapar': 4., 'sigmaper': 3.}, kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 3}, theory=theory, shotnoise=1e4)
    cov = spd_covariance(120, seed=5)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 24, seed=29)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    assert not errors
    power, flat = [], []
    for row in theta:
        like(**dict(zip(names, row)))
        power.append(np.asarray(theory.power).copy())
        flat.append(np.asarray(like.flattheory).copy())
    wm, tmpl = obs.wmatrix, theory.template
    c = {'ells': np.array(wm.ells), 'ellsin': np.array(wm.ellsin), 'kin': np.asarray(theory.k), 'mu': np.asarray(theory.mu), 'wmu_ell': np.asarray(theory.wmu), 'k11': np.asarray(tmpl.k),
         'pk_dd_fid': np.asarray(tmpl.pk_dd_fid), 'f_fid': float(tmpl.f_fid), 'nd': theory.nd, 'eta': tmpl.eta, 'matrix_full': np.asarray(wm.matrix_full),
         'shotnoisein': np.asarray(wm.shotnoisein), 'shotnoiseout': np.asarray(wm.shotnoiseout), 'flatdata': np.asarray(obs.flatdata), 'template': tmpl.__class__.__name__}
    save('simple_tracer', names=np.array(names), theta=theta, obs0=c, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         power=np.array(power), flattheory=np.array(flat))
    print(names)


def kaiser_xi(eft=False, interp_order=1):
    """Full-shape correlation function multipoles: (EFT-like) Kaiser P_ell -> xi_ell through get_corr (tgc/base.py:46-139; FFTLog = the refstub's transform,
    third-party in the reference) with a ShapeFit template, ell = (0, 2, 4), 30 s-bins."""
    from desilike.theories.galaxy_clustering import KaiserTracerCorrelationFunctionMultipoles, EFTLikeKaiserTracerCorrelationFunctionMultipoles
    from desilike.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    template = ShapeFitPowerSpectrumTemplate(z=0.5)
    kwargs = {} if interp_order == 1 else dict(interp_order=interp_order)     # 3: cubic interpolation of P_ell (100-point theory grid) to the FFTLog grid, tgc/base.py:54-57, 66, 132
    theory = (EFTLikeKaiserTracerCorrelationFunctionMultipoles if eft else KaiserTracerCorrelationFunctionMultipoles)(template=template, **kwargs)
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 2.}, s=np.linspace(22.5, 167.5, 30), ells=(0, 2, 4), theory=theory)
    n, scale = 90, 3e-4
    rng = np.random.RandomState(14)
    A = rng.standard_normal((n, n)) * scale
    cov = A.dot(A.T) + (10. * scale)**2 * np.eye(n)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=cov)
    like()
    names = like.varied_params.names()
    theta = sample_theta(like, 24 if interp_order == 1 else 8, seed=19)
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: theta[:, i] for i, name in enumerate(names)})
    assert not errors
    power, corr, flat = [], [], []
    for row in theta:
        like(**dict(zip(names, row)))
        power.append(np.asarray(theory.power.power).copy())
        corr.append(np.asarray(theory.corr).copy())
        flat.append(np.asarray(like.flattheory).copy())
    wm, pw = obs.wmatrix, theory.power
    pt, tmpl = pw.pt, pw.pt.template
    c = {'ells': np.array(wm.ells), 'ellsin': np.array(wm.ellsin), 'kin': np.asarray(pt.k), 'mu': np.asarray(pt.mu), 'wmu_ell': np.asarray(pt.wmu), 'k11': np.asarray(tmpl.k),
         'pk_dd_fid': np.asarray(tmpl.pk_dd_fid), 'f_fid': float(tmpl.f_fid), 'kp': tmpl.kp, 'a': tmpl.a, 'nd': pw.nd, 's': np.asarray(theory.s), 'sout': np.concatenate(wm.s),
         'flatdata': np.asarray(obs.flatdata), 'template': tmpl.__class__.__name__}
    if eft:
        c.update(ct_matrix=np.asarray(pw.counterterm_matrix), sn_matrix=np.asarray(pw.stochastic_matrix), ct_params=np.array(pw.counterterm_params), sn_params=np.array(pw.stochastic_params))
    save('kaiser_xi' + ('_eft' if eft else '') + ('_cubic' if interp_order == 3 else ''), names=np.array(names), theta=theta, obs0=c, precision=np.asarray(like.precision), covariance=cov,
         priors=np.array([[{'uniform': 0, 'norm': 1}[s['dist']], s['lo'], s['hi'], s['loc'], s['scale']] for s in map(prior_spec, like.varied_params)]),
         logposterior=np.asarray(logpost), loglikelihood=np.asarray(derived[like._param_loglikelihood]), logprior=np.asarray(derived[like._param_logprior]),
         power=np.array(power), theory=np.array(corr), flattheory=np.array(flat))
    print(names)


if __name__ == '__main__':
    todo = sys.argv[1:] or ['cfg1', 'cfg2', 'cfg2_dense', 'cfg2_variants', 'marg_grid', 'cfg5', 'cfg4', 'cfg4_pk', 'cfg3_table', 'kaiser_xi', 'kaiser_xi_eft', 'cfg2_fc_syst', 'simple_tracer', 'cfg4_pcs', 'cfg4_models', 'cfg3_table_xi', 'cfg4_resummed', 'cfg4_flexible', 'cfg3_full', 'kaiser_xi_cubic', 'marg_multi', 'cfg5_bench']
    if 'cfg1' in todo: cfg1()
    if 'cfg2' in todo: cfg2(dense=False)
    if 'cfg2_dense' in todo: cfg2(dense=True)
    if 'cfg2_variants' in todo: cfg2_variants()
    if 'marg_grid' in todo: marg_grid()
    if 'cfg5' in todo: cfg5()
    if 'cfg4' in todo: cfg4('xi')
    if 'cfg4_pk' in todo: cfg4('pk')
    if 'cfg4_pcs' in todo: cfg4_kernel_broadband('xi'); cfg4_kernel_broadband('pk')
    if 'cfg3_table_xi' in todo: cfg3_table_xi()
    if 'cfg4_models' in todo: cfg4_models()
    if 'cfg4_resummed' in todo: cfg4_resummed()
    if 'cfg4_flexible' in todo: cfg4_flexible()
    if 'cfg2_fc_syst' in todo: cfg2_fc_syst()
    if 'simple_tracer' in todo: simple_tracer()
    if 'kaiser_xi' in todo: kaiser_xi(False)
    if 'kaiser_xi_eft' in todo: kaiser_xi(True)
    if 'cfg3_table' in todo: cfg3_table()
    if 'cfg3_full' in todo: cfg3_full()
    if 'kaiser_xi_cubic' in todo: kaiser_xi(eft=False, interp_order=3)
    if 'marg_multi' in todo: marg_multi()
    if 'cfg5_bench' in todo: cfg5_bench()
