"""Load golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    data = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    out = {}
    for key in data.files:
        value = data[key]
        if value.dtype.kind in 'US' and value.ndim == 0:
            value = str(value)
        elif value.ndim == 0:
            value = value[()]
        if '.' in key:
            group, sub = key.split('.', 1)
            out.setdefault(group, {})[sub] = value
        else:
            out[key] = value
    return out


def observable_constants(g, iobs=0):
    """Oracle-side constants dict of observable ``iobs`` from a golden fixture."""
    c = dict(g['obs{:d}'.format(iobs)])
    c['template'] = {'ShapeFitPowerSpectrumTemplate': 'shapefit', 'turnover': 'turnover'}.get(str(c['template']), 'fixed')
    c['ellsin'] = tuple(int(ell) for ell in c['ellsin'])
    c['ells'] = tuple(int(ell) for ell in c['ells'])
    return c


def prior_list(g):
    return [dict(dist=['uniform', 'norm'][int(row[0])], limits=(row[1], row[2]), loc=row[3], scale=row[4]) for row in g['priors']]


def spec_from_golden(g):
    """Likelihood spec (the nested dict ``desilike_amd._lib.fill_config`` flattens into C-ABI config keys) from a golden fixture."""
    names = [str(n) for n in g['names']]
    observables = []
    iobs = 0
    while 'obs{:d}'.format(iobs) in g:
        c = g['obs{:d}'.format(iobs)]

        tracer = str(c.get('tracer', ''))

        def inp(name, default):
            if name in ('b1', 'sn0') and tracer: name = tracer + '.' + name
            return (names.index(name), default) if name in names else (-1, default)

        inputs = {'qpar': inp('qpar', 1.), 'qper': inp('qper', 1.), 'qiso': inp('qiso', 1.), 'qap': inp('qap', 1.), 'df': inp('df', 1.), 'dm': inp('dm', 0.), 'dn': inp('dn', 0.),
                  'sigmapar': inp('sigmapar', 0.), 'sigmaper': inp('sigmaper', 0.), 'b1X': inp('b1', 1.), 'b1Y': inp('b1', 1.), 'sn0': inp('sn0', 0.)}
        apmode = 3 if 'qiso' in names and 'qap' in names else 0
        obs = dict(theory=np.array([1 if 'ct_matrix' in c else 0]), template=np.array([1 if str(c['template']).startswith('ShapeFit') else 0]),
                   apmode=np.array([apmode]), transform=np.array([0]), eta=[1. / 3.], f_fid=[c['f_fid']], a=[c.get('a', 0.6)], kp=[c.get('kp', 0.03)], nd=[c['nd']],
                   ells_in=np.asarray(c['ellsin'], dtype='i4'), kin=c['kin'], mu=c['mu'], wmu_ell=c['wmu_ell'], k_t=c['k11'], pk_dd_fid=c['pk_dd_fid'],
                   wmatrix=c.get('matrix_full', None), kmask=c.get('kmask', None), offset=c.get('offset', None),
                   shotnoise_in=c['shotnoisein'], shotnoise_out=c['shotnoiseout'], flatdata=c['flatdata'])
        if 'ct_matrix' in c:
            obs['ct_matrix'], obs['sn_matrix'] = c['ct_matrix'], c['sn_matrix']
            ct = [[inp(str(n), 0.)] * 2 for n in c['ct_params']]
            sn = [inp(str(n), 0.) for n in c['sn_params']]
            inputs['ct'] = ([[t[0] for t in row] for row in ct], [[t[1] for t in row] for row in ct])
            inputs['sn'] = ([t[0] for t in sn], [t[1] for t in sn])
        obs['inputs'] = inputs
        observables.append(obs)
        iobs += 1
    return dict(n_params=np.array([len(names)]), priors=g['priors'], precision=g['precision'], observables=observables)


def spec_from_golden_bao(g):
    """Spec of a BAO fixture (cfg4_bao_xi / cfg4_bao_pk): wiggle model on the device, Hankel operator and broadband folded into the window."""
    from oracle.np_fftlog import hankel_operator      # (the host construction: these specs also feed the CPU tests; the product builds the operator on the device)
    from scipy import linalg
    names = [str(n) for n in g['names']]
    c = g['obs0']
    ells = [int(ell) for ell in c['ells']]

    def inp(name, default):
        return (names.index(name), default) if name in names else (-1, default)

    bbnames = [str(n) for n in c['broadband_params']]
    nbb = len(bbnames)
    bbflat = c['broadband_matrix'].reshape(-1, nbb)        # [(ell, x), n_bb]
    if 's' in c:
        H = linalg.block_diag(*hankel_operator(c['kin'], c['s'], ells))
        wmatrix = np.hstack([H, bbflat])
        extra = {}
    else:
        wmatrix = np.hstack([c['matrix_full'], c['matrix_full'].dot(bbflat)])
        extra = dict(shotnoise_in=c['shotnoisein'], shotnoise_out=c['shotnoiseout'])
    inputs = {'qpar': inp('qpar', 1.), 'qper': inp('qper', 1.), 'df': inp('df', 1.), 'b1X': inp('b1', 1.), 'b1Y': inp('b1', 1.), 'dbeta': inp('dbeta', 1.), 'sigmas': inp('sigmas', 0.),
              'sigmapar': inp('sigmapar', 9.), 'sigmaper': inp('sigmaper', 6.)}
    passin = [inp(name, 0.) for name in bbnames]
    inputs['pass'] = ([t[0] for t in passin], [t[1] for t in passin])
    obs = dict(theory=np.array([2]), template=np.array([0]), apmode=np.array([0]), transform=np.array([0]), eta=[1. / 3.], f_fid=[c['f_fid']], nd=[1.],
               ells_in=np.asarray(c['ellsin'], dtype='i4'), kin=c['kin'], mu=c['mu'], wmu_ell=c['wmu_ell'], k_t=c['k11'], pk_dd_fid=c['pk_dd_fid'], pknow_dd_fid=c['pknow_dd_fid'],
               bao_mode=np.array([1 if str(c['mode']) == 'reciso' else 0]), smoothing_radius=[float(c['smoothing_radius'])], wmatrix=wmatrix, flatdata=c['flatdata'], inputs=inputs, **extra)
    return dict(n_params=np.array([len(names)]), priors=g['priors'], precision=g['precision'], observables=[obs])


def constants_from_mirror(obs):
    """Oracle-side constants dict of an initialised host-mirror observable (no GPU needed): what ``observable_constants`` reads from a fixture."""
    obs.initialize()
    wm, theory = obs.wmatrix, obs.wmatrix.theory
    template = theory.template
    c = dict(template='shapefit' if type(template).__name__.startswith('ShapeFit') else 'fixed', k11=template.k, pk_dd_fid=template.pk_dd_fid, f_fid=template.f_fid,
             kp=getattr(template, 'kp', 0.03), a=getattr(template, 'a', 0.6), kin=theory.k, mu=theory.mu, wmu_ell=theory.wmu, ellsin=tuple(theory.ells), ells=tuple(wm.ells), nd=theory.nd,
             matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout, flatdata=obs.flatdata)
    if hasattr(theory, 'counterterm_matrix'):
        c.update(ct_matrix=theory.counterterm_matrix, sn_matrix=theory.stochastic_matrix, ct_params=list(theory.counterterm_params), sn_params=list(theory.stochastic_params))
    return c
