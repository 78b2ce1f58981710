"""CPU: pin the NumPy oracle against golden vectors captured from the reference's own numpy path."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, observable_constants, prior_list


def point_inputs(g, row):
    names = [str(n) for n in g['names']]
    p = dict(zip(names, row))
    if 'qiso' in p:
        p['qpar'], p['qper'] = orc.ap_qparqper('qisoqap', 1. / 3., qiso=p['qiso'], qap=p['qap'])
    p['b1'] = (p['b1'], p['b1'])
    c = g['obs0']
    if 'ct_params' in c:
        p['ct'] = [2. * p.get(str(n), 0.) for n in c['ct_params']]   # auto-correlation: sum over the two (identical) tracers
        p['sn'] = [p.get(str(n), 0.) for n in c['sn_params']]
    return p


@pytest.mark.parametrize('name', ['cfg1_kaiser_nowindow', 'cfg2_shapefit_window', 'cfg2_shapefit_window_dense', 'cfg2v_eft_damping_qisoqap'])
def test_fullshape_chain(name):
    g = load_golden(name)
    c = observable_constants(g)
    priors = prior_list(g)
    nint = g['int_power'].shape[0]
    for i, row in enumerate(g['theta']):
        out = orc.fullshape_observable(c, point_inputs(g, row))
        if i < nint:
            for key in ['pk_dd_template', 'pk_dd', 'pk_dt', 'pk_tt', 'power', 'flatpower']:
                ref = g['int_' + key][i, 0]
                assert np.allclose(out[key], ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max()), key
        assert np.allclose(out['flattheory'], g['flattheory'][i], rtol=1e-12, atol=1e-9)
        logl, flatdiff = orc.gaussian_loglikelihood(out['flattheory'], c['flatdata'], g['precision'])
        lp = orc.logprior(row, priors)
        assert np.isclose(lp, g['logprior'][i], rtol=1e-13, atol=1e-13) or (np.isinf(lp) and np.isinf(g['logprior'][i]))
        # tolerance of the north star: 1e-10 on logL (relative above |logL| = 1)
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i])), (i, logl, g['loglikelihood'][i])


def test_gl_weights():
    # theories/tests/test_galaxy_clustering.py:6-12: sum of mu-weights = 1
    for n in [4, 8, 10, 20]:
        mu, w = orc.weights_leggauss_sym(n)
        assert np.isclose(w.sum(), 1., rtol=0, atol=1e-15)
        assert (mu > 0).all() and (mu < 1).all()


def test_kaiser_closed_form():
    # desilike/tests/generate_data_mocks.py:105-111: P0 = b^2 (1 + 2 beta/3 + beta^2/5) P etc. at qpar = qper = 1
    g = load_golden('cfg2_shapefit_window')
    c = observable_constants(g)
    b, f = 1.7, c['f_fid']
    out = orc.fullshape_observable({**c, 'shotnoisein': None}, dict(b1=(b, b), sn0=0.))
    beta = f / b
    pk = orc.interp1d(np.log10(c['kin']), np.log10(c['k11']), c['pk_dd_fid'])
    assert np.allclose(out['power'][0], b**2 * (1. + 2. / 3. * beta + beta**2 / 5.) * pk, rtol=1e-12)
    assert np.allclose(out['power'][1], b**2 * (4. / 3. * beta + 4. / 7. * beta**2) * pk, rtol=1e-11)
    assert np.allclose(out['power'][2], 8. / 35. * b**2 * beta**2 * pk, rtol=1e-9)


def test_notaknot_moment_form():
    # the kernel's spline formulation (moments + piecewise cubic) == scipy not-a-knot (jax.py:263-265)
    g = load_golden('cfg2_shapefit_window')
    c = observable_constants(g)
    x, y = np.log10(c['k11']), g['int_pk_dd_template'][3, 0]
    M = orc.notaknot_moments(x, y)
    xq = np.log10(np.geomspace(c['kin'][0] / 1.3, c['kin'][-1] * 1.3, 5000))
    ref = orc.interp1d(xq, x, y)
    assert np.allclose(orc.notaknot_eval(xq, x, y, M), ref, rtol=1e-12, atol=0)


def test_window_bininteg():
    g = load_golden('cfg2_shapefit_window')
    c = observable_constants(g)
    edges = np.column_stack([np.linspace(0., 0.2, 41)[:-1], np.linspace(0., 0.2, 41)[1:]])
    xin, mat = orc.window_matrix_bininteg([edges] * 3, resolution=10)
    assert np.allclose(xin, c['kin'], rtol=1e-14)
    assert np.allclose(mat.T, c['matrix_full'], rtol=1e-13, atol=1e-16)
