"""Oracle restatement of the reference's PNG theory (primordial_non_gaussianity.py:75-116) against golden vectors of the reference itself (tests/golden/make_png_fixture.py)."""
import os

import numpy as np
import pytest

from oracle import np_oracle as oc

here = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ['png_bp_fixed', 'png_bphi_shapefit']


def load(name):
    return dict(np.load(os.path.join(here, 'golden', name + '.npz'), allow_pickle=True))


def png_oracle_point(g, row, return_all=False):
    names = list(g['names'])
    p = dict(zip(names, row))
    kt = g['c.k11']
    factor = oc.shapefit_factor(kt, float(g['c.kp']), float(g['c.a']), dm=p.get('dm', 0.), dn=p.get('dn', 0.)) if 'ShapeFit' in str(g['c.template']) else 1.
    pk_dd = g['c.pk_dd_fid'] * factor
    alpha = oc.png_alpha_prim(kt, pk_dd, g['pk_prim'], float(g['h']))
    mode = str(g['mode'])
    bf = oc.png_bfnl(mode, p['b1'], fnl_loc=p.get('fnl_loc', 0.), p=p.get('p', 1.), bphi=p.get('bphi', 1.))
    power = oc.png_tracer_power(g['c.kin'], g['c.mu'], g['c.wmu_ell'], kt[1:], pk_dd[1:], alpha[1:], float(g['c.f_fid']) * p.get('df', 1.), float(g['c.nd']), p['b1'], p['b1'], bf, bf,
                                sn0=p['sn0'], sigmasX=p['sigmas'], sigmasY=p['sigmas'], qpar=p.get('qpar', 1.), qper=p.get('qper', 1.))
    flat = oc.window_apply(power, matrix_full=g['c.matrix_full'], shotnoisein=g['c.shotnoisein'], shotnoiseout=g['c.shotnoiseout'])
    logl = oc.gaussian_loglikelihood(flat, g['c.flatdata'], g['precision'])[0]
    return (logl, power, flat) if return_all else logl


@pytest.mark.parametrize('name', FIXTURES)
def test_power_and_loglikelihood(name):
    g = load(name)
    for i in range(len(g['int_power'])):
        row = g['theta'][i]
        if not np.all(np.isfinite(row)): row = g['theta'][0]
        logl, power, flat = png_oracle_point(g, row, return_all=True)
        assert np.allclose(power, g['int_power'][i], rtol=1e-11, atol=1e-12 * np.abs(power).max())
        assert np.allclose(flat, g['int_flattheory'][i], rtol=1e-11, atol=1e-12 * np.abs(flat).max())
    checked = 0
    for i, row in enumerate(g['theta']):
        if not np.isfinite(g['logprior'][i]): continue
        logl = png_oracle_point(g, row)
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i])), (i, logl, g['loglikelihood'][i])
        checked += 1
    assert checked >= 20


def test_scale_dependent_bias_matters():
    g = load('png_bp_fixed')
    names = list(g['names'])
    row = g['theta'][0].copy()
    base = png_oracle_point(g, row, return_all=True)[1]
    row[names.index('fnl_loc')] = 0.
    nofnl = png_oracle_point(g, row, return_all=True)[1]
    assert np.abs(base[0, 0] / nofnl[0, 0] - 1.) > 0.01     # the monopole at the lowest k moves by more than a per cent


@pytest.mark.parametrize('name', FIXTURES)
def test_device_phases_on_the_cpu_against_the_reference(name):
    """The phase functions of ``dl_png_kernel`` (csrc/dl_fullshape.h) and the host-side constant folding, compiled for the host and run thread by thread
    (tests/csrc/emulate.cpp): power and log-likelihood of the reference's points, without a GPU."""
    from emulation import Emulation
    from test_gpu_png import spec_from_png_golden
    g = load(name)
    emu = Emulation(spec_from_png_golden(g))
    n = len(g['int_power'])
    power = emu.eval_theory(g['theta'][:n], iobs=0)[0]
    assert np.allclose(power.reshape(g['int_power'].shape), g['int_power'], rtol=1e-10, atol=1e-12 * np.abs(g['int_power']).max())
    ok = np.isfinite(g['logprior'])
    loglike = emu.eval_batch(g['theta'][ok])[0]
    ref = g['loglikelihood'][ok]
    assert (np.abs(loglike - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), (np.abs(loglike - ref) / np.maximum(1., np.abs(ref))).max()
