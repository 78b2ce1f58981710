"""Oracle restatement of the reference's TNS one-loop theory (full_shape.py:688-971) against golden vectors of the reference itself
(tests/golden/make_tns_fixture.py): kernels, the 29 loop tables, the projected tables, the tracer power, the likelihood."""
import os

import numpy as np
import pytest

from oracle import np_oracle as oc
from bench_configs import load_tns as load, tns_oracle_point   # noqa: E402,F401  (shared with bench.py / tools)

here = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ['tns', 'tns_eft', 'tns_standard_gaussian']


def test_trapz_weights_and_table_grid():
    g = load('tns')
    assert np.allclose(oc.tns_k11(g['c.kin']), g['k11_table'], rtol=1e-15)
    wq = oc.weights_trapz(g['c.k11'])
    assert np.isclose(wq.sum(), g['c.k11'][-1] - g['c.k11'][0], rtol=1e-14)


@pytest.mark.parametrize('name', FIXTURES)
def test_kernels(name):
    g = load(name)
    q = g['c.k11']
    rows = g['kernel_rows']
    k13d, k13t, ka = oc.tns_kernels(g['k11_table'][rows], q, oc.weights_trapz(q))
    for mine, ref in [(k13d, g['kernel13_d']), (k13t, g['kernel13_t']), (ka, g['kernel_a'])]:
        assert np.allclose(mine, ref, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize('name', FIXTURES)
def test_loop_tables_and_power(name):
    g = load(name)
    q = g['c.k11']
    kernels = oc.tns_kernels(g['k11_table'], q, oc.weights_trapz(q))
    for i in range(3):
        row = g['theta'][i]
        if not np.all(np.isfinite(row)): row = g['theta'][0]
        logl, pt, power, flat, pk_q = tns_oracle_point(g, row, kernels=kernels, return_all=True)
        assert np.allclose(pk_q, g['int_pk_dd_template'][i], rtol=1e-13)
        tab = oc.tns_table_matrix(oc.tns_pt(g['k11_table'], q, oc.weights_trapz(q), pk_q, kernels=kernels))
        ref = g['int_tables'][i]
        for r in range(29):   # every table to 1e-11 of its own largest entry (the sums are the reference's, in the reference's order)
            assert np.max(np.abs(tab[r] - ref[r])) <= 1e-11 * np.max(np.abs(ref[r])), r
        poles = np.concatenate([np.array([pt[key] for key in oc.TNS_NAMES]), pt['A'], pt['B']], axis=0)
        refp = g['int_poles'][i]
        for r in range(poles.shape[0]):
            assert np.max(np.abs(poles[r] - refp[r])) <= 1e-11 * np.max(np.abs(refp[r])), r
        assert np.allclose(power, g['int_power'][i], rtol=1e-11, atol=1e-11 * np.max(np.abs(power)))
        assert np.allclose(flat, g['int_flattheory'][i], rtol=1e-11, atol=1e-11 * np.max(np.abs(flat)))


@pytest.mark.parametrize('name', FIXTURES)
def test_loglikelihood(name):
    g = load(name)
    q = g['c.k11']
    kernels = oc.tns_kernels(g['k11_table'], q, oc.weights_trapz(q))
    checked = 0
    for i, row in enumerate(g['theta']):
        if not np.all(np.isfinite(row)) or not np.isfinite(g['logprior'][i]): continue
        logl = tns_oracle_point(g, row, kernels=kernels)
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i])), (i, logl, g['loglikelihood'][i])
        checked += 1
    assert checked >= 8


def test_loop_terms_matter():
    """The fixture is sensitive to the loop terms: dropping them moves the power by several per cent at k = 0.2."""
    g = load('tns')
    i = 0
    tab = g['int_tables'][i]
    k11 = g['k11_table']
    sel = k11 > 0.15
    assert np.max(np.abs(tab[1][sel] / tab[0][sel] - 1.)) > 0.02     # pk_dd vs pk11


@pytest.mark.parametrize('name,eft', [('tns', False), ('tns_eft', True)])
def test_host_mirror_parameters_match_the_reference(name, eft):
    """The mirror classes declare the parameters of the reference's parameter files (full_shape.yaml): same varied names, same priors, as recorded from the reference."""
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, TNSTracerPowerSpectrumMultipoles, EFTLikeTNSTracerPowerSpectrumMultipoles
    g = load(name)
    theory = (EFTLikeTNSTracerPowerSpectrumMultipoles if eft else TNSTracerPowerSpectrumMultipoles)(template=ShapeFitPowerSpectrumTemplate(z=0.5, fiducial='synthetic'))
    if not eft:
        for pname in ['bs', 'b3']: theory.init.params[pname].update(fixed=False)
    params = theory._all_params()
    varied = [param.name for param in params if param.varied]
    rnames = [str(n) for n in g['names']]
    assert sorted(varied) == sorted(rnames), (varied, rnames)
    for iname, pname in enumerate(rnames):
        assert np.array_equal(np.asarray(params[pname].prior.spec(), dtype='f8'), g['priors'][iname]), pname
    assert np.allclose(theory.k11, oc.tns_k11(theory.k)) and theory.template.k.size == 500 and np.isclose(theory.template.k[-1], 2.)


def test_device_functions_on_the_cpu_against_the_reference_tables():
    """csrc/dl_tns.h (geometry, interpolation records, P13 / A kernels, table entries) compiled for the host and run sequentially: the 29 tables of the fixture's first
    templates equal the reference's own (1e-10 of each table's largest entry) -- the arithmetic the GPU kernels execute, checked without a GPU."""
    from emulation import tns_tables
    g = load('tns')
    mus, wmus = oc.weights_leggauss_sym(10)
    for i in range(2):
        tables, outside = tns_tables(g['k11_table'], g['c.k11'], mus, wmus, g['int_pk_dd_template'][i])
        assert outside > 0     # some |k - q| fall below the template's range: the zero-weight branch is exercised
        for r in range(29):
            ref = g['int_tables'][i][r]
            assert np.max(np.abs(tables[r] - ref)) <= 1e-10 * np.max(np.abs(ref)), (i, r)


def test_device_bias_combination_against_the_reference_power():
    """dl_tns_combine_coef: the coefficients of the 29 projected tables in the tracer power reproduce the reference's ``power`` from the reference's own projected
    tables (mu'^2n factors and f powers are already inside the projected tables, so the mu'^2n classes are summed)."""
    from emulation import tns_combine
    g = load('tns')
    names = [str(n) for n in g['names']]
    for i in range(3):
        row = g['theta'][i] if np.all(np.isfinite(g['theta'][i])) else g['theta'][0]
        p = dict(zip(names, row))
        f = float(g['int_f'][i])
        cvec = tns_combine(f, p['b1'], p['b2'], p['bs'], p['b3'])
        poles = g['int_poles'][i]                        # rows: 12 spectra (projected WITH their f mu'^2 factors), A [3], B [3] (grouped by b1^2, b1, 1, with f and mu' inside)
        b1 = p['b1']
        power = sum(cvec[0][r] * poles[r] for r in range(1, 8))
        power = power + 2. * b1 * poles[8] + p['b2'] * poles[9] + poles[11]          # pk_dt, pk_b2t, pk_tt carry f mu'^2 / f^2 mu'^4 already
        power = power + b1**2 * (poles[12] + poles[15]) + b1 * (poles[13] + poles[16]) + (poles[14] + poles[17])
        power = power + p['sn0'] / float(g['c.nd'])
        assert np.allclose(power, g['int_power'][i], rtol=1e-11, atol=1e-11 * np.abs(g['int_power'][i]).max())
        # the f / b1 factors the device puts on the raw tables: consistent with the groups above
        assert np.isclose(cvec[1][8], 2. * b1 * f) and np.isclose(cvec[1][9], p['b2'] * f) and np.isclose(cvec[2][11], f * f)
        assert np.isclose(cvec[1][12], b1 * b1 * f) and np.isclose(cvec[1][13], b1 * f * f) and np.isclose(cvec[2][14], b1 * f * f) and np.isclose(cvec[2][15], f**3) and np.isclose(cvec[3][16], f**3)
        assert np.isclose(cvec[1][17], b1 * b1 * f * f) and np.isclose(cvec[1][18], -b1 * f**3) and np.isclose(cvec[1][19], -b1 * f**3) and np.isclose(cvec[1][20], f**4)
        assert np.isclose(cvec[2][21], b1 * b1 * f * f) and np.isclose(cvec[2][22], -b1 * f**3) and np.isclose(cvec[2][24], f**4) and np.isclose(cvec[3][25], -b1 * f**3) and np.isclose(cvec[3][27], f**4)
        assert np.isclose(cvec[4][28], f**4) and cvec[5][0] == 1. and np.count_nonzero(cvec[5]) == 1


@pytest.mark.parametrize('name', FIXTURES)
def test_whole_device_path_on_the_cpu_against_the_reference(name):
    """The TNS path end to end on the CPU -- host-side constant folding (dl_host.hpp), the device's geometry / table functions, the combination into the mu'^2n
    polynomials, the spline operator and dl_tns_eval_k (what the assembly kernel runs), window, chi2 -- against the reference's power and log-likelihoods."""
    from emulation import Emulation
    from test_gpu_tns import spec_from_tns_golden
    g = load(name)
    emu = Emulation(spec_from_tns_golden(g))
    rows = g['theta'][:2].copy()
    for i in range(2):
        if not np.all(np.isfinite(rows[i])): rows[i] = g['theta'][0]
    power = emu.eval_theory(rows, iobs=0)[0]
    ref = g['int_power'][:2]
    assert np.allclose(power.reshape(ref.shape), ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max()), np.abs(power.reshape(ref.shape) - ref).max() / np.abs(ref).max()
    ok = np.isfinite(g['theta']).all(axis=1) & np.isfinite(g['logprior'])
    idx = np.flatnonzero(ok)[:3]
    loglike = emu.eval_batch(g['theta'][idx])[0]
    assert (np.abs(loglike - g['loglikelihood'][idx]) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood'][idx]))).all()
