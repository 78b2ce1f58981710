"""CPU: the device arithmetic (desilike_amd/csrc/dl_fullshape.h phases + dl_host.hpp constant folding), emulated
workgroup by workgroup on the host, against golden vectors from the reference.  Catches kernel logic errors without a GPU."""
import numpy as np
import pytest

from golden_utils import load_golden, spec_from_golden
from emulation import Emulation


@pytest.mark.parametrize('name', ['cfg1_kaiser_nowindow', 'cfg2_shapefit_window', 'cfg2_shapefit_window_dense', 'cfg2v_eft_damping_qisoqap'])
def test_emulated_kernel_vs_reference(name):
    g = load_golden(name)
    emu = Emulation(spec_from_golden(g))
    theta = g['theta']
    nint = g['int_power'].shape[0]
    power, tables = emu.eval_theory(theta[:nint])
    for i, key in enumerate(['pk_dd', 'pk_dt', 'pk_tt']):
        ref = g['int_' + key][:, 0]
        assert np.allclose(tables[:, i], ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max()), key
    ref = g['int_power'][:, 0]
    assert np.allclose(power, ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())
    loglike, flat = emu.eval_batch(theta)
    assert np.allclose(flat, g['flattheory'], rtol=1e-11, atol=1e-8)
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(loglike - g['loglikelihood']) <= tol).all(), np.abs(loglike - g['loglikelihood']).max()


def test_emulated_two_tracers():
    """Two observables with a joint covariance, shared template parameters, per-tracer b1 / sn0 (config 5 geometry)."""
    g = load_golden('cfg5_two_tracers')
    emu = Emulation(spec_from_golden(g))
    loglike, flat = emu.eval_batch(g['theta'])
    assert np.allclose(flat, g['flattheory'], rtol=1e-11, atol=1e-8)
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(loglike - g['loglikelihood']) <= tol).all()


@pytest.mark.parametrize('space', ['xi', 'pk'])
def test_emulated_bao_vs_reference(space):
    """BAO wiggle kernel (bao.py:117-140) + Hankel operator / broadband folded into the window, vs the reference running on the oracle's FFTLog."""
    from golden_utils import spec_from_golden_bao
    g = load_golden('cfg4_bao_' + space)
    emu = Emulation(spec_from_golden_bao(g))
    power, _ = emu.eval_theory(g['theta'])
    assert np.allclose(power, g['wiggle_power'], rtol=1e-11, atol=1e-12 * np.abs(g['wiggle_power']).max())
    loglike, flat = emu.eval_batch(g['theta'])
    assert np.allclose(flat, g['flattheory'], rtol=1e-10, atol=1e-12 * np.abs(g['flattheory']).max())
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(loglike - g['loglikelihood']) <= tol).all(), np.abs((loglike - g['loglikelihood']) / g['loglikelihood']).max()


def test_emulated_velocileptors_vs_reference():
    """Taylor-emulated PT node + REPT velocileptors tracer: device features x folded matrix vs the reference running the same (polynomial) node."""
    from test_host_api import make_cfg3
    g, like = make_cfg3()
    spec = like._spec({}, like._flatdata_list(), like.precision)
    emu = Emulation(spec)
    loglike, flat = emu.eval_batch(g['theta'])
    assert np.allclose(flat, g['flattheory'], rtol=1e-11, atol=1e-8)
    tol = 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))
    assert (np.abs(loglike - g['loglikelihood']) <= tol).all(), np.abs((loglike - g['loglikelihood']) / g['loglikelihood']).max()


def test_emulation_under_sanitizers():
    """SURVEY.md section 5 (sanitizer runs on the CPU build): the device phase functions and the host-side constant folding compiled with AddressSanitizer +
    UndefinedBehaviorSanitizer, driven through the fixtures of this file in a child interpreter with libasan preloaded; any report fails the test."""
    import os, subprocess, sys
    from emulation import build_emulation
    build_emulation(sanitize=True)
    libasan = subprocess.check_output(['g++', '-print-file-name=libasan.so']).decode().strip()
    if not os.path.isabs(libasan) or not os.path.isfile(libasan):
        pytest.skip('libasan not found')
    here = os.path.dirname(os.path.abspath(__file__))
    code = ('import sys; sys.path.insert(0, {here!r}); sys.path.insert(0, {root!r})\n'
            'import test_emulation as t\n'
            'for name in ["cfg1_kaiser_nowindow", "cfg2_shapefit_window_dense", "cfg2v_eft_damping_qisoqap"]: t.test_emulated_kernel_vs_reference(name)\n'
            't.test_emulated_two_tracers()\n'
            'import test_oracle_png as p, test_oracle_tns as n\n'
            'p.test_device_phases_on_the_cpu_against_the_reference("png_bphi_shapefit")\n'
            'n.test_device_functions_on_the_cpu_against_the_reference_tables()\n'
            'n.test_whole_device_path_on_the_cpu_against_the_reference("tns_standard_gaussian")\n'
            'print("sanitized emulation ok")\n').format(here=here, root=os.path.dirname(here))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1', DL_EMULATION_SANITIZE='1')
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    err = out.stderr.decode()
    assert out.returncode == 0 and 'sanitized emulation ok' in out.stdout.decode(), err[-3000:]
    assert 'AddressSanitizer' not in err and 'runtime error' not in err, err[-3000:]


@pytest.mark.parametrize('name', ['cfg1_kaiser_nowindow', 'cfg2_shapefit_window_dense', 'cfg5_two_tracers'])
def test_emulated_analytic_gradient(name):
    """The device's gradient phase functions (csrc/dl_fullshape_grad.h: contraction of d(theory) / d(qpar, qper, f, b1, sn0, dm) with Y = -W~^T d~, chain rule to the
    sampled parameters) run on the CPU, against the five-point stencil of the emulated log-likelihood (itself pinned on the reference above)."""
    g = load_golden(name)
    emu = Emulation(spec_from_golden(g))
    theta = g['theta'][:3].copy()
    logl, grad = emu.eval_grad(theta)
    assert np.allclose(logl, emu.eval_batch(theta)[0], rtol=1e-13, atol=1e-12)
    fd = np.zeros_like(grad)
    for p in range(theta.shape[1]):
        h = 1e-3

        def f(x):
            th = theta.copy(); th[:, p] += x
            return emu.eval_batch(th)[0]

        fd[:, p] = (-f(2 * h) + 8 * f(h) - 8 * f(-h) + f(-2 * h)) / (12 * h)
    assert (np.abs(grad - fd).max(axis=0) <= 1e-8 * np.abs(fd).max(axis=0)).all(), np.abs(grad - fd).max(axis=0) / np.abs(fd).max(axis=0)


def test_emulated_analytic_gradient_with_dn():
    """dn sampled as well (a third spline pass: the template's dn-derivative), on the config-2 fixture with one more theta column."""
    g = load_golden('cfg2_shapefit_window')
    spec = spec_from_golden(g)
    P = int(spec['n_params'][0])
    spec['n_params'] = np.array([P + 1])
    spec['priors'] = np.vstack([spec['priors'], [0., -0.5, 0.5, 0., 1.]])
    spec['observables'][0]['inputs']['dn'] = (P, 0.)
    emu = Emulation(spec)
    theta = np.column_stack([g['theta'][:3], [0.02, -0.03, 0.01]])
    logl, grad = emu.eval_grad(theta)
    fd = np.zeros_like(grad)
    for p in range(P + 1):
        h = 1e-3

        def f(x):
            th = theta.copy(); th[:, p] += x
            return emu.eval_batch(th)[0]

        fd[:, p] = (-f(2 * h) + 8 * f(h) - 8 * f(-h) + f(-2 * h)) / (12 * h)
    assert (np.abs(grad - fd).max(axis=0) <= 1e-8 * np.abs(fd).max(axis=0)).all(), np.abs(grad - fd).max(axis=0) / np.abs(fd).max(axis=0)
