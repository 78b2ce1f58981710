"""GPU (-m gpu): the reference's call surface (likelihood(**params), vmap, varied_params) on the HIP path,
against golden vectors produced by the same calls on the reference."""
import numpy as np
import pytest

from golden_utils import load_golden
from test_host_api import make_cfg2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dense', [False, True])
def test_likelihood_call_surface(dense):
    g, like = make_cfg2(dense=dense)
    names = [str(n) for n in g['names']]
    for i in [0, 1, 5]:
        params = dict(zip(names, g['theta'][i]))
        logpost = like(**params)
        assert abs(logpost - g['logposterior'][i]) <= 1e-10 * max(1., abs(g['logposterior'][i]))
        assert abs(like.loglikelihood - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))
        assert np.isclose(like.logprior, g['logprior'][i], rtol=1e-13, atol=1e-13)
        # state of the last call, as the reference leaves it on the calculators (likelihoods/base.py:658-664; power_spectrum.py:400-404; full_shape.py:502-510)
        assert np.allclose(like.flattheory, g['flattheory'][i], rtol=1e-11, atol=1e-8) and np.allclose(like.flatdiff, g['flattheory'][i] - like.flatdata, rtol=1e-11, atol=1e-8)
        assert np.allclose(like.observable_flattheory(0), g['flattheory'][i], rtol=1e-11, atol=1e-8)
        if i < g['int_power'].shape[0]:
            assert np.allclose(like.theory_power(0), g['int_power'][i, 0], rtol=1e-11, atol=1e-12 * np.abs(g['int_power'][i, 0]).max())
    assert like.catch_errors == ()
    # missing parameters take their default value (b1 -> ref centre 1.5, base.py:1194-1196)
    assert np.isfinite(like())


def test_vmap_conventions():
    """vmap(likelihood, errors='return', return_derived=True)(dict of arrays) -> ((logposterior, derived), errors) (base.py:232-258; samplers/base.py:151-193)."""
    from desilike_amd import vmap
    g, like = make_cfg2(dense=True)
    names = [str(n) for n in g['names']]
    vlike = vmap(like, backend=None, errors='return', return_derived=True)
    (logpost, derived), errors = vlike({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert errors == {}
    ref = g['logposterior']
    finite = np.isfinite(ref)
    assert (np.abs(logpost[finite] - ref[finite]) <= 1e-10 * np.maximum(1., np.abs(ref[finite]))).all()
    assert np.array_equal(np.isneginf(logpost), np.isneginf(ref))   # rows outside the prior
    assert np.allclose(derived[like._param_loglikelihood], g['loglikelihood'], rtol=1e-12, atol=1e-10)
    # 2-D batch shape and NaN rows
    theta = g['theta'][:6].copy()
    theta[2, 0] = np.nan
    (logpost2, _), errors2 = vlike({name: theta[:, i].reshape(2, 3) for i, name in enumerate(names)})
    assert logpost2.shape == (2, 3) and list(errors2) == [2]
    with pytest.raises(Exception):
        vmap(like, errors='raise')({name: theta[:, i] for i, name in enumerate(names)})


def test_data_from_theory_and_sum():
    """data=dict(params) generates the data vector from theory => likelihood(fiducial) == 0 (likelihoods/tests/test_galaxy_clustering.py:6-16);
    (L + L)() == 2 L() - logprior (observables/tests/test_galaxy_clustering.py:229)."""
    g, like = make_cfg2(dense=False, data={'b1': 2.})
    assert abs(like(b1=2.)) < 1e-12
    assert np.allclose(like.observables[0].flatdata, g['obs0']['flatdata'], rtol=1e-12)
    g2, like2 = make_cfg2(dense=False, data={'b1': 2.})
    total = like + like2
    params = dict(b1=1.7, qpar=1.02, sn0=0.3)
    single = like(**params)
    assert np.isclose(total(**params), 2. * single - like.logprior, rtol=1e-12, atol=1e-10)


def test_fixed_parameter_override():
    g, like = make_cfg2(dense=False)
    base = like(b1=1.8)
    damped = like(b1=1.8, sigmapar=4., sigmaper=3.)   # fixed parameters can still be passed explicitly, like in the reference
    assert damped != base and np.isfinite(damped)
    assert like(b1=1.8) == base


def test_set_speed_reports_per_calculator_speeds():
    """base.py:695-735: ``runtime_info.speed`` of the calculators (evaluations per second of their own part), measured from the device kernels' own intervals."""
    from test_host_api import make_cfg2
    g, like = make_cfg2(dense=True)
    speeds = like._set_speed(niterations=3, batch=256)
    theory, window = like.observables[0].wmatrix.theory, like.observables[0].wmatrix
    assert set(speeds) == {theory, window, like}
    for calculator in (theory, window, like):
        info = calculator.runtime_info
        assert info.monitor.counter == 3 * 256 and info.speed == speeds[calculator] and 1e5 < info.speed < 1e10
    total = 1. / sum(1. / speed for speed in speeds.values())
    assert total > 1e5                                   # (the north star's figure, on 256-point batches)
    before = dict(speeds)
    assert like._set_speed(niterations=1, batch=256) == before        # speeds already set are left untouched (override=False)
