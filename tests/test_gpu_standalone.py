"""GPU: calculators evaluated on their own -- ``theory(**params).power``, ``observable(**params).flattheory`` (desilike/base.py:1194-1196: ``calculator(**params)``
runs the pipeline below the calculator and returns it) -- against the reference's own intermediate outputs stored in the fixtures (``int_power``, ``int_flatpower``)."""
import numpy as np
import pytest

from golden_utils import load_golden
from test_host_api import make_cfg2

pytestmark = pytest.mark.gpu


def test_theory_and_observable_calls():
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, KaiserTracerCorrelationFunctionMultipoles
    g = load_golden('cfg2_shapefit_window_dense')
    names = [str(name) for name in g['names']]
    kin = g['obs0']['kin']
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5), k=kin, ells=(0, 2, 4))
    for i in range(3):
        params = dict(zip(names, g['theta'][i]))
        power = theory(**params).power                               # the theory alone, on its own k
        ref = g['int_power'][i, 0]                                    # [3, 400] of the reference
        assert power.shape == ref.shape and np.allclose(power, ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max())
    # parameters by dict, defaults for the others; a change of a parameter is followed
    power0 = theory({'b1': 2.}).power
    assert np.allclose(theory(b1=2.).power, power0, rtol=0, atol=0)
    theory.all_params['b1'].update(value=2.)
    assert np.allclose(theory().power, power0, rtol=0, atol=0)
    # observable = window output (before the subtraction of the data), one array per multipole
    g2, like = make_cfg2(dense=True)
    observable = like.init['observables'][0]
    for i in range(3):
        params = dict(zip(names, g['theta'][i]))
        flat = observable(**params).flattheory
        assert np.allclose(flat, g['int_flatpower'][i, 0], rtol=1e-11, atol=1e-11 * np.abs(g['int_flatpower'][i, 0]).max())
        assert len(observable.theory) == 3 and np.allclose(np.concatenate(observable.theory), flat, rtol=0, atol=0)
    # correlation function theory: corr [n_ell, n_s], equal to the flat theory of an observable built on it
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    s = np.linspace(20., 150., 27)
    xi = KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5), s=s, ells=(0, 2))
    corr = xi(b1=1.8, qpar=1.01).corr
    assert corr.shape == (2, 27) and np.isfinite(corr).all()
    obs = TracerCorrelationFunctionMultipolesObservable(data=np.zeros(54), s=s, ells=(0, 2), theory=KaiserTracerCorrelationFunctionMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5)))
    assert np.allclose(obs(b1=1.8, qpar=1.01).flattheory, corr.ravel(), rtol=1e-13, atol=1e-13 * np.abs(corr).max())
    # calculators that are not end points of a pipeline say so
    with pytest.raises(NotImplementedError):
        ShapeFitPowerSpectrumTemplate(z=0.5)(dm=0.01)
