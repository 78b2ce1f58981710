"""Duck-typed measurement / window / covariance containers: the arrays of a reference fixture wrapped in minimal objects that
expose what the reference reads of ``lsstypes`` objects (observables/galaxy_clustering/power_spectrum.py:165-179, window.py:337-352, likelihoods/base.py:594-603) compile
to the SAME device configuration as the plain arrays."""
import numpy as np
import pytest

from golden_utils import load_golden


class Pole(object):
    def __init__(self, x, edges, value, shotnoise=None):
        self._x, self._edges, self._value, self._shotnoise = np.asarray(x), np.asarray(edges), np.asarray(value), shotnoise

    def coords(self, name): return self._x
    def edges(self, name): return self._edges
    def value(self): return self._value

    def values(self, name):
        if name != 'shotnoise' or self._shotnoise is None: raise KeyError(name)
        return np.full(self._x.size, self._shotnoise)


class Tree(object):
    """``.ells`` / ``.get(ells=)`` for a measurement, ``.observables`` / ``.get(observables=)`` for a set of them."""

    def __init__(self, poles=None, observables=None):
        self._poles, self._observables = poles or {}, observables or {}
        self.ells = list(self._poles)
        self.observables = list(self._observables)

    def get(self, ells=None, observables=None):
        return self._observables[observables] if observables is not None else self._poles[ells]


class Matrix(object):
    def __init__(self, value, observable, theory=None):
        self._value, self.observable, self.theory = np.asarray(value), observable, theory

    def value(self): return self._value


def measurement(kedges, ells, flat, shotnoise=None, pad=3):
    """A measurement on a WIDER k range than the analysis (``pad`` extra bins on each side of every multipole): the containers are cut to the requested range."""
    dk = kedges[1] - kedges[0]
    wide = np.concatenate([kedges[0] - dk * np.arange(pad, 0, -1), kedges, kedges[-1] + dk * np.arange(1, pad + 1)])
    mid = 0.5 * (wide[:-1] + wide[1:])
    n = len(kedges) - 1
    poles = {}
    for ill, ell in enumerate(ells):
        value = np.concatenate([np.full(pad, -7.), flat[ill * n:(ill + 1) * n], np.full(pad, -9.)])
        poles[ell] = Pole(mid, np.column_stack([wide[:-1], wide[1:]]), value, shotnoise=shotnoise if ell == 0 else None)
    return Tree(poles=poles)


def build(g, containers):
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    c = g['obs0']
    kedges, ells = np.linspace(0., 0.2, 41), (0, 2, 4)
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    if not containers:
        obs = TracerPowerSpectrumMultipolesObservable(data=c['flatdata'], kedges=kedges, ells=ells, wmatrix=c['matrix_full'], kin=c['kin'], ellsin=ells, theory=theory, shotnoise=1e4)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    data = measurement(kedges, ells, c['flatdata'], shotnoise=1e4)
    # window: rows on the wide output grid (padding rows are garbage the cut must drop), columns = (ell_in, kin), plus an unused input multipole
    n, nin, pad = 40, len(c['kin']), 3
    rows = []
    for ill in range(3):
        rows += [np.full(3 * nin, 1e30)] * pad + list(c['matrix_full'][ill * n:(ill + 1) * n]) + [np.full(3 * nin, -1e30)] * pad
    rows = np.array(rows)
    value = np.hstack([rows, np.full((rows.shape[0], nin), 5e29)])                      # a fourth input multipole (ell = 6) nobody asks for
    theory_tree = Tree(poles={ell: Pole(c['kin'], np.column_stack([c['kin'], c['kin']]), np.zeros(nin)) for ell in (0, 2, 4, 6)})
    wmatrix = Matrix(value, observable=data, theory=theory_tree)
    obs = TracerPowerSpectrumMultipolesObservable(data=data, klim={ell: (0., 0.2) for ell in ells}, wmatrix=wmatrix, ellsin=ells, theory=theory)
    # covariance on the wide vector
    m = n + 2 * pad
    index = np.concatenate([ill * m + pad + np.arange(n) for ill in range(3)])
    wide = 1e12 * np.eye(3 * m)
    wide[np.ix_(index, index)] = g['covariance']
    covariance = Matrix(wide, observable=Tree(observables={obs.name: data}))
    return ObservablesGaussianLikelihood(observables=[obs], covariance=covariance)


def test_containers_compile_to_the_same_configuration():
    from desilike_amd._lib import fill_config
    g = load_golden('cfg2_shapefit_window_dense')
    keys = []
    for containers in (False, True):
        like = build(g, containers)
        like.initialize()
        out = {}
        fill_config(like._spec({}, like._flatdata_list(), like.precision), lambda key, a: out.__setitem__(key, a), lambda key, a: out.__setitem__(key, a))
        keys.append(out)
    plain, duck = keys
    assert sorted(plain) == sorted(duck)
    for key in plain:
        assert np.allclose(np.ravel(plain[key]), np.ravel(duck[key]), rtol=1e-13, atol=0.), key
    obs = like.observables[0]
    assert obs.shotnoise == 1e4 and tuple(obs.ells) == (0, 2, 4) and all(len(kk) == 40 for kk in obs.k)


def test_mocks_give_data_and_covariance():
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    g = load_golden('cfg2_shapefit_window_dense')
    c = g['obs0']
    rng = np.random.RandomState(5)
    kedges, ells = np.linspace(0., 0.2, 41), (0, 2, 4)
    draws = c['flatdata'] + rng.standard_normal((200, c['flatdata'].size)) * 50.
    mocks = [measurement(kedges, ells, draw, shotnoise=1e4, pad=0) for draw in draws]
    theory = KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5))
    # the reference's call form (power_spectrum.py:64-75): mocks as ``covariance`` give the sample covariance, mocks as ``data`` only their mean
    obs = TracerPowerSpectrumMultipolesObservable(data=mocks, wmatrix=c['matrix_full'], kin=c['kin'], ellsin=ells, theory=theory)
    obs.initialize()
    assert np.allclose(obs.flatdata, draws.mean(axis=0)) and obs.covariance is None and obs.nobs is None
    obs = TracerPowerSpectrumMultipolesObservable(data=mocks[:3], covariance=mocks, wmatrix=c['matrix_full'], kin=c['kin'], ellsin=ells, theory=theory)
    obs.initialize()
    assert np.allclose(obs.flatdata, draws[:3].mean(axis=0)) and obs.nobs == 200 and obs.mocks.shape == draws.shape
    assert np.allclose(obs.covariance, np.cov(draws, rowvar=False, ddof=1))
    # a covariance-matrix container at the observable level (power_spectrum.py:115-117): the block of this observable's bins
    obs = TracerPowerSpectrumMultipolesObservable(data=mocks[0], covariance=Matrix(g['covariance'], observable=mocks[0]), wmatrix=c['matrix_full'], kin=c['kin'], ellsin=ells, theory=theory)
    obs.initialize()
    assert np.allclose(obs.covariance, g['covariance'])
    obs = TracerPowerSpectrumMultipolesObservable(data=c['flatdata'], covariance=mocks, wmatrix=c['matrix_full'], kin=c['kin'], ellsin=ells, theory=theory)   # binning from the mocks
    obs.initialize()
    assert obs.nobs == 200 and tuple(obs.ells) == ells and np.allclose(obs.covariance, np.cov(draws, rowvar=False, ddof=1))
    with pytest.raises(NotImplementedError):
        TracerPowerSpectrumMultipolesObservable(data=mocks[0], klim={0: (0., 0.2, 0.01)}, wmatrix=c['matrix_full'], kin=c['kin'], ellsin=ells, theory=theory).initialize()


def test_window_with_input_multipoles_on_different_grids():
    """A window container whose input multipoles live on DIFFERENT wavenumber grids: accepted with ``kin`` -- every multipole rebinned from its own grid, as the reference
    does (window.py:347-349) --, refused without (350-351)."""
    from desilike_amd import utils
    from desilike_amd.observables.galaxy_clustering import _containers
    rng = np.random.RandomState(3)
    kedges, ells = np.linspace(0., 0.2, 11), (0, 2)
    grids = {0: np.linspace(0.005, 0.25, 30), 2: np.linspace(0.01, 0.24, 24)}
    value = rng.standard_normal((20, 54))
    theory = Tree(poles={ell: Pole(grid, np.column_stack([grid, grid]), np.zeros(grid.size)) for ell, grid in grids.items()})
    observable = measurement(kedges, ells, np.zeros(20), pad=0)
    wmatrix = Matrix(value, observable=observable, theory=theory)
    mid = [0.5 * (kedges[:-1] + kedges[1:])] * 2
    with pytest.raises(ValueError, match='not the same for all multipoles'):
        _containers.read_window(wmatrix, ells, mid)
    kin = np.linspace(0.01, 0.2, 40)
    matrix, kout, ellsin = _containers.read_window(wmatrix, ells, mid, kin=kin)
    assert matrix.shape == (20, 80) and np.array_equal(kout, kin) and ellsin == (0, 2)
    expected = np.hstack([value[:, :30].dot(utils.matrix_lininterp(kin, grids[0]).T), value[:, 30:].dot(utils.matrix_lininterp(kin, grids[2]).T)])
    assert np.allclose(matrix, expected, rtol=1e-14, atol=0.)
