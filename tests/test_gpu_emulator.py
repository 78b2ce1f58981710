"""GPU (-m gpu): BASELINE config 3 -- emulated perturbation-theory tables (Taylor exact vs the reference fixture; MLP vs the NumPy oracle),
velocileptors bias combination, analytic marginalisation of counter / stochastic terms, 4096 batched evaluations."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden
from test_host_api import make_cfg3
from test_oracle_emulator import table_point
from emulator_utils import taylor_state, EMU_PARAMS
from bench_configs import make_cfg3_full, cfg3_oracle_solution   # noqa: E402,F401  (shared with bench.py / tools)
from desilike_amd._lib import refresh_options as _refresh_options   # the library reads its DL_* switches once per process

pytestmark = pytest.mark.gpu


def test_taylor_emulated_velocileptors_vs_reference():
    from desilike_amd import vmap
    g, like = make_cfg3()
    names = [str(n) for n in g['names']]
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert errors == {}
    assert (np.abs(logpost - g['logposterior']) <= 1e-10 * np.maximum(1., np.abs(g['logposterior']))).all()
    like._evaluate_dict({name: g['theta'][:3, i] for i, name in enumerate(names)}, (3,), errors='return', return_flattheory=True)
    assert np.allclose(like.flattheory, g['flattheory'][:3], rtol=1e-11, atol=1e-8)


def make_mlp_likelihood(marg=True, seed=1, derived=None, hidden=(64, 64, 64), activation='silu'):
    from desilike_amd.emulators import EmulatedCalculator, MLPEmulatorEngine
    from desilike_amd.theories.galaxy_clustering import LPTVelocileptorsTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg3_velocileptors_table')
    rng = np.random.RandomState(seed)
    kpt = np.concatenate([[0.0005], np.geomspace(0.0015, 0.025, 20), np.arange(0.03, 0.51, 0.01)])
    nk, nin, hidden = kpt.size, 3, list(hidden)
    base = 2e4 * (kpt / 0.05)**0.96 / (1. + (kpt / 0.02)**2.5)

    def mlp(nout, ylimits, widths):
        layers, last = [], nin
        for width in widths + [nout]:
            layers.append((rng.standard_normal((last, width)) / last**0.5, 0.1 * rng.standard_normal(width)))
            last = width
        return MLPEmulatorEngine(xlimits=[[0.9, 1.1], [0.9, 1.1], [-0.1, 0.1]], layers=layers, activation=activation, ylimits=ylimits)

    amp = np.concatenate([[1.], 0.2 * np.ones(11), 0.05 * np.ones(4), [0., 0., 0.]])
    ylim = np.stack([-(base[None, :, None] * amp) * np.ones((3, 1, 1)), (base[None, :, None] * amp) * np.ones((3, 1, 1))], axis=-1).reshape(-1, 2)
    table = mlp(3 * nk * 19, ylim, hidden)
    table.yshape = (3, nk, 19)
    # stochastic monomials are exact constants (1, k^2, k^4 on the diagonal ells): zero output range + offset via ylimits lo = hi
    yl = table.ylimits.copy().reshape(3, nk, 19, 2)
    for ill in range(3): yl[ill, :, 16 + ill, :] = (kpt**(2 * ill))[:, None]
    table.ylimits = yl.reshape(-1, 2)
    engines = {'pktable': table, 'sigma8': mlp(1, [[0.7, 0.9]], [16]), 'fsigma8': mlp(1, [[0.4, 0.5]], [16])}
    specs = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
             'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}
    pt = EmulatedCalculator(EMU_PARAMS, engines, k=kpt, ells=(0, 2, 4), z=0.8, param_specs=specs)
    theory = LPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, tracer='ELG')
    solved = ['alpha0p', 'alpha2p', 'alpha4p', 'sn0p', 'sn2p'] if marg else []
    if derived is not None: solved = list(derived)
    for name in solved:
        theory.init.params[name].update(derived='.marg' if derived is None else derived[name])
    theory.init.params['sn4p'].update(fixed=True, value=0.3)
    obs = TracerPowerSpectrumMultipolesObservable(data=g['obs0']['flatdata'], kedges=np.linspace(0.02, 0.2, 37), ells=(0, 2, 4), wmatrix={'resolution': 2}, theory=theory, shotnoise=8e3)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    return g, like, pt, theory, solved


def oracle_flat(like, pt, theory, row, names, x):
    p = dict(zip(names, row)); p.update(x)
    xin = np.array([p[name] for name in EMU_PARAMS])
    eng = pt.engines
    act = eng['pktable'].activation
    pktable = orc.mlp_predict(xin, eng['pktable'].xlimits, eng['pktable'].layers, act, eng['pktable'].ylimits).reshape(3, -1, 19)
    sigma8 = orc.mlp_predict(xin, eng['sigma8'].xlimits, eng['sigma8'].layers, act, eng['sigma8'].ylimits)[0]
    fsigma8 = orc.mlp_predict(xin, eng['fsigma8'].xlimits, eng['fsigma8'].layers, act, eng['fsigma8'].ylimits)[0]
    params = {name: p.get(name, like.all_params[name].value) for name in ['b1p', 'b2p', 'bsp', 'b3p', 'alpha0p', 'alpha2p', 'alpha4p', 'sn0p', 'sn2p', 'sn4p']}
    pars = orc.velocileptors_pars(params, sigma8, fsigma8 / sigma8, basis='physical', model='lpt', snd=theory.snd, fsat=theory.fsat, sigv=theory.sigv)
    power = orc.interp1d(theory.k, pt.k, orc.tablevel_combine_bias_terms_poles(pktable, pars, nd=theory.nd).T).T
    wm = like.observables[0].wmatrix
    return orc.window_apply(power, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout)


def test_mlp_emulated_marginalised_4096():
    g, like, pt, theory, solved = make_mlp_likelihood(marg=True)
    names = like.varied_params.names()
    assert like.solved_params.names() == solved
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=4096, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status, xsolved = ctx.eval_batch_host(theta, return_solved=True)
    assert (status == 0).all() and np.isfinite(loglike).all()
    flatdata = like.flatdata
    nsol = len(solved)
    locs, scales = np.zeros(nsol), np.array([like.all_params[name].prior.scale for name in solved])
    for i in range(0, 4096, 64):     # 64 points of the 4096 against the oracle
        f0 = oracle_flat(like, pt, theory, theta[i], names, {name: 0. for name in solved})
        T = np.array([oracle_flat(like, pt, theory, theta[i], names, {n2: float(n2 == name) for n2 in solved}) - f0 for name in solved])
        sol = orc.solve_marginalized(f0 - flatdata, T, like.precision, x0=np.zeros(nsol), prior_loc=locs, prior_scale=scales, marg_mask=np.ones(nsol, dtype='?'))
        assert abs(loglike[i] - sol['loglikelihood']) <= 1e-10 * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
        assert np.allclose(xsolved[i], sol['x'], rtol=1e-7, atol=1e-9)


def test_mlp_emulated_not_marginalised():
    g, like, pt, theory, solved = make_mlp_likelihood(marg=False)
    names = like.varied_params.names()
    rng = np.random.RandomState(4)
    theta = np.column_stack([np.clip(param.ref.sample(size=64, random_state=rng), *param.prior.limits) for param in like.varied_params])
    loglike, logprior, status, flat = like._get_context().eval_batch_host(theta, return_flattheory=True)
    for i in range(0, 64, 8):
        ref = oracle_flat(like, pt, theory, theta[i], names, {})
        assert np.allclose(flat[i], ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())
        logl = orc.gaussian_loglikelihood(ref, like.flatdata, like.precision)[0]
        assert abs(loglike[i] - logl) <= 1e-10 * max(1., abs(logl))


@pytest.mark.parametrize('hidden,activation,marg', [((8,), 'silu', True), ((24, 40), 'tanh', True), ((100,), 'relu', False), ((128, 128, 128, 128, 128), 'silu', True),
                                                    ((16, 16, 16, 16, 16, 16, 16), 'tanh', False), ((64, 32), 'relu', True), ((5, 7, 3), 'silu', False)])
def test_mlp_architectures(hidden, activation, marg):
    """Widths that are not multiples of the 16 x 16 x 4 MFMA tile, one to seven hidden layers, the three activations of emulators/conversion.py:27-34, with and without
    solved parameters: 257 points (a ragged last tile of 16) against the oracle."""
    g, like, pt, theory, solved = make_mlp_likelihood(marg=marg, seed=5, hidden=hidden, activation=activation)
    names = like.varied_params.names()
    rng = np.random.RandomState(6)
    theta = np.column_stack([np.clip(param.ref.sample(size=257, random_state=rng), *param.prior.limits) for param in like.varied_params])
    loglike, logprior, status = like._get_context().eval_batch_host(theta)
    assert (status == 0).all()
    nsol = len(solved)
    scales = np.array([like.all_params[name].prior.scale for name in solved])
    for i in (0, 15, 16, 100, 255, 256):
        f0 = oracle_flat(like, pt, theory, theta[i], names, {name: 0. for name in solved})
        if nsol:
            T = np.array([oracle_flat(like, pt, theory, theta[i], names, {n2: float(n2 == name) for n2 in solved}) - f0 for name in solved])
            ref = orc.solve_marginalized(f0 - like.flatdata, T, like.precision, x0=np.zeros(nsol), prior_loc=np.zeros(nsol), prior_scale=scales, marg_mask=np.ones(nsol, dtype='?'))['loglikelihood']
        else:
            ref = orc.gaussian_loglikelihood(f0, like.flatdata, like.precision)[0]
        assert abs(loglike[i] - ref) <= 1e-10 * max(1., abs(ref)), (hidden, activation, i, loglike[i], ref)


def test_feature_path_matches_dense_path_and_is_repeatable():
    """The separable feature path (batched MFMA emulator + feature GEMM) against the dense per-point path of the same library (DL_NO_FEATURE_PATH), on a
    ragged batch, with and without marginalisation; 100 repeated evaluations are bit-identical."""
    import os
    import torch
    for marg in (True, False):
        g, like, pt, theory, solved = make_mlp_likelihood(marg=marg)
        rng = np.random.RandomState(12)
        theta = np.column_stack([np.clip(param.ref.sample(size=1000 + 7, random_state=rng), *param.prior.limits) for param in like.varied_params])
        ctx = like._get_context()
        fast = ctx.eval_batch_host(theta, return_solved=marg)
        os.environ['DL_NO_FEATURE_PATH'] = '1'; _refresh_options()
        try:
            from desilike_amd._lib import Context
            dense_ctx = Context(like._spec({}, like._flatdata_list(), like.precision), device=0)
            dense = dense_ctx.eval_batch_host(theta, return_solved=marg)
            dense_ctx.close()
        finally:
            del os.environ['DL_NO_FEATURE_PATH']; _refresh_options()
        assert np.array_equal(fast[2], dense[2]) and (fast[2] == 0).all()
        assert (np.abs(fast[0] - dense[0]) <= 1e-10 * np.maximum(1., np.abs(dense[0]))).all(), np.abs(fast[0] - dense[0]).max()
        assert np.allclose(fast[1], dense[1], rtol=1e-12, atol=1e-12)
        if marg: assert np.allclose(fast[3], dense[3], rtol=1e-8, atol=1e-10)
        th = torch.as_tensor(theta, dtype=torch.float64, device='cuda').contiguous()
        out, first = torch.empty(len(theta), dtype=torch.float64, device='cuda'), None
        for it in range(100):
            ctx.eval_logposterior(th, out)
            if first is None: first = out.clone()
            else: assert torch.equal(first, out), it


def test_taylor_emulator_fitted_on_the_gpu_theory():
    """(f2) Taylor emulator of the Kaiser multipoles fitted by ONE batch of dl_eval_theory (emulators/__init__.py:430-507), then used through
    EmulatedTracerPowerSpectrumMultipoles: exact at the centre (test_taylor.py:99-104), converging with the order nearby."""
    from desilike_amd.emulators import emulate_power
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles, EmulatedTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g = load_golden('cfg2_shapefit_window')
    kedges = np.linspace(0., 0.2, 41)

    def make(theory):
        obs = TracerPowerSpectrumMultipolesObservable(data=g['obs0']['flatdata'], kedges=kedges, ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=1e4)
        return ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])

    direct = make(KaiserTracerPowerSpectrumMultipoles(template=ShapeFitPowerSpectrumTemplate(z=0.5)))
    names = direct.varied_params.names()
    center = np.array([param.value for param in direct.varied_params])
    rng = np.random.RandomState(12)
    delta = np.array([0.5 * (param.delta[1] + param.delta[2]) for param in direct.varied_params])
    theta = np.vstack([center, center + 0.5 * delta * rng.uniform(-1., 1., (32, len(names)))])
    ref = direct._get_context().eval_batch_host(theta)[0]
    errors = {}
    for order in (2, 4):
        emulated = make(EmulatedTracerPowerSpectrumMultipoles(pt=emulate_power(direct, order=order)))
        assert emulated.varied_params.names() == names
        ll = emulated._get_context().eval_batch_host(theta)[0]
        assert abs(ll[0] - ref[0]) <= 1e-9 * max(1., abs(ref[0]))                 # the centre is reproduced
        errors[order] = np.abs(ll[1:] - ref[1:]).max()
    assert errors[4] < 0.05 * errors[2] and errors[4] < 1e-3 * np.abs(ref).max(), errors


def test_gram_finalize_one_lane_per_point():
    """The finalize of the fused emulator path (one lane per point on the Gram matrix of the feature GEMM: in the tail of the same kernel, or as
    `dl_finalize_marg_gram_kernel` with DL_NO_FUSED_SOLVE=1) against the 16-lanes-per-point kernel of the same library (DL_FM_NO_LANE_SOLVE=1) and against the oracle: all solved parameters marginalised, a mix of '.best' and '.marg' (the marginalised
    sub-block's determinant), a single '.best'; ragged batch."""
    import os
    cases = [None,
             {'alpha0p': '.marg', 'alpha2p': '.best', 'alpha4p': '.marg', 'sn0p': '.best'},
             {'sn0p': '.best'},
             {'alpha0p': '.best', 'alpha2p': '.best', 'alpha4p': '.marg', 'sn0p': '.marg', 'sn2p': '.marg'}]
    for derived in cases:
        g, like, pt, theory, solved = make_mlp_likelihood(marg=True, derived=derived)
        assert like.solved_params.names() == solved
        names = like.varied_params.names()
        rng = np.random.RandomState(21)
        theta = np.column_stack([np.clip(param.ref.sample(size=1000 + 7, random_state=rng), *param.prior.limits) for param in like.varied_params])
        theta[5, 0] = np.nan                                   # status codes travel through the same kernel
        theta[6, 1] = like.varied_params[names[1]].prior.limits[1] + 1.
        ctx = like._get_context()
        fast = ctx.eval_batch_host(theta, return_solved=True)          # solve in the tail of the fused kernel
        os.environ['DL_NO_FUSED_SOLVE'] = '1'; _refresh_options()
        try:
            separate = ctx.eval_batch_host(theta, return_solved=True)  # Gram matrix through memory, one lane per point in its own launch
            os.environ['DL_FM_NO_LANE_SOLVE'] = '1'; _refresh_options()
            wide = ctx.eval_batch_host(theta, return_solved=True)      # ... 16 lanes per point
        finally:
            os.environ.pop('DL_NO_FUSED_SOLVE', None); os.environ.pop('DL_FM_NO_LANE_SOLVE', None); _refresh_options()
        os.environ['DL_NO_SCALED_ROW0'] = '1'; _refresh_options()  # every monomial group with a full epilogue on the X rows (the form before the register path of row 0)
        try:
            plain = ctx.eval_batch_host(theta, return_solved=True)
        finally:
            os.environ.pop('DL_NO_SCALED_ROW0', None); _refresh_options()
        os.environ['DL_EF_NO_EARLY_THETA'] = '1'; _refresh_options()  # the code path of more than 32 sampled parameters (theta / priors from memory where they are used)
        try:
            late = ctx.eval_batch_host(theta, return_solved=True)
        finally:
            os.environ.pop('DL_EF_NO_EARLY_THETA', None); _refresh_options()
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(fast, late))
        assert np.array_equal(fast[2], separate[2]) and np.array_equal(fast[2], plain[2])
        ok = fast[2] == 0
        assert (np.abs(fast[0][ok] - plain[0][ok]) <= 1e-11 * np.maximum(1., np.abs(plain[0][ok]))).all() and np.allclose(fast[3][ok], plain[3][ok], rtol=1e-9, atol=1e-11)
        assert (np.abs(fast[0][ok] - separate[0][ok]) <= 1e-13 * np.maximum(1., np.abs(separate[0][ok]))).all() and np.allclose(fast[3][ok], separate[3][ok], rtol=1e-12, atol=1e-13)
        assert np.allclose(fast[1][ok], separate[1][ok], rtol=1e-13, atol=1e-13)
        assert np.array_equal(fast[2], wide[2]) and fast[2][5] != 0 and fast[2][6] != 0 and (np.delete(fast[2], [5, 6]) == 0).all()
        good = fast[2] == 0
        assert (np.abs(fast[0][good] - wide[0][good]) <= 1e-12 * np.maximum(1., np.abs(wide[0][good]))).all(), np.abs(fast[0][good] - wide[0][good]).max()
        assert np.allclose(fast[1][good], wide[1][good], rtol=1e-12, atol=1e-12)
        assert np.allclose(fast[3][good], wide[3][good], rtol=1e-9, atol=1e-11)
        # the samplers' entry point (log-posterior with -inf where a sampler rejects the point, samplers/base.py:185-191) through the fused tail
        logpost, st = ctx.eval_logposterior_host(theta)
        assert np.array_equal(st, fast[2]) and (logpost[~good] == -np.inf).all() and np.allclose(logpost[good], fast[0][good] + fast[1][good], rtol=1e-14, atol=0.)
        nsol = len(solved)
        scales = np.array([like.all_params[name].prior.scale for name in solved])
        mask = np.array([derived is None or derived[name] == '.marg' for name in solved])
        # derived outputs (likelihood Hessian w.r.t. the solved parameters, likelihoods/base.py:388-390) through the same three paths
        sub = np.ascontiguousarray(theta[7:7 + 40])
        dfast = ctx.eval_batch_derived_host(sub)
        os.environ['DL_NO_FUSED_SOLVE'] = '1'; _refresh_options()
        try:
            dsep = ctx.eval_batch_derived_host(sub)
        finally:
            os.environ.pop('DL_NO_FUSED_SOLVE', None); _refresh_options()
        assert dfast[4].shape == (40, nsol, nsol) and np.array_equal(dfast[4], dsep[4]) and np.allclose(dfast[0], fast[0][7:47], rtol=1e-13, atol=0.)
        for i in (0, 17, 500, 1006):
            f0 = oracle_flat(like, pt, theory, theta[i], names, {name: 0. for name in solved})
            T = np.array([oracle_flat(like, pt, theory, theta[i], names, {n2: float(n2 == name) for n2 in solved}) - f0 for name in solved])
            sol = orc.solve_marginalized(f0 - like.flatdata, T, like.precision, x0=np.zeros(nsol), prior_loc=np.zeros(nsol), prior_scale=scales, marg_mask=mask)
            assert abs(fast[0][i] - sol['loglikelihood']) <= 1e-10 * max(1., abs(sol['loglikelihood'])), (derived, i, fast[0][i], sol['loglikelihood'])
            assert np.allclose(fast[3][i], sol['x'], rtol=1e-7, atol=1e-9)
            if i == 17:
                scale_h = np.abs(sol['likelihood_hessian']).max()
                assert np.allclose(dfast[4][i - 7], sol['likelihood_hessian'], rtol=1e-9, atol=1e-11 * scale_h)


def test_unmarginalised_likelihood_in_one_launch():
    """No solved parameters: the fused emulator / feature-GEMM kernel keeps the residual row in LDS, chi2 = G[0][0], priors and status in its tail (one launch) -- against the
    two-launch form (residual rows to memory + finalize, DL_NO_GRAM_PLAIN=1) and against the oracle; ragged batch, NaN and out-of-prior rows."""
    import os
    g, like, pt, theory, solved = make_mlp_likelihood(marg=False)
    names = like.varied_params.names()
    rng = np.random.RandomState(5)
    theta = np.column_stack([np.clip(param.ref.sample(size=2000 + 3, random_state=rng), *param.prior.limits) for param in like.varied_params])
    theta[11, 2] = np.nan
    theta[12, 0] = like.varied_params[names[0]].prior.limits[0] - 1.
    ctx = like._get_context()
    one = ctx.eval_batch_host(theta)
    post, st = ctx.eval_logposterior_host(theta)
    os.environ['DL_NO_GRAM_PLAIN'] = '1'; _refresh_options()
    try:
        two = ctx.eval_batch_host(theta)
    finally:
        del os.environ['DL_NO_GRAM_PLAIN']; _refresh_options()
    assert np.array_equal(one[2], two[2]) and np.array_equal(st, one[2]) and one[2][11] != 0 and one[2][12] != 0 and (np.delete(one[2], [11, 12]) == 0).all()
    good = one[2] == 0
    assert (np.abs(one[0][good] - two[0][good]) <= 1e-11 * np.maximum(1., np.abs(two[0][good]))).all(), np.abs(one[0][good] - two[0][good]).max()
    assert np.allclose(one[1][good], two[1][good], rtol=1e-13, atol=1e-13) and (post[~good] == -np.inf).all() and np.allclose(post[good], one[0][good] + one[1][good], rtol=1e-14, atol=0.)
    for i in (0, 999, 2002):
        ref = oracle_flat(like, pt, theory, theta[i], names, {})
        logl = orc.gaussian_loglikelihood(ref, like.flatdata, like.precision)[0]
        assert abs(one[0][i] - logl) <= 1e-10 * max(1., abs(logl))


# north star: 1e-10 on logL -- also for the analytically marginalised value
MARG_TOL = 1e-10


def test_cfg3_full_size_vs_reference():
    """The reference ran its own velocileptors combination + interpolation + window + chi2 on the tables of these MLPs (tests/golden/make_golden.py::cfg3_full)."""
    from desilike_amd import vmap
    g, like, pt, theory, solved = make_cfg3_full(marg=False)
    names = [str(n) for n in g['names']]
    assert like.varied_params.names() == names
    wm = like.observables[0].wmatrix
    assert wm.matrix_full.shape == (120, 1200) and len(theory.k) == 400 and pt.engines['pktable'].layers[-1][0].shape == (64, 7296)
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert errors == {}
    assert (np.abs(derived[like._param_loglikelihood] - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all(), np.abs(derived[like._param_loglikelihood] - g['loglikelihood']).max()
    assert np.allclose(derived[like._param_logprior], g['logprior'], rtol=1e-12, atol=1e-12)
    like._evaluate_dict({name: g['theta'][:4, i] for i, name in enumerate(names)}, (4,), errors='return', return_flattheory=True)
    assert np.allclose(like.flattheory, g['flattheory'], rtol=1e-11, atol=1e-8)


def test_cfg3_full_size_marginalised_4096():
    """The 4096-point batch of BASELINE configs[2], 5 analytically marginalised parameters: feature path vs the oracle's per-point solve on reference-pinned theory
    vectors (theory of the solved parameters' unit vectors through the same oracle chain)."""
    from emulator_utils import CFG3_PARAMS
    g, like, pt, theory, solved = make_cfg3_full(marg=True)
    names = like.varied_params.names()
    assert like.solved_params.names() == solved
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=4096, random_state=rng), *param.prior.limits) for param in like.varied_params])
    ctx = like._get_context()
    loglike, logprior, status, xsolved = ctx.eval_batch_host(theta, return_solved=True)
    assert (status == 0).all() and np.isfinite(loglike).all()
    nsol = len(solved)
    worst = 0.
    for i in range(0, 4096, 64):    # 64 points spread over the batch (every 16-point tile position modulo 64 is the same lane: the offsets below walk the tile)
        i += (i // 64) % 16
        sol = cfg3_oracle_solution(like, pt, theory, solved, theta[i])
        err = abs(loglike[i] - sol['loglikelihood']) / max(1., abs(sol['loglikelihood']))
        worst = max(worst, err)
        assert err <= MARG_TOL, (i, loglike[i], sol['loglikelihood'], err)
        assert np.allclose(xsolved[i], sol['x'], rtol=1e-7, atol=1e-9)
    print('cfg3 marginalised, 64 of 4096 points: max relative error on logL {:.2e}'.format(worst))
