"""GPU (-m gpu): SURVEY 8 row a12 with the emulator layout the reference ships (emulators/conversion.py:44-98) -- engines '11' / 'loop' / 'ct' / 'st' stacked over (z, ell),
amplitude rescale by logA, redshift blend -- at the size of BASELINE configs[2] (4096 batched evaluations, 5 analytically marginalised parameters), through the MFMA kernel
``dl_emulated_stacked_kernel`` and through the general per-point path, against the oracle chain that tests/test_stacked.py pins on the reference's own outputs."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from bench_configs import make_cfg3_stacked, cfg3_stacked_oracle_solution, stacked_state, STACKED_ZGRID   # noqa: E402
from emulator_utils import STK_PARAMS, STK_SPECS, stacked_networks

pytestmark = pytest.mark.gpu
TOL = 1e-10


def sample(like, size, seed):
    rng = np.random.RandomState(seed)
    return np.column_stack([np.clip(param.ref.sample(size=size, random_state=rng), *param.prior.limits) for param in like.varied_params])


def test_stacked_full_size_marginalised_4096():
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=True)
    like.initialize()
    assert like.solved_params.names() == solved
    spec = like._spec({}, like._flatdata_list(), like.precision)['observables'][0]
    groups = np.asarray(spec['emu0']['groups'])
    assert groups.tolist() == [[0, 6, 0, 3], [6, 12, 3, 12], [12, 18, 12, 16], [18, 18, 16, 19]]          # two redshifts x three multipoles per engine; 'st' holds constant tables
    assert spec['wmatrix'].shape == (120, (6 * 64 + 1) * 16 + 3)
    theta = sample(like, 4096, 3)
    ctx = like._get_context()
    loglike, logprior, status, xsolved = ctx.eval_batch_host(theta, return_solved=True)
    assert (status == 0).all() and np.isfinite(loglike).all()
    worst = 0.
    for i in range(0, 4096, 64):
        i += (i // 64) % 16
        sol = cfg3_stacked_oracle_solution(like, pt, theory, solved, theta[i])
        err = abs(loglike[i] - sol['loglikelihood']) / max(1., abs(sol['loglikelihood']))
        worst = max(worst, err)
        assert err <= TOL, (i, loglike[i], sol['loglikelihood'], err)
        assert np.allclose(xsolved[i], sol['x'], rtol=1e-7, atol=1e-9)
    print('stacked emulator, marginalised, 64 of 4096 points: max relative error on logL {:.2e}'.format(worst))
    # ragged batches: the same rows whatever the batch they sit in
    for size in (1, 17, 100):
        part = ctx.eval_batch_host(theta[:size])[0]
        assert np.array_equal(part, loglike[:size]), size


def test_stacked_unmarginalised_and_general_path():
    """No solved parameters; the theory vector itself through the general per-point path (``return_flattheory``: power -> window GEMM), against the oracle; both paths agree."""
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=False, z=0.955, hidden=(48, 80), activation='silu', nk=40, seed=2)   # on an emulated redshift: one stack per engine survives
    like.initialize()
    spec = like._spec({}, like._flatdata_list(), like.precision)['observables'][0]
    assert np.asarray(spec['emu0']['groups'])[:, :2].tolist() == [[0, 3], [3, 6], [6, 9], [9, 9]]
    theta = sample(like, 300, 5)
    names = like.varied_params.names()
    ctx = like._get_context()
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert (status == 0).all()
    for i in range(0, 300, 23):
        sol = cfg3_stacked_oracle_solution(like, pt, theory, [], theta[i])
        assert abs(loglike[i] - sol['loglikelihood']) <= TOL * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
    like._evaluate_dict({name: theta[:5, j] for j, name in enumerate(names)}, (5,), errors='return', return_flattheory=True)
    flat = np.array(like.flattheory)
    for i in range(5):
        ll = orc.gaussian_loglikelihood(flat[i], like.flatdata, like.precision)[0]
        assert abs(ll - loglike[i]) <= TOL * max(1., abs(loglike[i]))


def test_stacked_with_scalar_engines_physical_basis():
    """The physical prior basis (full_shape.py:1577-1592) on a stacked node that also emulates sigma8 / fsigma8 (scalar MLP engines run beside the table networks)."""
    from desilike_amd.emulators import EmulatedCalculator, MLPEmulatorEngine
    from desilike_amd.theories.galaxy_clustering import REPTVelocileptorsTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    zgrid = STACKED_ZGRID[:3]
    networks = stacked_networks(zgrid, hidden=(32, 32), activation='tanh', seed=8, nk=30)
    pt = EmulatedCalculator.from_state(stacked_state(networks, zgrid, STK_PARAMS), param_specs=STK_SPECS)
    rng = np.random.RandomState(4)
    xlimits = pt.engines['11'].xlimits

    def scalar(lo, hi):
        layers, last = [], len(STK_PARAMS)
        for width in [16, 1]:
            layers.append((rng.standard_normal((last, width)) / last**0.5, 0.1 * rng.standard_normal(width)))
            last = width
        return MLPEmulatorEngine(xlimits=xlimits, layers=layers, activation='tanh', ylimits=[[lo, hi]])

    pt = EmulatedCalculator(STK_PARAMS, {**pt.engines, 'sigma8': scalar(0.7, 0.9), 'fsigma8': scalar(0.4, 0.5)}, k=pt.k, ells=pt.ells, z=zgrid, param_specs=STK_SPECS)
    theory = REPTVelocileptorsTracerPowerSpectrumMultipoles(pt=pt, z=0.4, tracer='LRG')
    for name in ['alpha0p', 'sn0p']: theory.init.params[name].update(derived='.marg')
    obs = TracerPowerSpectrumMultipolesObservable(data={'b1p': 1.4, 'b2p': 0.3}, kedges=np.linspace(0.02, 0.2, 19), ells=(0, 2), wmatrix={'resolution': 3}, theory=theory, shotnoise=8e3)
    A = rng.standard_normal((36, 36)) * 40.
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + 4e4 * np.eye(36))
    like.initialize()
    names, solved = like.varied_params.names(), like.solved_params.names()
    theta = sample(like, 64, 9)
    loglike, logprior, status, xsolved = like._get_context().eval_batch_host(theta, return_solved=True)
    assert (status == 0).all()
    wm = like.observables[0].wmatrix
    index = [list(pt.ells).index(ell) for ell in theory.ells]
    for i in range(0, 64, 7):
        def flat(x):
            p = dict(zip(names, theta[i])); p.update(x)
            X = {name: p[name] for name in STK_PARAMS}
            components = [orc.stacked_mlp_predict(X, STK_PARAMS, pt.engines[n].xlimits, pt.engines[n].layers, 'tanh', pt.engines[n].ylimits, amplitude_power=pw) for n, pw in [('11', 1), ('loop', 2), ('ct', 1), ('st', 0)]]
            pktable = orc.jaxeffort_pktable(components, zgrid=zgrid, z=[0.4])[..., 0][index]
            xin = np.array([X[name] for name in STK_PARAMS])
            sigma8 = orc.mlp_predict(xin, xlimits, pt.engines['sigma8'].layers, 'tanh', pt.engines['sigma8'].ylimits)[0]
            fsigma8 = orc.mlp_predict(xin, xlimits, pt.engines['fsigma8'].layers, 'tanh', pt.engines['fsigma8'].ylimits)[0]
            params = {name: p.get(name, like.all_params[name].value) for name in ['b1p', 'b2p', 'bsp', 'b3p', 'alpha0p', 'alpha2p', 'alpha4p', 'sn0p', 'sn2p', 'sn4p']}
            pars = orc.velocileptors_pars(params, sigma8, fsigma8 / sigma8, basis='physical', model='rept', snd=theory.snd, fsat=theory.fsat, sigv=theory.sigv)
            power = orc.interp1d(theory.k, pt.k, orc.tablevel_combine_bias_terms_poles(pktable, pars, nd=theory.nd).T).T
            return orc.window_apply(power, matrix_full=wm.matrix_full, shotnoisein=wm.shotnoisein, shotnoiseout=wm.shotnoiseout)
        f0 = flat({name: 0. for name in solved})
        T = np.array([flat({n2: float(n2 == name) for n2 in solved}) - f0 for name in solved])
        scales = np.array([like.all_params[name].prior.scale for name in solved])
        sol = orc.solve_marginalized(f0 - like.flatdata, T, like.precision, x0=np.zeros(len(solved)), prior_loc=np.zeros(len(solved)), prior_scale=scales, marg_mask=np.ones(len(solved), dtype='?'))
        assert abs(loglike[i] - sol['loglikelihood']) <= TOL * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
        assert np.allclose(xsolved[i], sol['x'], rtol=1e-7, atol=1e-9)


def test_stacked_correlation_function_multipoles():
    """xi_ell from the stacked emulated node (full_shape.py:1603-1629: the table combination on the log grid, then get_corr): the Hankel operator folds into the same
    per-group operators; counter terms marginalised; against the oracle's get_corr on the oracle's P_ell."""
    from desilike_amd.emulators import EmulatedCalculator
    from desilike_amd.theories.galaxy_clustering import REPTVelocileptorsTracerCorrelationFunctionMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerCorrelationFunctionMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    zgrid = STACKED_ZGRID[:4]
    networks = stacked_networks(zgrid, hidden=(64, 64), activation='silu', seed=12, nk=48, kmax=0.62)     # (the tables cover the 300-point log grid of the transform: no cubic extrapolation at high k)
    pt = EmulatedCalculator.from_state(stacked_state(networks, zgrid, STK_PARAMS), param_specs=STK_SPECS)
    theory = REPTVelocileptorsTracerCorrelationFunctionMultipoles(pt=pt, z=0.6, prior_basis='standard')
    for name in ['b3', 'alpha6']: theory.init.params[name].update(fixed=True)
    for name in ['alpha0', 'alpha2', 'alpha4']: theory.init.params[name].update(prior=dict(dist='norm', loc=0., scale=20.), derived='.marg')
    s = np.linspace(22.5, 167.5, 30)
    rng = np.random.RandomState(6)
    A = rng.standard_normal((90, 90)) * 3e-4
    obs = TracerCorrelationFunctionMultipolesObservable(data={'b1': 1.7, 'b2': 0.4, 'alpha0': 3.}, s=s, ells=(0, 2, 4), theory=theory)
    like = ObservablesGaussianLikelihood(observables=[obs], covariance=A.dot(A.T) + (3e-3)**2 * np.eye(90))
    like.initialize()
    names, solved = like.varied_params.names(), like.solved_params.names()
    assert solved == ['alpha0', 'alpha2', 'alpha4']
    theta = sample(like, 128, 10)
    loglike, logprior, status, xsolved = like._get_context().eval_batch_host(theta, return_solved=True)
    assert (status == 0).all()
    for i in range(0, 128, 17):
        def flat(x):
            p = dict(zip(names, theta[i])); p.update(x)
            X = {name: p[name] for name in STK_PARAMS}
            components = [orc.stacked_mlp_predict(X, STK_PARAMS, pt.engines[n].xlimits, pt.engines[n].layers, 'silu', pt.engines[n].ylimits, amplitude_power=pw) for n, pw in [('11', 1), ('loop', 2), ('ct', 1), ('st', 0)]]
            pktable = orc.jaxeffort_pktable(components, zgrid=zgrid, z=[0.6])[..., 0]
            params = {name: p.get(name, like.all_params[name].value if name in like.all_params else 0.) for name in ['b1', 'b2', 'bs', 'b3', 'alpha0', 'alpha2', 'alpha4', 'alpha6', 'sn0', 'sn2', 'sn4']}
            pars = orc.velocileptors_pars(params, 1., 1., basis='standard', model='rept')
            power = orc.interp1d(theory.k, pt.k, orc.tablevel_combine_bias_terms_poles(pktable, pars, nd=theory.nd).T).T
            return np.ravel(orc.get_corr(power, theory.k, s, (0, 2, 4)))
        f0 = flat({name: 0. for name in solved})
        T = np.array([flat({n2: float(n2 == name) for n2 in solved}) - f0 for name in solved])
        sol = orc.solve_marginalized(f0 - like.flatdata, T, like.precision, x0=np.zeros(3), prior_loc=np.zeros(3), prior_scale=np.full(3, 20.), marg_mask=np.ones(3, dtype='?'))
        assert abs(loglike[i] - sol['loglikelihood']) <= TOL * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])
        assert np.allclose(xsolved[i], sol['x'], rtol=1e-7, atol=1e-9)


def test_ensemble_sampler_on_the_stacked_likelihood():
    """The device-resident ensemble (dl_ensemble_*) on a likelihood whose theory is the stacked emulator: the same chain, bit for bit, as the host-driven stretch move
    with the same counter-based generator (both evaluate through the MFMA kernel; the proposals are the same arithmetic on either side)."""
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=True, hidden=(32, 32), nk=30, seed=4)
    like.initialize()
    ndim = len(like.varied_params)
    sampler = EmceeSampler(like, nwalkers=64, seed=11)
    start, logp0 = sampler._get_start(64)
    chain = sampler.run(niterations=8, start=start)
    host = EnsembleStretchMove(64, ndim, sampler.logposterior, rng=CounterRNG(sampler.counter_seed))
    coords, logp = start.copy(), sampler.logposterior(start)
    for it in range(8):
        coords, logp = host.step(coords, logp)
        assert np.array_equal(np.column_stack([chain[param.name][it] for param in like.varied_params]), coords), it
        assert np.array_equal(chain['logposterior'][it], logp), it
    assert np.isfinite(logp).all()


def test_stacked_batches_beyond_one_pass():
    """40 000 points (the context's workspaces hold 32 768 per pass: two passes, the second ragged): every row equals, bit for bit, the row a 4096-point batch gives it,
    and rows of the second pass agree with the oracle."""
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=True)
    like.initialize()
    base = sample(like, 4096, 13)
    reps = 40000 // 4096 + 1
    theta = np.tile(base, (reps, 1))[:40000]
    ctx = like._get_context()
    small = ctx.eval_batch_host(base)[0]
    loglike, logprior, status = ctx.eval_batch_host(theta)
    assert (status == 0).all()
    assert np.array_equal(loglike, np.tile(small, reps)[:40000])
    for i in (32768, 36001, 39999):
        sol = cfg3_stacked_oracle_solution(like, pt, theory, solved, theta[i])
        assert abs(loglike[i] - sol['loglikelihood']) <= TOL * max(1., abs(sol['loglikelihood'])), (i, loglike[i], sol['loglikelihood'])


@pytest.mark.parametrize('hidden,activation,marg,z', [((20, 36, 12), 'relu', True, 0.8), ((128,), 'silu', False, 0.62), ((8, 8, 8, 8, 8, 8), 'tanh', True, 0.955), ((100, 28), 'silu', True, 0.3),
                                                     ((64, 64, 128), 'tanh', True, 0.955), ((64, 64, 80), 'silu', True, 0.8), ((64, 64, 128), 'silu', False, 0.8)])   # (two 16-step layers, then another tile count -- waves without a task in one layer and with one in the next)
def test_stacked_architectures(hidden, activation, marg, z):
    """Widths off the 16 x 16 x 4 tile, one to six hidden layers, the three activations, between / on emulated redshifts: 273 points (ragged last tile) against the oracle."""
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=marg, z=z, hidden=hidden, activation=activation, nk=24, seed=21)
    like.initialize()
    theta = sample(like, 273, 17)
    loglike, logprior, status = like._get_context().eval_batch_host(theta)
    assert (status == 0).all()
    for i in (0, 15, 16, 150, 271, 272):
        sol = cfg3_stacked_oracle_solution(like, pt, theory, solved, theta[i])
        assert abs(loglike[i] - sol['loglikelihood']) <= TOL * max(1., abs(sol['loglikelihood'])), (hidden, activation, i, loglike[i], sol['loglikelihood'])


def test_stacked_overlapped_experiment_kernel():
    """``dl_emulated_stacked_ov_kernel`` (round 6: the networks of the next batch under the feature GEMM of the current group, or the two halves of the workgroup on half of the
    networks each; measured slower than the plain form and off by default, docs/EXPERIMENTS.md) at the size of BASELINE configs[2]: the same log-posteriors as the default
    kernel (another summation order inside the networks: 1e-12) and the oracle's at 1e-10, marginalised and plain."""
    import os
    from desilike_amd import _lib
    lib = _lib.load()
    try:
        for marg in (True, False):
            like, pt, theory, solved, networks = make_cfg3_stacked(marg=marg)
            like.initialize()
            theta = sample(like, 1000, 23)       # (ragged last tile)
            ctx = like._get_context()
            os.environ.pop('DL_STK_OVERLAP', None); lib.dl_options_refresh()
            base = ctx.eval_batch_host(theta)[0]
            for mode in ('1', '3', '4'):
                os.environ['DL_STK_OVERLAP'] = mode; lib.dl_options_refresh()
                loglike, logprior, status = ctx.eval_batch_host(theta)
                assert (status == 0).all()
                assert np.allclose(loglike, base, rtol=1e-12, atol=1e-12), (marg, mode, np.abs(loglike - base).max())
                for i in (0, 15, 16, 999):
                    sol = cfg3_stacked_oracle_solution(like, pt, theory, solved, theta[i])
                    assert abs(loglike[i] - sol['loglikelihood']) <= TOL * max(1., abs(sol['loglikelihood'])), (marg, mode, i)
    finally:
        os.environ.pop('DL_STK_OVERLAP', None); lib.dl_options_refresh()
