"""Child process of tests/test_gpu_switches.py: parity checks of the sections named on the command line, under whatever ``DL_*`` kernel-selection switches the parent put
in the environment (most are read once per process).  Prints ``switch probe ok`` at the end."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))

from golden_utils import load_golden, spec_from_golden, observable_constants
from oracle import np_oracle as orc


def close(got, ref, what):
    err = np.abs(got - ref) / np.maximum(1., np.abs(ref))
    assert (err <= 1e-10).all(), (what, float(err.max()))


def section_fs():
    """config 2 (dense window): the 64 reference points, and a 2537-row batch (ragged; above the chi2-GEMM limit when DL_CHI2_GEMM_MAX is lowered) against the oracle."""
    from desilike_amd._lib import Context
    g = load_golden('cfg2_shapefit_window_dense')
    ctx = Context(spec_from_golden(g), device=0)
    loglike, logprior, status = ctx.eval_batch_host(g['theta'])
    close(loglike, g['loglikelihood'], 'cfg2 dense vs reference')
    assert np.array_equal(status == 1, np.isneginf(g['logprior']))
    rng = np.random.RandomState(5)
    theta = rng.uniform([0.9, 0.9, -0.5, 0.5, 0.5, -3.], [1.1, 1.1, 0.5, 1.5, 3.5, 3.], size=(2537, 6))
    loglike = ctx.eval_batch_host(theta)[0]
    c, names = observable_constants(g), [str(n) for n in g['names']]
    rows = np.arange(0, 2537, 181)
    ref = []
    for row in theta[rows]:
        p = dict(zip(names, row)); p['b1'] = (p['b1'], p['b1'])
        ref.append(orc.gaussian_loglikelihood(orc.fullshape_observable(c, p)['flattheory'], c['flatdata'], g['precision'])[0])
    close(loglike[rows], np.array(ref), 'cfg2 dense 2537 rows vs oracle')
    for B in [1, 33, 256, 512, 1024]:
        close(ctx.eval_batch_host(theta[:B])[0], loglike[:B], 'batch {:d} vs the 2537-row pass'.format(B))


def section_two():
    """the benchmarked two-tracer shape against the reference fixture"""
    import bench
    g = load_golden('cfg5_bench')
    like = bench.make_likelihood_config5(0)
    names, rnames = like.varied_params.names(), [str(n) for n in g['names']]
    theta = g['theta'][:, [rnames.index(name) for name in names]]
    ctx, offset = like._get_posterior_context()
    ref = g['loglikelihood'] + g['logprior']
    for B in [128, 37, 256]:
        th = np.concatenate([theta, theta])[:B]
        got = ctx.eval_logposterior_host(th)[0] + offset
        r = np.concatenate([ref, ref])[:B]
        fin = np.isfinite(r)
        assert np.array_equal(np.isneginf(got), ~fin)
        close(got[fin], r[fin], 'two tracers, {:d} rows'.format(B))


def section_ens():
    """device-resident ensemble vs the NumPy stretch move with the same counter-based generator, bit for bit"""
    import bench
    from desilike_amd.samplers import EmceeSampler, EnsembleStretchMove, CounterRNG
    like = bench.make_likelihood_config5(0)
    sampler = EmceeSampler(like, nwalkers=64, seed=42)
    start, _ = sampler._get_start(64)
    chain = sampler.run(niterations=8, start=start)
    host = EnsembleStretchMove(64, 8, sampler.logposterior, rng=CounterRNG(sampler.counter_seed))
    coords, logp = start.copy(), sampler.logposterior(start)
    for it in range(8):
        coords, logp = host.step(coords, logp)
        assert np.array_equal(chain['logposterior'][it], logp), it


def section_mh():
    """device-resident Metropolis-Hastings chains vs the host driver (NumPy proposals, same counter-based draws): same weights, same positions"""
    import bench
    from desilike_amd.samplers import MCMCSampler
    kw = dict(chains=3, vectorize=4, seed=7, learn=False)
    dev = MCMCSampler(bench.make_likelihood_config5(0), **kw)
    host = MCMCSampler(bench.make_likelihood_config5(0), device_resident=False, **kw)
    start = dev._get_start(3)[0]
    cd, ch = dev.run(check_every=25, max_iterations=50, start=start), host.run(check_every=25, max_iterations=50, start=start)
    for a, b in zip(cd, ch):
        assert a['fweight'].size > 3 and np.array_equal(a['fweight'], b['fweight'])
        close(a['logposterior'], b['logposterior'], 'mh chains: device vs host driver')
        assert np.allclose(a['qpar'], b['qpar'], rtol=1e-11, atol=1e-13)


def section_emu():
    """config 3 at full size: the reference fixture (no marginalisation) and the oracle's marginalised solve"""
    from desilike_amd import vmap
    from test_gpu_emulator import make_cfg3_full, cfg3_oracle_solution
    g, like, pt, theory, solved = make_cfg3_full(marg=False)
    names = [str(n) for n in g['names']]
    (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(names)})
    assert errors == {}
    close(derived[like._param_loglikelihood], g['loglikelihood'], 'cfg3 vs reference')
    g, like, pt, theory, solved = make_cfg3_full(marg=True)
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=300, random_state=rng), *param.prior.limits) for param in like.varied_params])
    loglike = like._get_context().eval_batch_host(theta)[0]
    for i in [0, 157, 299]:
        close(loglike[i], cfg3_oracle_solution(like, pt, theory, solved, theta[i])['loglikelihood'], 'cfg3 marginalised vs oracle')


def section_bao():
    """config 4 (xi_ell and P_ell) against the reference fixtures, at the fixture's size and inside a 5000-row batch (one wave per point above 4096 rows)"""
    from desilike_amd import vmap
    from test_host_api import make_cfg4
    for space in ['xi', 'pk']:
        g, like = make_cfg4(space)
        rnames = [str(n) for n in g['names']]
        (logpost, derived), errors = vmap(like, errors='return', return_derived=True)({name: g['theta'][:, i] for i, name in enumerate(rnames)})
        assert errors == {}
        close(logpost, g['logposterior'], 'cfg4 ' + space)
        names = like.varied_params.names()
        theta = np.tile(g['theta'][:, [rnames.index(n) for n in names]], (5000 // len(g['theta']) + 1, 1))[:5000]
        loglike = like._get_context().eval_batch_host(theta)[0]
        close(loglike[:len(g['theta'])], g['loglikelihood'], 'cfg4 ' + space + ' in a 5000-row batch')


def section_tns():
    """the TNS one-loop theory against the reference fixture, at the fixture's size (one tile: split-K loop kernel) and inside a 1100-row batch (one wavenumber per wave)"""
    from desilike_amd._lib import Context
    from test_oracle_tns import load
    from test_gpu_tns import spec_from_tns_golden
    g = load('tns')
    ctx = Context(spec_from_tns_golden(g), device=0)
    ok = np.isfinite(g['theta']).all(axis=1)
    close(ctx.eval_batch_host(g['theta'])[0][ok], g['loglikelihood'][ok], 'tns vs reference')
    theta = np.tile(g['theta'][ok], (1100 // ok.sum() + 1, 1))[:1100]
    close(ctx.eval_batch_host(theta)[0][:ok.sum()], g['loglikelihood'][ok], 'tns in a 1100-row batch')


def section_png():
    """the PNG theory (two splines per point) against the reference fixtures"""
    from desilike_amd._lib import Context
    from test_oracle_png import load
    from test_gpu_png import spec_from_png_golden
    for name in ['png_bp_fixed', 'png_bphi_shapefit']:
        g = load(name)
        ctx = Context(spec_from_png_golden(g), device=0)
        close(ctx.eval_batch_host(g['theta'])[0], g['loglikelihood'], name + ' vs reference')


def section_stk():
    """the stacked emulator layout (emulators/conversion.py:44-98): the binding's key sets against the reference's own numbers (plain and marginalised), and a reduced
    copy of BASELINE configs[2] on that layout against the oracle (marginalised: finalize in the kernel's tail, or rows + the general finalize under DL_NO_GRAM_EPILOGUE)"""
    import ctypes
    from desilike_amd._lib import load
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'integration'))
    from desilike_mi355x import Library
    library = Library(os.path.join(os.path.dirname(HERE), 'desilike_amd', 'lib', 'libdesilike_amd.so'))
    g = np.load(os.path.join(HERE, 'golden', 'boundary_cfg3_stacked.npz'))
    ctx = library.create({key[4:]: g[key] for key in g.files if key.startswith('cfg/')}, device=0)
    loglike, logprior, status = library.eval_batch(ctx, g['theta'])
    inside = np.isfinite(g['logprior'])
    close(loglike[inside], g['loglikelihood'][inside], 'stacked emulator vs reference')
    library.lib.dl_destroy(ctx)
    from bench_configs import make_cfg3_stacked, cfg3_stacked_oracle_solution
    for marg in (True, False):
        like, pt, theory, solved, networks = make_cfg3_stacked(marg=marg, hidden=(32, 32), nk=30, seed=4)
        like.initialize()
        rng = np.random.RandomState(2)
        theta = np.column_stack([np.clip(param.ref.sample(size=70, random_state=rng), *param.prior.limits) for param in like.varied_params])
        out = like._get_context().eval_batch_host(theta, return_solved=marg)
        for i in range(0, 70, 9):
            sol = cfg3_stacked_oracle_solution(like, pt, theory, solved, theta[i])
            close(np.array([out[0][i]]), np.array([sol['loglikelihood']]), 'stacked emulator (marg = {}) vs oracle'.format(marg))
            if marg: assert np.allclose(out[3][i], sol['x'], rtol=1e-7, atol=1e-9)


if __name__ == '__main__':
    for name in sys.argv[1:]:
        globals()['section_' + name]()
    print('switch probe ok')
