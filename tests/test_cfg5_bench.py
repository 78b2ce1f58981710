"""The shape bench.py times for BASELINE configs[4] (two config-2 tracers with dense 120 x 1200 windows, block-diagonal precision, n = 240, 8 parameters) against
the REFERENCE's own outputs on that shape (tests/golden/cfg5_bench.npz, written by tests/golden/make_golden.py::cfg5_bench): the oracle on CPU, the HIP path on the
GPU with the optimised launches on (merged theory launch, zero-panel skipping, 16-row tile, row alignment: the library's defaults), through the samplers' entry point
``dl_eval_logposterior`` and through the device-resident ensemble's own evaluation.  Tolerance: the north star's 1e-10 on logL (relative above |logL| = 1)."""
import os
import sys

import numpy as np
import pytest

from golden_utils import load_golden

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402  (the generators of the benchmark's inputs: the fixture holds checksums of what the reference consumed)


def checksum(a):
    a = np.asarray(a, dtype='f8')
    return np.concatenate([[a.sum(), (a**2).sum()], a.ravel()[::997][:64]])


def bench_inputs():
    from scipy import linalg
    kedges = np.linspace(0., 0.2, 41)
    windows = [bench.dense_window(kedges, (0, 2, 4), seed=seed) for seed in [7, 8]]
    covariance = linalg.block_diag(bench.synthetic_covariance(120, seed=1), bench.synthetic_covariance(120, seed=2))
    return windows, covariance


def test_bench_inputs_are_what_the_reference_consumed():
    g = load_golden('cfg5_bench')
    windows, covariance = bench_inputs()
    for (kin, wmat), ref in zip(windows, g['window_checksum']):
        assert wmat.shape == (120, 1200)
        assert np.allclose(checksum(wmat), ref, rtol=1e-12, atol=1e-300)
    assert np.allclose(checksum(covariance), g['covariance_checksum'], rtol=1e-13)
    from desilike_amd.utils import blockinv
    precision = blockinv([[covariance[:120, :120], covariance[:120, 120:]], [covariance[120:, :120], covariance[120:, 120:]]])
    assert np.allclose(checksum(precision), g['precision_checksum'], rtol=1e-9, atol=1e-16)


def oracle_constants(g, windows):
    names = [str(n) for n in g['names']]
    consts = []
    from oracle import np_oracle as orc
    mu, wmu = orc.weights_leggauss_sym(8)
    for iobs, (tracer, shotnoise) in enumerate([('LRG', 1e4), ('ELG', 4e3)]):
        kin, wmat = windows[iobs]
        kin1 = kin[:kin.size // 3] if kin.size == 1200 else kin
        consts.append(dict(template='shapefit', k11=g['k11'], pk_dd_fid=g['pk_dd_fid'], f_fid=float(g['f_fid']), kp=0.03, a=0.6, kin=kin1, mu=mu,
                           wmu_ell=orc.multipole_weights(mu, wmu, (0, 2, 4)), ellsin=(0, 2, 4), ells=(0, 2, 4), nd=1e-4, matrix_full=wmat,   # (the theory's own default nd: the observable's shotnoise is not forwarded to it, full_shape.py:162)
                           shotnoisein=np.array([shotnoise, 0., 0.]), shotnoiseout=np.array([shotnoise] * 40 + [0.] * 80), flatdata=g['flatdata'][120 * iobs:120 * (iobs + 1)], tracer=tracer))
    return names, consts


def oracle_loglikelihood(g, windows, precision, theta):
    from oracle import np_oracle as orc
    names, consts = oracle_constants(g, windows)
    logl, flat = [], []
    for row in theta:
        p = dict(zip(names, row))
        theory = []
        for c in consts:
            q = {name: p[name] for name in ['qpar', 'qper', 'dm', 'df']}
            q['b1'] = (p[c['tracer'] + '.b1'],) * 2
            q['sn0'] = p[c['tracer'] + '.sn0']
            theory.append(orc.fullshape_observable(c, q)['flattheory'])
        theory = np.concatenate(theory)
        flat.append(theory)
        logl.append(orc.gaussian_loglikelihood(theory, g['flatdata'], precision)[0])
    return np.array(logl), np.array(flat)


def test_oracle_on_the_bench_shape():
    """(Pins the oracle at the benchmarked shape: bench.py's config-5 leg checks the device against this oracle on the final walker positions.)"""
    g = load_golden('cfg5_bench')
    windows, covariance = bench_inputs()
    precision = np.linalg.inv(covariance)
    sel = np.r_[0:12, 64:76, 125:128]
    logl, flat = oracle_loglikelihood(g, windows, precision, g['theta'][sel])
    assert np.allclose(flat, g['flattheory'][sel], rtol=1e-11, atol=1e-8)
    ref = g['loglikelihood'][sel]
    assert (np.abs(logl - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), np.abs(logl - ref).max()


@pytest.mark.gpu
def test_gpu_logposterior_on_the_bench_shape():
    """The benchmarked likelihood object itself (bench.make_likelihood_config5), default (optimised) launch path, against the reference's outputs."""
    import torch
    g = load_golden('cfg5_bench')
    like = bench.make_likelihood_config5(0)
    names = like.varied_params.names()
    rnames = [str(n) for n in g['names']]
    assert sorted(names) == sorted(rnames)
    theta = g['theta'][:, [rnames.index(name) for name in names]]
    # data generated from theory by the mirror == data generated from theory by the reference
    flatdata = np.concatenate(like._flatdata_list())
    assert np.allclose(flatdata, g['flatdata'], rtol=1e-11, atol=1e-9)
    ctx, offset = like._get_posterior_context()
    ref = g['loglikelihood'] + g['logprior']
    for B in [128, 37]:    # the sampler's half-step is 256 rows; ragged and full tiles
        got = ctx.eval_logposterior_host(theta[:B])[0] + offset
        assert np.array_equal(np.isneginf(got), np.isneginf(ref[:B]))
        finite = np.isfinite(ref[:B])
        assert (np.abs(got[finite] - ref[:B][finite]) <= 1e-10 * np.maximum(1., np.abs(ref[:B][finite]))).all(), np.abs(got[finite] - ref[:B][finite]).max()
    # 256 rows = the half-step of the 512-walker ensemble (two copies of the fixture's points): device buffers, asynchronous entry point
    th = torch.as_tensor(np.concatenate([theta, theta]), dtype=torch.float64, device='cuda').contiguous()
    out = torch.empty(256, dtype=torch.float64, device='cuda')
    ctx.eval_logposterior(th, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy() + offset
    ref2 = np.concatenate([ref, ref])
    finite = np.isfinite(ref2)
    assert np.array_equal(np.isneginf(got), ~finite)
    assert (np.abs(got[finite] - ref2[finite]) <= 1e-10 * np.maximum(1., np.abs(ref2[finite]))).all()
    # separate loglikelihood / logprior / flattheory of the plain context
    loglike, logprior, status, flat = like._get_context().eval_batch_host(theta, return_flattheory=True)
    assert np.allclose(flat, g['flattheory'], rtol=1e-11, atol=1e-8)
    assert (np.abs(loglike - g['loglikelihood']) <= 1e-10 * np.maximum(1., np.abs(g['loglikelihood']))).all()
    fin = np.isfinite(g['logprior'])
    assert np.allclose(logprior[fin], g['logprior'][fin], rtol=1e-13, atol=1e-13) and np.array_equal(status == 1, ~fin)


@pytest.mark.gpu
def test_gpu_ensemble_log_posteriors_on_the_bench_shape():
    """The log-posteriors the device-resident 512-walker ensemble carries after a run == the reference-pinned oracle at those walker positions."""
    from desilike_amd.samplers import EmceeSampler
    g = load_golden('cfg5_bench')
    like = bench.make_likelihood_config5(0)
    sampler = EmceeSampler(like, nwalkers=512, seed=42, device_resident=True)
    sampler.run(niterations=20)
    coords, logp = sampler._last
    windows, covariance = bench_inputs()
    names = like.varied_params.names()
    rnames = [str(n) for n in g['names']]
    sel = np.arange(0, 512, 16)
    theta = coords[sel][:, [names.index(name) for name in rnames]]
    logl, _ = oracle_loglikelihood(g, windows, np.linalg.inv(covariance), theta)
    from oracle import np_oracle as orc
    from golden_utils import prior_list
    ref = logl + orc.logprior(theta, prior_list(g))
    assert (np.abs(logp[sel] - ref) <= 1e-10 * np.maximum(1., np.abs(ref))).all(), np.abs(logp[sel] - ref).max()
