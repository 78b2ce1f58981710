"""Validate the chain writer of desilike_amd/io.py against the REFERENCE, in the build container only (the reference never travels to the GPU box):

    python tests/golden/validate_chain_io.py

1. a synthetic chain of a marginalised fit (3 sampled + 2 analytically marginalised parameters; loglikelihood / logprior carrying the Hessian entries w.r.t. the
   solved parameters as derivatives) is written with ``desilike_amd.io.ChainFile.save`` (.npz and .npy);
2. the reference (imported through tests/golden/refstub) loads both files with ``Chain.load`` and runs ``Chain.sample_solved`` (samples/chain.py:229-263) on them;
3. the reference re-saves the chain: key sets / parameter-state key sets of its file and ours are compared;
4. fixtures: ``chain_io.npz`` (inputs + the reference's sample_solved outputs + key sets) and ``chain_reference_written.npz`` (a file written by the
   reference's own ``Chain.save``: a data fixture for the reader test).
"""
import os
import sys
import tempfile
import warnings

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
warnings.filterwarnings('ignore')


def synthetic_chain(seed=11, shape=(6, 4)):
    """What ``derived_for_chain`` produces for a run: sampled columns, solved values, loglikelihood / logprior with packed Hessians."""
    from desilike_amd.parameter import Parameter
    from desilike_amd import io
    rng = np.random.RandomState(seed)
    params = {'qpar': Parameter('qpar', value=1., prior=dict(limits=[0.9, 1.1]), ref=dict(limits=[0.99, 1.01]), delta=0.01, latex='q_{\\parallel}'),
              'df': Parameter('df', value=1., prior=dict(limits=[0., 2.]), ref=dict(limits=[0.95, 1.05])),
              'LRG.b1': Parameter('b1', namespace='LRG', value=2., prior=dict(limits=[0., 4.]), ref=dict(dist='norm', loc=2., scale=0.1)),
              'alpha0': Parameter('alpha0', value=0., prior=dict(dist='norm', loc=0., scale=12.5), derived='.marg'),
              'sn0': Parameter('sn0', value=0., prior=dict(dist='norm', loc=0.2, scale=2.), derived='.auto')}
    arrays = {name: params[name].ref.sample(size=shape, random_state=rng) for name in ['qpar', 'df', 'LRG.b1']}
    solved = ['alpha0', 'sn0']
    for name in solved: arrays[name] = rng.standard_normal(shape)
    A = rng.standard_normal(shape + (2, 3))
    hess_like = -np.einsum('...ik,...jk->...ij', A, A) - 0.1 * np.eye(2)          # negative definite
    hess_prior = np.zeros(shape + (2, 2)); hess_prior[..., 0, 0] = -1. / 12.5**2; hess_prior[..., 1, 1] = -1. / 2.**2
    loglike, logprior = -rng.uniform(10., 60., size=shape), -rng.uniform(0., 2., size=shape)
    arrays['loglikelihood'] = io.pack_hessian(loglike, hess_like)
    arrays['logprior'] = io.pack_hessian(logprior, hess_prior)
    arrays['logposterior'] = loglike + logprior
    arrays['aweight'] = np.ones(shape)
    arrays['fweight'] = np.ones(shape, dtype='i8')
    derivs = {'loglikelihood': io.solved_derivs(solved), 'logprior': io.solved_derivs(solved)}
    return io.ChainFile(arrays, params=params, derivs=derivs), solved


def main():
    from desilike.samples import Chain
    from desilike_amd import io
    ours, solved = synthetic_chain()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for ext in ['npz', 'npy']:
            fn = os.path.join(tmp, 'ours.' + ext)
            ours.save(fn)
            chain = Chain.load(fn)                                      # the reference reads our file
            assert chain.shape == ours.shape, (chain.shape, ours.shape)
            assert [str(p) for p in chain.params(solved=True)] == solved
            for name, value in ours.arrays.items():
                assert np.array_equal(np.asarray(chain[name]), value), name
            for name in ['loglikelihood', 'logprior']:
                assert chain[name].derivs is not None and len(chain[name].derivs) == 4
                assert np.array_equal(np.asarray(chain[name][('alpha0', 'sn0')]), ours.arrays[name][..., 2])
            sampled = chain.sample_solved(size=1, seed=42)               # samples/chain.py:229-263 on OUR file (size = 1: for size > 1 the reference itself fails to broadcast its log-determinant term on 2-D chains)
            if ext == 'npz':
                for name in ['alpha0', 'sn0', 'loglikelihood', 'logprior', 'logposterior', 'qpar']:
                    out['sample_solved.' + name] = np.asarray(sampled[name])
                out['sample_solved.shape'] = np.array(sampled.shape)
            else:
                for name in ['alpha0', 'sn0', 'loglikelihood', 'logprior', 'logposterior']:
                    assert np.array_equal(out['sample_solved.' + name], np.asarray(sampled[name])), name
        # the reference writes the same chain: compare file structure with ours
        fn_ref = os.path.join(here, 'chain_reference_written.npz')
        chain = Chain.load(os.path.join(tmp, 'ours.npz')) if False else None
        ours.save(os.path.join(tmp, 'a.npz'))
        Chain.load(os.path.join(tmp, 'a.npz')).save(fn_ref)
        ref_raw, our_raw = dict(np.load(fn_ref, allow_pickle=True)), dict(np.load(os.path.join(tmp, 'a.npz'), allow_pickle=True))
        assert sorted(ref_raw) == sorted(our_raw), (sorted(ref_raw), sorted(our_raw))
        assert tuple(ref_raw['__class__']) == tuple(our_raw['__class__'])
        ro, oo = ref_raw['others'][()], our_raw['others'][()]
        assert sorted(ro) == sorted(oo), (sorted(ro), sorted(oo))
        for rp, op in zip(ref_raw['params'][()], our_raw['params'][()]):
            assert sorted(rp) == sorted(op) and sorted(rp['param']) == sorted(op['param']), (sorted(rp['param']), sorted(op['param']))
            assert rp['derivs'] == op['derivs'], (rp['derivs'], op['derivs'])
            for key in ['basename', 'namespace', 'fixed', 'derived']:
                assert rp['param'][key] == op['param'][key], (key, rp['param'][key], op['param'][key])
            for key in ['prior', 'ref']:
                assert rp['param'][key]['dist'] == op['param'][key]['dist'] and tuple(rp['param'][key]['limits']) == tuple(op['param'][key]['limits'])
        out['file_keys'] = np.array(sorted(ref_raw))
        out['others_keys'] = np.array(sorted(ro))
        out['param_state_keys'] = np.array(sorted(ref_raw['params'][()][0]['param']))
    for name, value in ours.arrays.items():
        out['input.' + name] = value
    out['solved'] = np.array(solved)
    np.savez(os.path.join(here, 'chain_io.npz'), **out)
    print('chain writer validated against the reference: Chain.load + Chain.sample_solved on our .npz / .npy; fixtures written')


if __name__ == '__main__':
    main()
