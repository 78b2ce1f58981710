"""Validate the weighted chains of the Metropolis-Hastings sampler against the REFERENCE, in the build container only (the reference never travels to the GPU box):

    python tests/golden/validate_mh_chain.py

1. ``MCMCSampler`` (host driver, toy likelihood) writes its chains with ``save`` (desilike's ``Chain.save`` layout, multiplicities in ``fweight``);
2. the reference loads them with ``Chain.load``: columns, weighted mean and covariance agree;
3. the reference's weighted Gelman-Rubin and Geweke statistics (samples/diagnostics.py) on those chains are stored with the chains in ``mh_weighted_diagnostics.npz``:
   the fixture of tests/test_mcmc.py::test_weighted_diagnostics_against_the_reference.
"""
import os, sys, tempfile, warnings
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, 'refstub')); sys.path.insert(0, '/root/reference'); sys.path.insert(0, os.path.dirname(os.path.dirname(here))); sys.path.insert(0, os.path.dirname(here))
warnings.filterwarnings('ignore')
from test_samplers import ToyGaussianLikelihood
from desilike_amd.samplers import MCMCSampler
tmp = tempfile.mkdtemp()
s = MCMCSampler(ToyGaussianLikelihood(), chains=2, vectorize=2, seed=3, save_fn=os.path.join(tmp, 'c_*.npy'))
s.run(check_every=200, max_iterations=400)
from desilike.samples import Chain
for i in range(2):
    chain = Chain.load(os.path.join(tmp, 'c_{:d}.npy'.format(i)))
    mine = s.chains[i]
    assert np.array_equal(np.asarray(chain['a']), mine['a']) and np.array_equal(np.asarray(chain.fweight), mine['fweight']) and np.array_equal(np.asarray(chain.logposterior), mine['logposterior'])
    print(i, chain.shape, 'weighted mean a (reference Chain.mean):', float(chain.mean('a')), 'ours:', np.average(mine['a'], weights=mine['fweight']), 'cov', np.asarray(chain.covariance(['a','b'])).ravel())
    assert np.isclose(float(chain.mean('a')), np.average(mine['a'], weights=mine['fweight']))
    from desilike_amd.io import ChainFile
    ours = ChainFile.load(os.path.join(tmp, 'c_{:d}.npy'.format(i)))      # the statistics of desilike_amd.io.ChainFile against the reference's Chain
    assert np.isclose(ours.mean('a'), float(chain.mean('a')), rtol=1e-13) and np.allclose(ours.covariance(['a', 'b']), np.asarray(chain.covariance(['a', 'b'])), rtol=1e-12)
    assert np.isclose(ours.std('b'), float(chain.std('b')), rtol=1e-12)
# Gelman-Rubin of the reference on these weighted chains vs ours
from desilike.samples import diagnostics as rdiag
chains = [Chain.load(os.path.join(tmp, 'c_{:d}.npy'.format(i))) for i in range(2)]
ref_gr = rdiag.gelman_rubin(chains, ['a', 'b'], method='eigen')
ref_diag = rdiag.gelman_rubin(chains, ['a', 'b'], method='diag')
ref_geweke = rdiag.geweke(chains, ['a', 'b'], first=0.1, last=0.5)
from desilike_amd import diagnostics as diag
x = [np.column_stack([c['a'], c['b']]) for c in s.chains]; w = [c['fweight'] for c in s.chains]
print(ref_gr, diag.gelman_rubin(x, method='eigen', weights=w)); print(ref_diag, diag.gelman_rubin(x, method='diag', weights=w))
assert np.allclose(ref_gr, diag.gelman_rubin(x, method='eigen', weights=w), rtol=1e-10) and np.allclose(ref_diag, diag.gelman_rubin(x, method='diag', weights=w), rtol=1e-10)
print(np.asarray(ref_geweke), diag.geweke(x, weights=w))
assert np.allclose(np.asarray(ref_geweke), diag.geweke(x, weights=w), rtol=1e-10)
np.savez(os.path.join(here, 'mh_weighted_diagnostics.npz'), a0=s.chains[0]['a'], b0=s.chains[0]['b'], w0=s.chains[0]['fweight'], a1=s.chains[1]['a'], b1=s.chains[1]['b'], w1=s.chains[1]['fweight'],
         eigen_gr=ref_gr, diag_gr=ref_diag, geweke=np.asarray(ref_geweke))
print('ok')
