"""Convergence statistics of seeded chains computed by the REFERENCE's own functions (desilike/samples/diagnostics.py) on the split samples that
``BaseBatchPosteriorSampler.check`` builds (desilike/samplers/base.py:586-672).  Build container only:

    python tests/golden/make_diagnostics_fixture.py
"""
import os
import sys
import warnings

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, 'refstub'))
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')

from desilike.samples import Chain, diagnostics


def seeded_chains(nchains=3, niterations=240, nwalkers=6, ndim=3, seed=0):
    """AR(1) walkers around chain-dependent means, correlated parameters."""
    rng = np.random.RandomState(seed)
    mix = np.eye(ndim) + 0.3 * rng.standard_normal((ndim, ndim))
    chains = []
    for ichain in range(nchains):
        x = np.zeros((niterations, nwalkers, ndim))
        state = rng.standard_normal((nwalkers, ndim))
        for it in range(niterations):
            state = 0.8 * state + 0.6 * rng.standard_normal((nwalkers, ndim))
            x[it] = state
        chains.append(x.dot(mix.T) + 0.05 * ichain)
    return chains


def main():
    arrays = seeded_chains()
    names = ['a', 'b', 'c']
    chains = [Chain([x[..., i] for i in range(len(names))], params=names) for x in arrays]
    out = {'chains': np.array(arrays)}
    # the whole chains
    out['eigen_gr_full'] = diagnostics.gelman_rubin(chains, names, method='eigen', check_valid='ignore')
    out['diag_gr_full'] = diagnostics.gelman_rubin(chains, names, method='diag')
    # the split samples of check() (samplers/base.py:586-592): nsplits = 4, burnin = 0.5
    nsplits, burnin = 4, 0.5
    burnin = int(burnin * chains[0].shape[0] + 0.5)
    nsplits = int((nsplits + len(chains) - 1) / len(chains))
    lensplits = (chains[0].shape[0] - burnin) // nsplits
    split = [chain[burnin + islab * lensplits:burnin + (islab + 1) * lensplits] for islab in range(nsplits) for chain in chains]
    out['check_nsplits'], out['check_burnin'], out['check_lensplits'] = np.array(nsplits), np.array(burnin), np.array(lensplits)
    out['eigen_gr'] = diagnostics.gelman_rubin(split, names, method='eigen', check_valid='ignore')
    out['diag_gr'] = diagnostics.gelman_rubin(split, names, method='diag')
    out['geweke'] = diagnostics.geweke(split, names, first=0.1, last=0.5)
    walkers = []
    for chain in chains:   # samplers/base.py:646-651: one series per walker after burn-in
        chain = chain[burnin:]
        chain = chain.reshape(len(chain), -1)
        for iwalker in range(chain.shape[1]):
            walkers.append(chain[:, iwalker])
    out['iact'] = diagnostics.integrated_autocorrelation_time(walkers, names, check_valid='ignore')
    fn = os.path.join(here, 'diagnostics.npz')
    np.savez_compressed(fn, **out)
    print('saved', fn, {k: np.round(v, 4) for k, v in out.items() if k != 'chains'})


if __name__ == '__main__':
    main()
