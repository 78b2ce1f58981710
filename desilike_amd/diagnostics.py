"""Convergence diagnostics of posterior chains, on plain arrays (reference: desilike/samples/diagnostics.py; used by ``BaseBatchPosteriorSampler.check``,
desilike/samplers/base.py:504-724).  A chain is an array ``[..., ndim]`` (iterations x walkers x parameters, or already flattened samples x parameters) with unit
weights: the samplers of this package do not produce weighted samples.

Pinned on outputs of the reference's own functions for seeded chains (tests/golden/diagnostics.npz, tests/golden/make_diagnostics_fixture.py)."""
import numpy as np

from . import utils


def _flat(chain):
    chain = np.asarray(chain, dtype='f8')
    return chain.reshape(-1, chain.shape[-1])


def gelman_rubin(chains, method='eigen', check_valid='raise', weights=None):
    """Gelman-Rubin statistics (Brooks & Gelman 1998) of two or more chains: covariance of the chain means ("between") against the mean of the chains' covariances
    ("within") -- samples/diagnostics.py:13-107.  ``method='eigen'``: eigenvalues of ``W^-1 V`` (after scaling by the standard deviations), else the diagonal ratio.
    Returns an array of size ndim (ascending eigenvalues for 'eigen').  ``weights``: per chain the integer multiplicities of its samples (Metropolis-Hastings chains;
    the reference's frequency weights, diagnostics.py:78-84: means and covariances are weighted, the chains count by their summed weights)."""
    chains = [_flat(chain) for chain in chains]
    nchains = len(chains)
    if nchains < 2:
        raise ValueError('Provide at least 2 chains to estimate Gelman-Rubin')
    sizes = np.array([chain.shape[0] for chain in chains], dtype='f8')
    if (sizes < 2).any():
        raise ValueError('Not enough samples ({}) to estimate Gelman-Rubin'.format(sizes))
    if weights is None:
        means = np.array([chain.mean(axis=0) for chain in chains])
        covs = np.array([np.atleast_2d(np.cov(chain, rowvar=False, ddof=1)) for chain in chains])
    else:
        weights = [np.asarray(weight, dtype='i8').ravel() for weight in weights]
        means = np.array([np.average(chain, weights=weight, axis=0) for chain, weight in zip(chains, weights)])
        covs = np.array([np.atleast_2d(np.cov(chain, rowvar=False, fweights=weight, ddof=1)) for chain, weight in zip(chains, weights)])
        sizes = np.array([weight.sum() for weight in weights], dtype='f8')      # wsum = w2sum for frequency weights
    # unit weights: wsum = w2sum = size (diagnostics.py:80-84)
    Wn1 = np.average(covs, weights=sizes, axis=0)
    Wn = np.average(((sizes - 1.) / sizes)[:, None, None] * covs, weights=sizes, axis=0)
    B = np.atleast_2d(np.cov(means.T, ddof=1))          # not weighted by the chains' lengths: short chains should stand out (diagnostics.py:86-88)
    V = Wn + (nchains + 1.) / nchains * B
    if method == 'eigen':
        stddev = np.sqrt(np.diag(V).real)
        V = V / stddev[:, None] / stddev[None, :]
        invWn1 = utils.inv(Wn1 / stddev[:, None] / stddev[None, :], check_valid=check_valid)
        if invWn1 is None:
            raise ValueError('cannot compute inverse')
        try:
            return np.linalg.eigvalsh(invWn1.dot(V))
        except np.linalg.LinAlgError as exc:
            raise ValueError from exc
    return np.diag(V) / np.diag(Wn1)


def _autocorrelation_1d(x):
    """Normalised autocorrelation function by FFT (emcee's estimator; diagnostics.py:279-303)."""
    x = np.atleast_1d(x)
    if x.ndim != 1 or x.size < 2:
        raise ValueError('Not enough samples to estimate autocorrelation')
    n = 2**(2 * len(x) - 1).bit_length()
    f = np.fft.fft(x, n=n)
    acf = np.fft.ifft(f * np.conjugate(f))[:len(x)].real
    return acf / acf[0]


def autocorrelation(series):
    """Mean autocorrelation of a list of 1-D series (one per walker), each centred on its own mean (diagnostics.py:108-141); ``series`` as ``[nseries, nsamples]``
    -> ``[nsamples]``, or ``[nseries, nsamples, ndim]`` -> ``[ndim, nsamples]``."""
    series = np.asarray(series, dtype='f8')
    if series.ndim == 3:
        return np.array([autocorrelation(series[..., idim]) for idim in range(series.shape[-1])])
    return sum(_autocorrelation_1d(x - x.mean()) for x in series) / len(series)


def integrated_autocorrelation_time(series, c=5.):
    """Integrated autocorrelation time with Sokal's automated window (diagnostics.py:144-276, criterion 'sokal'): ``tau = 2 sum_{t <= N} rho_t - 1``, N the first
    index with ``N >= c tau_N``.  ``series [nseries, nsamples(, ndim)]`` (e.g. one series per walker); returns a scalar or an array of size ndim."""
    series = np.asarray(series, dtype='f8')
    if series.shape[1] < 2:
        raise ValueError('Not enough samples ({:d}) to estimate autocorrelation time'.format(series.shape[1]))
    if series.ndim == 3:
        return np.array([integrated_autocorrelation_time(series[..., idim], c=c) for idim in range(series.shape[-1])])
    taus = 2. * np.cumsum(autocorrelation(series)) - 1.
    mask = np.arange(len(taus)) < c * taus
    window = np.argmin(mask) if np.any(mask) else len(taus) - 1
    return taus[window]


def geweke(chains, first=0.1, last=0.5, weights=None):
    """Geweke statistics: difference of the means of the first ``first`` and the last ``1 - last`` fractions of each chain over the root of the summed variances
    (diagnostics.py:306-343); returns ``[ndim, nchains]``."""
    out = []
    for ichain, chain in enumerate(chains):
        chain = _flat(chain)
        size = chain.shape[0]
        ifirst, ilast = int(first * size + 0.5), int(last * size + 0.5)
        head, tail = chain[:ifirst], chain[ilast:]
        if head.shape[0] < 2 or tail.shape[0] < 2:
            raise ValueError('Not enough samples ({:d}) to estimate geweke'.format(size))
        if weights is None:
            out.append(np.abs(head.mean(axis=0) - tail.mean(axis=0)) / (head.var(axis=0, ddof=1) + tail.var(axis=0, ddof=1))**0.5)
        else:   # frequency weights (diagnostics.py:330-339)
            weight = np.asarray(weights[ichain], dtype='i8').ravel()
            whead, wtail = weight[:ifirst], weight[ilast:]
            var = [np.diag(np.atleast_2d(np.cov(part, rowvar=False, fweights=w, ddof=1))) for part, w in ((head, whead), (tail, wtail))]
            out.append(np.abs(np.average(head, weights=whead, axis=0) - np.average(tail, weights=wtail, axis=0)) / (var[0] + var[1])**0.5)
    return np.array(out).T


class Diagnostics(dict):
    """History of the convergence statistics (samplers/base.py:733-791): ``add`` appends a value under a key; ``add_test`` also appends the outcome of the test
    ``low < value < up`` under ``<key>_test`` and returns whether it held over the last ``stable_over`` calls."""

    def add(self, key, value):
        self.setdefault(key, []).append(value)
        return value

    def is_stable(self, key, stable_over=2):
        return len(self[key]) >= stable_over and all(self[key][-stable_over:])

    def add_test(self, key, name, value, limits=None, stable_over=2, quiet=True, log=None):
        self.add(key, value)
        low, up = limits if limits is not None else (None, None)
        if low is None and up is None:
            if not quiet and log is not None: log('- {} is {:.3g}.'.format(name, value))
            return True
        test = (low is None or value > low) and (up is None or value < up)     # (a NaN statistic fails any test)
        if not quiet and log is not None:
            bounds = '> {:.3g}'.format(low) if up is None else '< {:.3g}'.format(up) if low is None else 'in [{:.3g}, {:.3g}]'.format(low, up)
            log('- {} is {:.3g}; {}{}.'.format(name, value, '' if test else 'not ', bounds))
        self.add(key + '_test', bool(test))
        return self.is_stable(key + '_test', stable_over=stable_over)
