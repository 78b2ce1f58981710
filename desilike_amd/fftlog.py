r"""FFTLog Hankel transform P_\ell(k) -> \xi_\ell(s), init-time host code.

In the reference this is the third-party ``cosmoprimo.PowerToCorrelation(k, ell, q=0, lowring=True)`` (call sites
theories/galaxy_clustering/base.py:76-77, 135), absent here and not pinned by any reference test ("parity unpinned"): this module is
our own implementation of the public algorithm (Hamilton 2000, "FFTLog"), checked against ``scipy.fft.fht`` (oracle) and against the
brute-force integral the reference itself uses as a cross-check (theories/galaxy_clustering/base.py:163-168).

    \xi_\ell(s) = (-1)^{\ell/2} / (2 \pi^2) \int dk k^2 P_\ell(k) j_\ell(k s)
                = (-1)^{\ell/2} (2 \pi)^{-3/2} s^{-3/2} \int d\ln k [k^{3/2} P_\ell(k)] (k s) J_{\ell + 1/2}(k s)

Because every step of the reference's ``get_corr`` (theories/galaxy_clustering/base.py:127-136: linear interpolation of P_\ell in log k onto
the FFTLog grid, linear-in-log high-k tail with Gaussian damping, FFTLog, linear interpolation to the data separations) is LINEAR in
P_\ell(k_in), the whole map is one constant matrix per multipole (``hankel_operator``) which the GPU path folds into the window matrix:
the FFT never runs in the hot loop.

The transform itself runs on the device: the batched LDS-resident FFTLog of the C ABI (``dl_fftlog_*``, csrc/dl_fftlog.hip: one workgroup per (point, multipole)) --
what the theory classes use to build their operator (all unit vectors of the input grid in ONE batch) and what transforms batches of P_ell already resident on the
GPU.  This module holds the grid constants (Mellin coefficients, low-ringing offset, pre / post factors) and the call surface; there is NO host engine here: without the
library or a GPU every call raises.  The ``numpy.fft`` restatement of the same algorithm that the CPU checks use lives with the oracle (``oracle/np_fftlog.py``, test
infrastructure: tests/test_oracle_bao.py, tests/test_gpu_fftlog.py).
"""
import numpy as np
from scipy import special

LN_2 = np.log(2.)


def fftlog_offset(dln, mu, initial=0., bias=0.):
    """Low-ringing offset ln(k_c s_c) closest to ``initial`` (Hamilton 2000, eq. 186): makes the Nyquist coefficient real."""
    xp, xm = (mu + 1. + bias) / 2., (mu + 1. - bias) / 2.
    y = np.pi / (2. * dln)
    zp, zm = special.loggamma(xp + 1j * y), special.loggamma(xm + 1j * y)
    arg = (LN_2 - initial) / dln + (zp.imag + zm.imag) / np.pi
    return initial + (arg - np.round(arg)) * dln


def fftlog_coefficients(n, dln, mu, offset=0.):
    """u_m, m = 0 .. n/2: Mellin transform of z J_mu(z) on the imaginary axis times the phase of the output-grid offset (unbiased, q = 0)."""
    y = np.pi * np.arange(n // 2 + 1) / (n * dln)
    xh = (mu + 1.) / 2.
    phase = 2. * special.loggamma(xh + 1j * y).imag + 2. * y * (LN_2 - offset)
    u = np.exp(1j * phase)
    u.imag[-1] = 0.   # Nyquist term real
    return u


class PowerToCorrelation(object):
    """Drop-in for the call surface ``PowerToCorrelation(k, ell=ells, q=0, lowring=True)(pk[n_ell, N]) -> (s[n_ell, N], xi[n_ell, N])``.

    ``k`` must be log-spaced; the input is zero-padded to ``minfolds * N`` points (half on each side) before the transform.
    """

    def __init__(self, k, ell=0, q=0, lowring=True, minfolds=2, engine='hip', device=None):
        if engine != 'hip':
            raise ValueError('engine must be "hip": the transform runs on the device (the NumPy restatement is test infrastructure: oracle/np_fftlog.py)')
        self.engine, self.device, self._plan = engine, device, None
        self.k = np.asarray(k, dtype='f8')
        self.ells = np.atleast_1d(ell)
        if q != 0:
            raise NotImplementedError('only the unbiased transform q = 0 is implemented')
        n = self.k.size
        self.dln = np.log(self.k[-1] / self.k[0]) / (n - 1)
        if not np.allclose(np.diff(np.log(self.k)), self.dln, rtol=1e-8):
            raise ValueError('k must be log-spaced')
        self.npad = int(2**np.ceil(np.log2(minfolds * n)))
        self.pad = (self.npad - n) // 2
        self.kpad = self.k[0] * np.exp(self.dln * (np.arange(self.npad) - self.pad))
        self.offsets, self.u, self.s = [], [], []
        for ell in self.ells:
            mu = ell + 0.5
            offset = fftlog_offset(self.dln, mu) if lowring else 0.
            self.offsets.append(offset)
            self.u.append(fftlog_coefficients(self.npad, self.dln, mu, offset=offset))
            # output grid: s_j k_{npad - 1 - j} = exp(offset)
            self.s.append(np.exp(offset) / self.kpad[::-1])
        self.prefactor = [(-1.)**(ell // 2) / (2. * np.pi)**1.5 for ell in self.ells]

    def _get_plan(self):
        """Device plan (``dl_fftlog_create``): uploads k^{3/2}, u_ell and prefactor * s^{-3/2} once."""
        if self._plan is None:
            import os
            from ._lib import FFTLogPlan
            device = self.device
            if device is None: device = int(os.environ.get('LOCAL_RANK', 0))
            sl = slice(self.pad, self.pad + self.k.size)
            u = np.array([np.column_stack([u.real, u.imag]) for u in self.u], dtype='f8')
            post = np.array([prefactor * s[sl]**(-1.5) for prefactor, s in zip(self.prefactor, self.s)], dtype='f8')
            self._plan = FFTLogPlan(self.k.size, self.npad, self.k**1.5, u, post, device=device)
        return self._plan

    def close(self):
        """Release the device plan (if any)."""
        if self._plan is not None:
            self._plan.close()
            self._plan = None

    def apply_device(self, fun, out=None, stream=None):
        """``fun [B, n_ell, N]`` CUDA(ROCm) torch tensor -> ``xi [B, n_ell, N]`` on the same device, asynchronous on ``stream`` (no host round trip)."""
        return self._get_plan().apply(fun, out=out, stream=stream)

    def __call__(self, fun):
        import torch
        plan = self._get_plan()
        fun = np.asarray(fun, dtype='f8')
        shape = fun.shape
        fun = fun.reshape((-1,) + (len(self.ells), self.k.size))
        xi = plan.apply(torch.as_tensor(fun, dtype=torch.float64, device=torch.device('cuda', plan.device)).contiguous()).cpu().numpy()
        sl = slice(self.pad, self.pad + self.k.size)
        return np.array([s[sl] for s in self.s]), xi.reshape(shape)


def _interp_to_grid(logk, logkin, pk, interp_order=1):
    """P_ell from the theory grid to the FFTLog grid, in log10 k: linear (``interp_order = 1``) or the numpy-backend cubic of the reference's ``interp1d``
    (``interp_order = 3``: scipy not-a-knot cubic with extrapolation, desilike/jax.py:263-265)."""
    if interp_order == 1:
        return np.interp(logk, logkin, pk)
    from scipy import interpolate
    return interpolate.interp1d(logkin, pk, kind='cubic', fill_value='extrapolate', axis=0)(logk)


def hankel_operator(kin, s, ells, k=None, engine='hip', device=None, interp_order=1):
    r"""Matrices H_\ell [len(s), len(kin)] with \xi_\ell(s) = H_\ell P_\ell(k_in), reproducing the reference's ``get_corr`` grids
    (theories/galaxy_clustering/base.py:62-77: k = logspace(-4, 3, 2048), tail beyond kin[-1]).

    The transforms of all len(kin) unit vectors run as ONE batch of the device FFTLog (``dl_fftlog_apply``; ``engine`` must be 'hip').
    ``interp_order``: 1 (linear) or 3 (cubic) interpolation of P_ell to the FFTLog grid (tgc/base.py:54-57, 132) -- either is linear in P_ell, so it folds
    into the operator."""
    kin = np.asarray(kin, dtype='f8')
    if k is None: k = np.logspace(-4., 3., 2048)
    mask = k > kin[-1]
    logk_high = np.log10(k[mask] / kin[-1])
    damp_high = np.exp(-(k[mask] / kin[-1] - 1.)**2 / (2. * (10.)**2))
    fftlog = PowerToCorrelation(k, ell=ells, q=0, lowring=True, engine=engine, device=device)
    nell = len(ells)
    # interpolation + tail of every unit vector (the same for all multipoles), then one batched transform, then the interpolation to s
    logkin, logk_mid = np.log10(kin), np.log10(k[~mask])
    unit = np.zeros(kin.size, dtype='f8')
    tmp = np.empty((kin.size, k.size), dtype='f8')
    for i in range(kin.size):
        unit[i] = 1.
        slope_high = (unit[-1] - unit[-2]) / np.log10(kin[-1] / kin[-2])
        tmp[i] = np.concatenate([_interp_to_grid(logk_mid, logkin, unit, interp_order=interp_order), (unit[-1] + slope_high * logk_high) * damp_high])
        unit[i] = 0.
    ss, corr = fftlog(np.repeat(tmp[:, None, :], nell, axis=1))     # corr [n_kin, n_ell, N]
    fftlog.close()   # free the device plan now (a hipFree deferred to the garbage collector would synchronise the device at an arbitrary later time)
    H = np.empty((nell, len(s), kin.size), dtype='f8')
    for ill in range(nell):
        for i in range(kin.size):
            H[ill, :, i] = np.interp(s, ss[ill], corr[i, ill])
    return H
