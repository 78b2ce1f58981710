"""NumPy oracle: CPU restatement of desilike's theory -> observable -> Gaussian-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product path (``desilike_amd``) never does.

Every function cites the reference file:line (relative to /root/reference/desilike) it restates.
Parity pin: checked against golden vectors captured from the reference's own numpy path
(``tests/golden/*.npz``, produced by ``tests/golden/make_golden.py``) in ``tests/test_oracle.py``
for rows a1-a9, a14 of SURVEY.md section 8.  Rows that the reference cannot run here are marked
"parity unpinned" where they are defined (FFTLog: third-party cosmoprimo absent; analytic
marginalisation: needs jax; emulator forward: third-party) and are pinned by closed-form /
brute-force identities instead.

Everything is float64.
"""
import numpy as np
from scipy import interpolate, special


# ----------------------------------------------------------------------------------------------
# a4: mu-quadrature                                            utils.py:625-643, tgc/base.py:201-208
# ----------------------------------------------------------------------------------------------
def weights_leggauss_sym(nmu):
    """Gauss-Legendre nodes on (0, 1): leggauss(2n), positive half, symmetrised weights (sum = 1).

    utils.py:625-630 (``weights_leggauss(nx, sym=True)``), used by ``weights_mu`` utils.py:633-643.
    """
    x, w = np.polynomial.legendre.leggauss(2 * nmu)
    return x[nmu:], (w[nmu:] + w[nmu - 1::-1]) / 2.


def multipole_weights(mu, wmu, ells):
    """``wmu[ell, m] = w_m (2 ell + 1) L_ell(mu_m)``; tgc/base.py:201-204."""
    return np.array([wmu * (2 * ell + 1) * special.legendre(ell)(mu) for ell in ells])


# ----------------------------------------------------------------------------------------------
# a1: Alcock-Paczynski                                               tgc/base.py:211-223, 325-353
# ----------------------------------------------------------------------------------------------
def ap_qparqper(mode, eta, **params):
    """(qpar, qper) from the AP parameterisation; tgc/base.py:341-350."""
    if mode == 'qiso':
        return params['qiso'], params['qiso']
    if mode == 'qap':
        qap = params['qap']
        return qap**(1 - eta), qap**(-eta)
    if mode == 'qisoqap':
        qiso, qap = params['qiso'], params['qap']
        return qiso * qap**(1 - eta), qiso * qap**(-eta)
    return params['qpar'], params['qper']


def ap_k_mu(k, mu, qpar=1., qper=1.):
    """tgc/base.py:211-223: returns jac, kap[k, mu], muap[k, mu]."""
    qap = qpar / qper
    jac = 1. / (qpar * qper**2)
    factorap = np.sqrt(1 + mu**2 * (1. / qap**2 - 1))
    kap = k[:, None] / qper * factorap
    muap = mu / qap / factorap
    return jac, kap, muap * np.ones_like(kap)


# ----------------------------------------------------------------------------------------------
# a2: templates                                 power_template.py:747-761, 592-596, 372-376, 198-202
# ----------------------------------------------------------------------------------------------
def turnover_pk(k, kTO_fid, pkTO_fid, m=0.6, n=0.9, qto=1., dpto=1.):
    """power_template.py:1326-1333: x = log10 k / log10 k_TO - 1; P_TO^(1 - m x^2) where x > 0, P_TO^(1 - n x^2) elsewhere."""
    kTO, pkTO = kTO_fid * qto, pkTO_fid * dpto
    x = np.log10(k) / np.log10(kTO) - 1.
    return np.where(x > 0., pkTO**(1. - m * x**2), pkTO**(1. - n * x**2))


def band_templates(k, kp):
    """power_template.py:931-939: tent functions of the theory wavenumbers around the pivots ``kp`` (the first and last wavenumber close the outer tents)."""
    k, kp = np.asarray(k, dtype='f8'), np.asarray(kp, dtype='f8')
    ekp = np.concatenate([[k[0]], kp, [k[-1]]])
    out = []
    for ip, pivot in enumerate(kp):
        diff = k - pivot
        neg = diff < 0
        diff[neg] /= (ekp[ip] - pivot)
        diff[~neg] /= (ekp[ip + 2] - pivot)
        out.append(np.maximum(1. - diff, 0.))
    return np.array(out)


def shapefit_factor(k, kp, a, dm=0., dn=0.):
    """power_template.py:749: exp(dm / a tanh(a ln(k / kp)) + dn ln(k / kp))."""
    return np.exp(dm / a * np.tanh(a * np.log(k / kp)) + dn * np.log(k / kp))


# ----------------------------------------------------------------------------------------------
# a3: interp1d                                                                    jax.py:211-265
# ----------------------------------------------------------------------------------------------
def interp1d(xq, x, f, method='cubic'):
    """numpy-backend branch jax.py:263-265: scipy not-a-knot cubic (or linear) with extrapolation."""
    method = {1: 'linear', 3: 'cubic'}.get(method, method)
    return interpolate.interp1d(x, f, kind=method, fill_value='extrapolate', axis=0)(xq)


def notaknot_moments(x, y):
    """Second derivatives M of the not-a-knot cubic spline through (x, y) (independent formulation).

    This is NOT how scipy computes it (B-spline collocation); the interpolant is unique, so the
    piecewise-cubic "moment" form used by the device kernel must agree to rounding.  Used to check
    the kernel's spline formulation on CPU (``notaknot_eval`` vs ``interp1d``).
    """
    n = len(x)
    h = np.diff(x)
    A = np.zeros((n, n))
    r = np.zeros(n)
    for i in range(1, n - 1):
        A[i, i - 1], A[i, i], A[i, i + 1] = h[i - 1], 2. * (h[i - 1] + h[i]), h[i]
        r[i] = 6. * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
    A[0, 0], A[0, 1], A[0, 2] = h[1], -(h[0] + h[1]), h[0]
    A[-1, -3], A[-1, -2], A[-1, -1] = h[-1], -(h[-2] + h[-1]), h[-2]
    return np.linalg.solve(A, r)


def notaknot_eval(xq, x, y, M):
    i = np.clip(np.searchsorted(x, xq, side='right') - 1, 0, len(x) - 2)
    h = x[i + 1] - x[i]
    a = (x[i + 1] - xq) / h
    b = (xq - x[i]) / h
    return a * y[i] + b * y[i + 1] + ((a**3 - a) * M[i] + (b**3 - b) * M[i + 1]) * h**2 / 6.


# ----------------------------------------------------------------------------------------------
# a4: Kaiser P(k, mu) -> multipole tables                          full_shape.py:488-500
# ----------------------------------------------------------------------------------------------
def kaiser_pktable(k, mu, wmu_ell, k11, pk11, f, qpar=1., qper=1., sigmapar=0., sigmaper=0.):
    """Returns pk_dd, pk_dt, pk_tt, each [n_ell, n_k]; full_shape.py:488-500, to_poles tgc/base.py:206-208."""
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    sigmanl2 = kap**2 * (sigmapar**2 * muap**2 + sigmaper**2 * (1. - muap**2))
    damping = np.exp(-sigmanl2 / 2.)
    pktable = jac * damping * interp1d(np.log10(kap), np.log10(k11), pk11, method='cubic')

    def to_poles(pkmu):
        return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)

    return to_poles(pktable), to_poles(f * muap**2 * pktable), to_poles(f**2 * muap**4 * pktable)


def simple_tracer_power(k, mu, wmu_ell, k11, pk11, f, nd, b1X, b1Y, sn0, qpar=1., qper=1., sigmapar=0., sigmaper=0.):
    """SimpleTracerPowerSpectrumMultipoles.calculate, full_shape.py:405-414: damping at the FIDUCIAL (k, mu), sn0 / nd added to P(k, mu) before the projection."""
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    sigmanl2 = k[:, None]**2 * (sigmapar**2 * mu**2 + sigmaper**2 * (1. - mu**2))
    damping = np.exp(-sigmanl2 / 2.)
    pkmu = jac * damping * (b1X + f * muap**2) * (b1Y + f * muap**2) * interp1d(np.log10(kap), np.log10(k11), pk11, method='cubic') + sn0 / nd
    return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)


# ----------------------------------------------------------------------------------------------
# f2: TNS one-loop tables (the reference's own in-repo PT producer)        full_shape.py:688-899
# ----------------------------------------------------------------------------------------------
def weights_trapz(x):
    """Trapezoidal weights, utils.py:620-622 (``jnp.insert`` with jax's index clamping: the last index lands at the end)."""
    x = np.asarray(x, dtype='f8')
    return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.


def tns_k11(k):
    """Wavenumbers of the loop tables, full_shape.py:875."""
    return np.linspace(k[0] * 0.7, k[-1] * 1.3, int(len(k) * 1.6 + 0.5))


def tns_kernels(k, q, wq):
    """Angle-integrated 13-type kernels (density, velocity) and the A-term kernel, [n_k, n_q] / [5, n_k, n_q]; full_shape.py:688-746."""
    jq = q**2 * wq / (4. * np.pi**2)
    x = q / k[:, None]

    def series13(x, poly, num, far, near):
        # closed form, its large-x expansion (x > 10) and its expansion around x = 1 (|x - 1| < 0.01): full_shape.py:695-716
        lg = 2. * np.log(np.abs((x - 1.) / (x + 1.)))
        out = (6. / x**2 + poly[0] + poly[1] * x**2 + poly[2] * x**4 + 0.75 * (1. / x - x)**3 * (2. + poly[3] * x**2) * lg) / num
        m = x > 10.
        out[m] = far[0] + far[1] / x[m]**2 + far[2] / x[m]**4
        dx = x - 1.
        m = np.abs(dx) < 0.01
        out[m] = near[0] + near[1] * dx[m] + near[2] * dx[m]**2
        return out / x**2

    with np.errstate(divide='ignore', invalid='ignore'):
        ff = series13(x, (-79., 50., -21., 7.), 504., (-61. / 630., 2. / 105., -10. / 1323.), (-11. / 126., 1. / 126., -29. / 252.))
        gg = series13(x, (-41., 2., -3., 1.), 168., (-3. / 10., 26. / 245., -38. / 2205.), (-3. / 14., -5. / 42., -1. / 84.))
        ka = np.zeros((5,) + x.shape, dtype='f8')    # full_shape.py:721-744
        lx = np.zeros_like(x)
        m = np.abs(x - 1.) > 1e-16
        lx[m] = np.log(np.abs((x[m] + 1.) / (x[m] - 1.)))
        ka[0] = -1. / 84. / x * (2 * x * (19 - 24 * x**2 + 9 * x**4) - 9 * (x**2 - 1)**3 * lx)
        ka[1] = 1. / 112. / x**3 * (2 * x * (x**2 + 1) * (3 - 14 * x**2 + 3 * x**4) - 3 * (x**2 - 1)**4 * lx)
        ka[2] = 1. / 336. / x**3 * (2 * x * (9 - 185 * x**2 + 159 * x**4 - 63 * x**6) + 9 * (x**2 - 1)**3 * (7 * x**2 + 1) * lx)
        ka[4] = 1. / 336. / x**3 * (2 * x * (9 - 109 * x**2 + 63 * x**4 - 27 * x**6) + 9 * (x**2 - 1)**3 * (3 * x**2 + 1) * lx)
    m = x < 1e-4
    xm = x[m]
    ka[0][m] = 8 * xm**8 / 735 + 24 * xm**6 / 245 - 24 * xm**4 / 35 + 8 * xm**2 / 7 - 2. / 3
    ka[1][m] = -16 * xm**8 / 8085 - 16 * xm**6 / 735 + 48 * xm**4 / 245 - 16 * xm**2 / 35
    ka[2][m] = 32 * xm**8 / 1617 + 128 * xm**6 / 735 - 288 * xm**4 / 245 + 64 * xm**2 / 35 - 4. / 3
    ka[4][m] = 24 * xm**8 / 2695 + 8 * xm**6 / 105 - 24 * xm**4 / 49 + 24 * xm**2 / 35 - 2. / 3
    m = x > 1e2
    xm = x[m]
    ka[0][m] = 2. / 105 - 24 / (245 * xm**2) - 8 / (735 * xm**4) - 8 / (2695 * xm**6) - 8 / (7007 * xm**8)
    ka[1][m] = -16. / 35 + 48 / (245 * xm**2) - 16 / (735 * xm**4) - 16 / (8085 * xm**6) - 16 / (35035 * xm**8)
    ka[2][m] = -44. / 105 - 32 / (735 * xm**4) - 64 / (8085 * xm**6) - 96 / (35035 * xm**8)
    ka[4][m] = -46. / 105 + 24 / (245 * xm**2) - 8 / (245 * xm**4) - 8 / (1617 * xm**6) - 8 / (5005 * xm**8)
    ka[3] = ka[1]
    return 2 * jq * ff, 2 * jq * gg, jq * ka / x**2


TNS_NAMES = ['pk11', 'pk_dd', 'pk_b2d', 'pk_bs2d', 'pk_sig3sq', 'pk_b22', 'pk_b2s2', 'pk_bs22', 'pk_dt', 'pk_b2t', 'pk_bs2t', 'pk_tt']   # full_shape.py:882


def tns_geometry_A(x, mu):
    """The ten polynomial kernels of the A term, Taruya et al. 2010 eq. A3 as coded in full_shape.py:800-810."""
    kA = [-x**3 / 7. * (mu + 6 * mu**3 + x**2 * mu * (-3 + 10 * mu**2) + x * (-3 + mu**2 - 12 * mu**4)),
          x**4 / 14. * (mu**2 - 1) * (-1 + 7 * x * mu - 6 * mu**2),
          x**3 / 14. * (x**2 * mu * (13 - 41 * mu**2) - 4 * (mu + 6 * mu**3) + x * (5 + 9 * mu**2 + 42 * mu**4)),
          None,
          x**3 / 14. * (1 - 7 * x * mu + 6 * mu**2) * (-2 * mu + x * (-1 + 3 * mu**2))]
    kA[3] = kA[1]
    ktA = [1. / 7. * (mu + x - 2 * x * mu**2) * (3 * x + 7 * mu - 10 * x * mu**2),
           x / 14. * (mu**2 - 1) * (3 * x + 7 * mu - 10 * x * mu**2),
           1. / 14. * (28 * mu**2 + x * mu * (25 - 81 * mu**2) + x**2 * (1 - 27 * mu**2 + 54 * mu**4)),
           x / 14. * (1 - mu**2) * (x - 7 * mu + 6 * x * mu**2),
           1. / 14. * (x - 7 * mu + 6 * x * mu**2) * (-2 * mu - x + 3 * x * mu**2)]
    return kA, ktA


def tns_geometry_B(x, mu, xmu):
    """The twelve polynomial kernels of the B term (n, a, b as commented in full_shape.py:813-826), WITHOUT the common 1 / (x^2 xmu)."""
    m21 = mu**2 - 1.
    return [x**2 * m21 / 2.,
            3. * x**2 * m21**2 / 8.,
            3. * x**4 * m21**2 / xmu / 8.,
            5. * x**4 * m21**3 / xmu / 16.,
            x * (x + 2. * mu - 3. * x * mu**2) / 2.,
            -3. * x * m21 * (-x - 2. * mu + 5. * x * mu**2) / 4.,
            3. * x**2 * m21 * (-2. + x**2 + 6. * x * mu - 5. * x**2 * mu**2) / xmu / 4.,
            -3. * x**2 * m21**2 * (6. - 5. * x**2 - 30. * x * mu + 35. * x**2 * mu**2) / xmu / 16.,
            x * (4. * mu * (3. - 5. * mu**2) + x * (3. - 30. * mu**2 + 35. * mu**4)) / 8.,
            x * (-8. * mu + x * (-12. + 36. * mu**2 + 12. * x * mu * (3. - 5. * mu**2) + x**2 * (3. - 30. * mu**2 + 35. * mu**4))) / xmu / 8.,
            3. * x * m21 * (-8. * mu + x * (-12. + 60. * mu**2 + 20. * x * mu * (3. - 7. * mu**2) + 5. * x**2 * (1. - 14. * mu**2 + 21. * mu**4))) / xmu / 16.,
            x * (8. * mu * (-3. + 5. * mu**2) - 6. * x * (3. - 30. * mu**2 + 35. * mu**4) + 6. * x**2 * mu * (15. - 70. * mu**2 + 63 * mu**4)
                 + x**3 * (5. - 21. * mu**2 * (5. - 15. * mu**2 + 11. * mu**4))) / xmu / 16.]


def tns_pt(k11, q, wq, pk_q, kernels=None, nmu=10):
    """One-loop tables of the TNS model on ``k11``: dict of the twelve spectra of ``TNS_NAMES`` [n_k] + 'A' [5, n_k] + 'B' [12, n_k].

    full_shape.py:749-833: P22-type integrals over q (trapezoid on the template's own k grid) and the cosine mu (10 Gauss-Legendre nodes on (0, 1)),
    P(|k - q|) by LINEAR interpolation of the template with zero outside its range; P13-type terms from the precomputed kernels.
    """
    if kernels is None: kernels = tns_kernels(k11, q, wq)
    k13d, k13t, ka = kernels
    k = k11[:, None]
    jq = q**2 * wq / (4. * np.pi**2)
    x = q / k
    mus, wmus = weights_leggauss_sym(nmu)
    pk_k = np.interp(k11, q, pk_q)
    out = {name: 0. for name in ['b2d', 'bs2d', 'b2t', 'bs2t', 'sig3sq', 'b22', 'b2s2', 'bs22', '22dd', '22dt', '22tt']}
    A, B = np.zeros((5, len(k11))), np.zeros((12, len(k11)))
    for mu, wmu in zip(mus, wmus):
        kdq = k * q * mu
        kq2 = k**2 - 2. * kdq + q**2
        qdkq = kdq - q**2
        c2 = qdkq**2 / (q**2 * kq2)
        half = 0.5 * qdkq * (1. / q**2 + 1. / kq2)
        F2d = 5. / 7. + half + 2. / 7. * c2
        F2t = 3. / 7. + half + 4. / 7. * c2
        S = c2 - 1. / 3.
        D = 2. / 7. * (mu**2 - 1.)
        pk_kq = np.interp(np.sqrt(kq2), q, pk_q, left=0., right=0.)
        pp = jq * pk_q * pk_kq
        out['b2d'] += wmu * np.sum(pp * F2d, axis=-1)
        out['bs2d'] += wmu * np.sum(pp * F2d * S, axis=-1)
        out['b2t'] += wmu * np.sum(pp * F2t, axis=-1)
        out['bs2t'] += wmu * np.sum(pp * F2t * S, axis=-1)
        out['sig3sq'] += wmu * np.sum(105. / 16. * jq * pk_q * (D * S + 8. / 63.), axis=-1)
        out['b22'] += wmu / 2. * np.sum(jq * pk_q * (pk_kq - pk_q), axis=-1)
        out['b2s2'] += wmu / 2. * np.sum(jq * pk_q * (pk_kq * S - 2. / 3. * pk_q), axis=-1)
        out['bs22'] += wmu / 2. * np.sum(jq * pk_q * (pk_kq * S**2 - 4. / 9. * pk_q), axis=-1)
        out['22dd'] += 2 * wmu * np.sum(F2d**2 * pp, axis=-1)
        out['22dt'] += 2 * wmu * np.sum(F2d * F2t * pp, axis=-1)
        out['22tt'] += 2 * wmu * np.sum(F2t**2 * pp, axis=-1)
        xmu = kq2 / k**2
        kA, ktA = tns_geometry_A(x, mu)
        for i in range(5):
            A[i] += wmu * np.sum(jq / x**2 * (kA[i] * pk_k[:, None] + ktA[i] * pk_q) * pk_kq / xmu**2, axis=-1)
        ppb = pp / (x**2 * xmu)
        for i, cb in enumerate(tns_geometry_B(x, mu, xmu)):
            B[i] += wmu * np.sum(cb * ppb, axis=-1)
    A += pk_k * np.sum(ka * pk_q, axis=-1)
    pk13_dd = 2. * np.sum(k13d * pk_q, axis=-1) * pk_k
    pk13_tt = 2. * np.sum(k13t * pk_q, axis=-1) * pk_k
    pk13_dt = (pk13_dd + pk13_tt) / 2.
    tab = {'pk11': pk_k, 'pk_dd': pk_k + out['22dd'] + pk13_dd, 'pk_b2d': out['b2d'], 'pk_bs2d': out['bs2d'], 'pk_sig3sq': out['sig3sq'] * pk_k,
           'pk_b22': out['b22'], 'pk_b2s2': out['b2s2'], 'pk_bs22': out['bs22'], 'pk_dt': pk_k + out['22dt'] + pk13_dt, 'pk_b2t': out['b2t'],
           'pk_bs2t': out['bs2t'], 'pk_tt': pk_k + out['22tt'] + pk13_tt, 'A': A, 'B': B}
    return tab


def tns_table_matrix(tab):
    """The 29 rows [pk11 ... pk_tt, A0..A4, B0..B11] x n_k11 in the order of full_shape.py:882-883."""
    return np.concatenate([np.array([tab[name] for name in TNS_NAMES]), tab['A'], tab['B']], axis=0)


def tns_pktable(k, mu, wmu_ell, q, pk_q, f, qpar=1., qper=1., sigmav=0., fog='lorentzian', kernels=None, k11=None):
    """``TNSPowerSpectrumMultipoles.calculate``, full_shape.py:865-899: dict name -> [n_ell, n_k] (A, B: [3, n_ell, n_k] for the b1^2, b1, 1 terms)."""
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    if fog == 'lorentzian': damping = 1. / (1. + (sigmav * kap * muap)**2 / 2.)**2.
    else: damping = np.exp(-(sigmav * kap * muap)**2)
    if k11 is None: k11 = tns_k11(k)
    tab = tns_table_matrix(tns_pt(k11, q, weights_trapz(q), pk_q, kernels=kernels))
    t = jac * damping * np.moveaxis(interp1d(np.log10(kap), np.log10(k11), tab.T, method='cubic'), [0, 1], [1, 2])   # [29, n_k, n_mu]
    A, B = t[12:17], t[17:]
    m2 = muap**2
    Ac = [f * A[0] * m2, f**2 * (A[1] * m2 + A[2] * m2**2), f**3 * (A[3] * m2**2 + A[4] * m2**3)]
    Bc = [f**2 * (B[0] * m2 + B[4] * m2**2), -f**3 * ((B[1] + B[2]) * m2 + (B[5] + B[6]) * m2**2 + (B[8] + B[9]) * m2**3),
          f**4 * (B[3] * m2 + B[7] * m2**2 + B[10] * m2**3 + B[11] * m2**4)]

    def to_poles(pkmu):
        return np.sum(pkmu[None, ...] * wmu_ell[:, None, :], axis=-1)

    out = {}
    for i, name in enumerate(TNS_NAMES):
        fac = 1. if i < 8 else (f * m2 if i < 11 else f**2 * m2**2)
        out[name] = to_poles(fac * t[i])
    out['A'] = np.array([to_poles(a) for a in Ac])
    out['B'] = np.array([to_poles(b) for b in Bc])
    return out


def tns_tracer_power(pt, nd, b1=1., b2=0., bs=0., b3=0., sn0=0.):
    """``TNSTracerPowerSpectrumMultipoles.calculate``, full_shape.py:957-971 (as coded: sn0 / nd goes to EVERY multipole; the last pk_sig3sq term carries no f mu^2)."""
    power = b1**2 * pt['pk_dd'] + 2. * b1 * pt['pk_dt'] + pt['pk_tt'] + sn0 / nd
    bs2 = bs - 4. / 7. * (b1 - 1.)
    b3nl = b3 + 32. / 315. * (b1 - 1.)
    power = power + 2 * b1 * b2 * pt['pk_b2d'] + 2. * b1 * bs2 * pt['pk_bs2d'] + 2 * b1 * b3nl * pt['pk_sig3sq'] + b2**2 * pt['pk_b22'] \
        + 2 * b2 * bs2 * pt['pk_b2s2'] + bs2**2 * pt['pk_bs22'] + b2 * pt['pk_b2t'] + b3nl * pt['pk_sig3sq']
    power = power + b1**2 * (pt['A'][0] + pt['B'][0]) + b1 * (pt['A'][1] + pt['B'][1]) + (pt['A'][2] + pt['B'][2])
    return power


# ----------------------------------------------------------------------------------------------
# scale-dependent bias from local primordial non-Gaussianity          primordial_non_gaussianity.py:75-116
# ----------------------------------------------------------------------------------------------
def png_alpha_prim(kin, pk_dd, pk_prim, h):
    """``alpha(k)`` of method 'prim': square root of the primordial potential spectrum over the density spectrum; primordial_non_gaussianity.py:84-86."""
    pphi_prim = 9 / 25 * 2 * np.pi**2 / kin**3 * pk_prim / h**3
    return 1. / (pk_dd / pphi_prim)**0.5


def png_bfnl(mode, b1, fnl_loc=0., p=1., bphi=1., bfnl_loc=0.):
    """``bfnl_loc`` of one tracer for the three parameterisations; primordial_non_gaussianity.py:97-106."""
    if mode == 'bphi': return bphi * fnl_loc
    if mode == 'b-p': return 2. * 1.686 * (b1 - p) * fnl_loc
    return bfnl_loc


def png_tracer_power(k, mu, wmu_ell, kin, pk_dd, alpha, f, nd, b1X, b1Y, bfnlX, bfnlY, sn0=0., sigmasX=0., sigmasY=0., qpar=1., qper=1.):
    """``PNGTracerPowerSpectrumMultipoles.calculate``, primordial_non_gaussianity.py:75-112: ``kin, pk_dd, alpha`` WITHOUT the template's first wavenumber (line 95)."""
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    a = interp1d(np.log10(kap), np.log10(kin), alpha, method='cubic')
    bX, bY = b1X + bfnlX * a, b1Y + bfnlY * a
    fog = 1. / ((1. + sigmasX**2 * kap**2 * muap**2 / 2.) * (1. + sigmasY**2 * kap**2 * muap**2 / 2.))
    pkmu = jac * fog * (bX + f * muap**2) * (bY + f * muap**2) * interp1d(np.log10(kap), np.log10(kin), pk_dd, method='cubic') + sn0 / nd
    return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)


def png_velocity_power(k, mu, wmu_ell, kin, pk_dd, alpha, f, z, b1, bfnl, bv=1., sigmas=0., sigmau=0., qpar=1., qper=1.):
    """``PNGTracerVelocityPowerSpectrumMultipoles.calculate``, primordial_non_gaussianity.py:284-320 (the tracer-velocity cross spectrum without its factor i):
    ``mu`` is the reference's grid on [-1, 1] (81 trapezoid nodes), ``wmu_ell`` the weights of the odd multipoles."""
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    bias = b1 + bfnl * interp1d(np.log10(kap), np.log10(kin), alpha, method='cubic')
    vel_bias = bv * f * muap * 100. / (1. + z) / kap
    fog = 1. / (1. + sigmas**2 * kap**2 * muap**2 / 2.) * np.sinc(sigmau * kap)
    pkmu = jac * fog * (bias + f * muap**2) * vel_bias * interp1d(np.log10(kap), np.log10(kin), pk_dd, method='cubic')
    return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)


# ----------------------------------------------------------------------------------------------
# a5: tracer combine                                          full_shape.py:545-550, 628-634
# ----------------------------------------------------------------------------------------------
def kaiser_tracer_power(ells, pk_dd, pk_dt, pk_tt, nd, b1X, b1Y, sn0):
    """full_shape.py:545-550 (cross-correlation aware: b1X b1Y dd + (b1X + b1Y) dt + tt + delta_l0 sn0 / nd)."""
    sn = np.array([(ell == 0) for ell in ells], dtype='f8')[:, None] * sn0 / nd
    return b1X * b1Y * pk_dd + (b1X + b1Y) * pk_dt + pk_tt + sn


def eftlike_addon(power, ells, pk_dd, ct_matrix, ct_values, sn_matrix, sn_values, nd):
    """full_shape.py:628-634: ``+ ct_matrix . (0.5 sum ct) * pk11[ell = 0] + sn_matrix . (sn / nd)``.

    ct_matrix, sn_matrix: [n_ell, n_k, n_par]; ct_values already summed over the two tracers.
    """
    power = power.copy()
    if ct_matrix.size:
        power += ct_matrix.dot(0.5 * np.asarray(ct_values)) * pk_dd[list(ells).index(0)]
    if sn_matrix.size:
        power += sn_matrix.dot(np.asarray(sn_values) / nd)
    return power


# ----------------------------------------------------------------------------------------------
# a6: window matrix                                                 window.py:14-68, 445-473
# ----------------------------------------------------------------------------------------------
def matrix_lininterp(xin, xout):
    """utils.py:646-657."""
    toret = np.zeros((len(xin), len(xout)), dtype='f8')
    for iout, xo in enumerate(xout):
        iin = np.searchsorted(xin, xo, side='right') - 1
        if 0 <= iin < len(xin) - 1:
            frac = (xo - xin[iin]) / (xin[iin + 1] - xin[iin])
            toret[iin, iout] = 1. - frac
            toret[iin + 1, iout] = frac
        elif np.isclose(xo, xin[-1]):
            toret[iin, iout] = 1.
    return toret


def window_matrix_bininteg(list_edges, resolution=1):
    """window.py:14-68: binning matrix in the continuous limit; returns xin, matrix [n_in_total, n_out_total]."""
    resolution = int(resolution)
    step = min((edges[..., 1] - edges[..., 0]).min() for edges in list_edges) / resolution
    start, stop = min(np.min(edges) for edges in list_edges), max(np.max(edges) for edges in list_edges)
    edgesin = np.arange(start, stop + step / 2., step)
    xin = 3. / 4. * (edgesin[1:]**4 - edgesin[:-1]**4) / (edgesin[1:]**3 - edgesin[:-1]**3)
    matrices = []
    for edges in list_edges:
        x, w = [], []
        for ibin, edge in enumerate(edges):
            edge = np.linspace(*edge, resolution + 1)
            x.append(3. / 4. * (edge[1:]**4 - edge[:-1]**4) / (edge[1:]**3 - edge[:-1]**3))
            line = np.zeros(len(edges) * resolution, dtype='f8')
            tmp = edge[1:]**3 - edge[:-1]**3
            line[ibin * resolution:(ibin + 1) * resolution] = tmp / tmp.sum()
            w.append(line)
        matrices.append(matrix_lininterp(xin, np.concatenate(x)).dot(np.column_stack(w)))
    n = len(matrices)
    full = np.block([[matrices[i] if i == j else np.zeros_like(matrices[j]) for j in range(n)] for i in range(n)])
    return xin, full


def window_apply(power, matrix_full=None, offset=None, kmask=None, shotnoisein=None, shotnoiseout=None):
    """window.py:459-473: ``W . ravel(power + sn_in[:, None]) (+ offset) [kmask] - sn_out``."""
    theory = power
    if shotnoisein is not None:
        theory = theory + np.asarray(shotnoisein)[:, None]
    theory = np.ravel(theory)
    if matrix_full is not None:
        theory = np.dot(matrix_full, theory)
    if offset is not None:
        theory = theory + offset
    if kmask is not None:
        theory = theory[kmask]
    if shotnoiseout is not None:
        theory = theory - shotnoiseout
    return theory


# ----------------------------------------------------------------------------------------------
# a7: observable                                                   power_spectrum.py:400-404
# ----------------------------------------------------------------------------------------------
def observable_transform(flattheory, flatdata, transform=None):
    if transform == 'cubic':
        return (3. * (flattheory / flatdata)**(1. / 3.) - 2.) * flatdata
    return flattheory


# ----------------------------------------------------------------------------------------------
# a8: Gaussian chi2                                         likelihoods/base.py:13-17, 658-660
# ----------------------------------------------------------------------------------------------
def chi2(flatdiff, precision):
    if precision.ndim == 1:
        return (flatdiff * precision).dot(flatdiff.T)
    return flatdiff.dot(precision).dot(flatdiff.T)


def gaussian_loglikelihood(flattheory, flatdata, precision):
    """flatdiff = theory - data (likelihoods/base.py:659); logL = -chi2/2."""
    flatdiff = flattheory - flatdata
    return -0.5 * chi2(flatdiff, precision), flatdiff


# ----------------------------------------------------------------------------------------------
# a9: priors                                                parameter.py:1889-1897, 1994-2017
# ----------------------------------------------------------------------------------------------
def prior_logpdf(x, dist='uniform', limits=(-np.inf, np.inf), loc=0., scale=1.):
    """Zero-lag-removed log-pdf, closed limits; parameter.py:1994-2017 (uniform and norm fast paths, scipy.stats for the rest)."""
    isin = (limits[0] <= x) & (x <= limits[1])
    if dist == 'uniform':
        return np.where(isin, 0., -np.inf)
    if dist == 'norm':
        return np.where(isin, -0.5 * (x - loc)**2 / scale**2, -np.inf)
    # any other distribution: scipy's frozen rv, value at loc removed (parameter.py:1958-1966, 2012-2016)
    from scipy import stats
    rv = getattr(stats, dist)(loc=loc, scale=scale)
    with np.errstate(divide='ignore'):
        return np.where(isin, rv.logpdf(x) - rv.logpdf(loc), -np.inf)


def logprior(theta, priors):
    """Sum over varied, non-solved parameters; parameter.py:1889-1897. theta[..., P]; priors: list of dict."""
    theta = np.asarray(theta, dtype='f8')
    toret = np.zeros(theta.shape[:-1])
    for i, prior in enumerate(priors):
        toret = toret + prior_logpdf(theta[..., i], **prior)
    return toret


# ----------------------------------------------------------------------------------------------
# Whole full-shape evaluation for one point (rows a1-a9 chained as in SURVEY.md section 3.1)
# ----------------------------------------------------------------------------------------------
def fullshape_observable(c, p):
    """One observable of a full-shape likelihood at one parameter point.

    ``c``: dict of constants (see ``tests/golden/make_golden.py`` for the layout), ``p``: dict of
    model inputs (qpar, qper, df, dm, dn, b1 (tuple of 2), sn0, sigmapar, sigmaper, + EFT terms).
    Returns dict of every intermediate.
    """
    out = {}
    k11 = c['k11']
    if c['template'] == 'shapefit':
        factor = shapefit_factor(k11, c['kp'], c['a'], dm=p.get('dm', 0.), dn=p.get('dn', 0.))
        pk11 = c['pk_dd_fid'] * factor
    elif c['template'] == 'turnover':
        pk11 = turnover_pk(k11, c['kTO_fid'], c['pkTO_dd_fid'], m=p.get('m', 0.6), n=p.get('n', 0.9), qto=p.get('qto', 1.), dpto=p.get('dpto', 1.))
    elif c['template'] == 'bands':   # power_template.py:955-961: P_tt = P_tt_fid (1 + sum (dptt_i - 1) tent_i), P_dd = P_tt / (f_fid df)^2
        factor = 1. + (np.asarray(p['dptt'], dtype='f8') - 1.).dot(c['band_templates'])
        pk11 = c['pk_tt_fid'] * factor / (c['f_fid'] * p.get('df', 1.))**2
    else:  # fixed / standard / bao: power_template.py:107-108
        pk11 = c['pk_dd_fid']
    f = c['f_fid'] * p.get('df', 1.)
    out['pk_dd_template'], out['f'] = pk11, f
    b1X, b1Y = p['b1']
    if c.get('simple_tracer', False):
        power = simple_tracer_power(c['kin'], c['mu'], c['wmu_ell'], k11, pk11, f, c['nd'], b1X, b1Y, p.get('sn0', 0.), qpar=p.get('qpar', 1.), qper=p.get('qper', 1.),
                                    sigmapar=p.get('sigmapar', 0.), sigmaper=p.get('sigmaper', 0.))
    else:
        dd, dt, tt = kaiser_pktable(c['kin'], c['mu'], c['wmu_ell'], k11, pk11, f, qpar=p.get('qpar', 1.), qper=p.get('qper', 1.),
                                    sigmapar=p.get('sigmapar', 0.), sigmaper=p.get('sigmaper', 0.))
        out['pk_dd'], out['pk_dt'], out['pk_tt'] = dd, dt, tt
        power = kaiser_tracer_power(c['ellsin'], dd, dt, tt, c['nd'], b1X, b1Y, p.get('sn0', 0.))
    if c.get('ct_matrix', None) is not None:
        power = eftlike_addon(power, c['ellsin'], dd, c['ct_matrix'], p['ct'], c['sn_matrix'], p['sn'], c['nd'])
    out['power'] = power
    flat = window_apply(power, matrix_full=c.get('matrix_full', None), offset=c.get('offset', None), kmask=c.get('kmask', None),
                        shotnoisein=c.get('shotnoisein', None), shotnoiseout=c.get('shotnoiseout', None))
    out['flatpower'] = flat
    out['flattheory'] = observable_transform(flat, c['flatdata'], c.get('transform', None))
    return out


# ----------------------------------------------------------------------------------------------
# a10: analytic marginalisation / best-fit of linear nuisance parameters
#      likelihoods/base.py:129-200 (FastFisher.__call__), 314-413 (BaseLikelihood._solve)
# The reference's own `_solve` cannot run here (it needs jax, likelihoods/base.py:130) and its tests only run / plot
# (samplers/tests/test_base.py:380-408): no direct output of it exists to compare with -- "pinned by identity on reference outputs":
#  (i) tests/golden/marg_sn0_grid.npz: the reference's NON-marginalised posterior on a grid of one solved parameter; the Gaussian integral must equal the
#      marginalised value (tests/test_oracle_marg.py, 1e-6);
#  (ii) tests/golden/marg_multi.npz: the exact quadratic form (c, g, H) of the reference's non-marginalised log-posterior in SEVERAL linear parameters (two counter
#      terms with point-dependent derivative rows + a stochastic term; two tracers), obtained from the reference on a stencil; the closed forms
#      x* = x0 - H^-1 g, c - g H^-1 g / 2 - logdet(-H[M, M]) / 2 pin this function and the HIP path to 1e-8 (tests/test_marg_multi.py), '.marg' / '.best' mixes included;
#  (iii) closed forms with flat priors / diagonal precisions (tests/test_oracle_marg.py).
# ----------------------------------------------------------------------------------------------
def solve_marginalized(flatdiff, flatderiv, precision, x0, prior_loc, prior_scale, marg_mask):
    """One parameter point.

    flatdiff: Delta = theory(x0) - data [n]; flatderiv: T = dDelta/dx [n_s, n] (rows = solved parameters, as
    likelihoods/base.py:162); precision [n, n] or [n]; x0: values at which Delta was evaluated (likelihoods/base.py:355);
    prior_loc / prior_scale: Gaussian prior of each solved parameter (scale = inf for flat priors, likelihoods/base.py:180-183);
    marg_mask: True where the parameter is marginalised ('.marg'), False where it is set to its best fit ('.best').
    Returns dict(loglikelihood, logprior_solved, x, dx, posterior_hessian, likelihood_hessian, likelihood_gradient).
    """
    flatdiff, flatderiv = np.asarray(flatdiff, dtype='f8'), np.atleast_2d(np.asarray(flatderiv, dtype='f8'))
    x0, prior_loc, prior_scale = (np.asarray(a, dtype='f8') for a in (x0, prior_loc, prior_scale))
    derivp = flatderiv * precision if precision.ndim == 1 else flatderiv.dot(precision)   # 169-172
    likelihood_gradient = -derivp.dot(flatdiff.T)                                          # 173
    likelihood_hessian = -derivp.dot(flatderiv.T)                                          # 174
    prec = prior_scale**(-2)                                                               # 181
    prior_gradient = -(x0 - prior_loc) * prec                                              # 182
    prior_hessian = np.diag(-prec)                                                         # 183-185
    posterior_gradient = prior_gradient + likelihood_gradient
    posterior_hessian = prior_hessian + likelihood_hessian
    dx = -np.linalg.solve(posterior_hessian, posterior_gradient)                           # 188
    x = x0 + dx                                                                            # 189
    loglikelihood = -0.5 * chi2(flatdiff, precision)
    loglikelihood += 0.5 * dx.dot(likelihood_hessian).dot(dx) + likelihood_gradient.dot(dx)   # 385-386
    # prior of the solved parameters at their solution, zero-lag removed (363-364, parameter.py:2003-2010)
    logprior = np.sum(np.where(np.isinf(prior_scale), 0., -0.5 * (x - prior_loc)**2 * prec))
    marg_mask = np.asarray(marg_mask, dtype='?')
    if marg_mask.any():                                                                    # 394-404: no (2 pi)^(n/2)
        loglikelihood += -0.5 * np.linalg.slogdet(-posterior_hessian[np.ix_(marg_mask, marg_mask)])[1]
    return dict(loglikelihood=loglikelihood, logprior_solved=logprior, x=x, dx=dx, posterior_hessian=posterior_hessian,
                likelihood_hessian=likelihood_hessian, likelihood_gradient=likelihood_gradient)


def marginalize_precision(precision, flatderiv, prior_scale):
    """'.prec' solved parameters: one-off precision marginalisation, likelihoods/base.py:280-309:
    P <- P - P T^T (-H)^-1 T P with H the posterior Hessian (T: [n_s, n])."""
    precision = np.asarray(precision, dtype='f8')
    flatderiv = np.atleast_2d(flatderiv)
    derivp = flatderiv * precision if precision.ndim == 1 else flatderiv.dot(precision)
    posterior_hessian = -derivp.dot(flatderiv.T) - np.diag(np.asarray(prior_scale, dtype='f8')**(-2))
    full = np.diag(precision) if precision.ndim == 1 else precision
    return full - derivp.T.dot(np.linalg.solve(-posterior_hessian, derivp))


# ----------------------------------------------------------------------------------------------
# a11: BAO wiggles P(k, mu) -> P_ell, FFTLog P_ell -> xi_ell, broadband
#      bao.py:117-151 (DampedBAOWigglesPowerSpectrumMultipoles, model 'standard'), bao.py:495-534, 881-905 (broadband),
#      theories/galaxy_clustering/base.py:46-139 (get_corr).
# The Hankel step is THIRD-PARTY in the reference (cosmoprimo.fftlog.PowerToCorrelation, un-vendored, no pinned version, no reference
# test with numbers): PARITY UNPINNED.  The oracle uses scipy.fft.fht (Hamilton 2000), an independent implementation of the published
# algorithm with the padding / grid conventions stated in desilike_amd/fftlog.py; tests/test_oracle_bao.py checks it against the
# brute-force integral of theories/galaxy_clustering/base.py:163-168.  Golden fixtures for this row are produced by the reference's own
# bao.py / base.py code running on top of THIS transform (tests/golden/refstub PowerToCorrelation).
# ----------------------------------------------------------------------------------------------
class FFTLogPowerToCorrelation(object):
    """``PowerToCorrelation(k, ell, q=0, lowring=True)``: fun[n_ell, N] -> (s[n_ell, N], xi[n_ell, N]) through scipy.fft.fht."""

    def __init__(self, k, ell=0, q=0, lowring=True, minfolds=2):
        from scipy import fft
        self.k = np.asarray(k, dtype='f8')
        self.ells = np.atleast_1d(ell)
        n = self.k.size
        self.dln = np.log(self.k[-1] / self.k[0]) / (n - 1)
        self.npad = int(2**np.ceil(np.log2(minfolds * n)))
        self.pad = (self.npad - n) // 2
        self.kpad = self.k[0] * np.exp(self.dln * (np.arange(self.npad) - self.pad))
        self.offsets = [fft.fhtoffset(self.dln, mu=ell + 0.5, initial=0., bias=0.) if lowring else 0. for ell in self.ells]

    def __call__(self, fun):
        from scipy import fft
        fun = np.atleast_2d(fun)
        s, xi = [], []
        sl = slice(self.pad, self.pad + self.k.size)
        for ill, ell in enumerate(self.ells):
            a = np.zeros(self.npad)
            a[sl] = fun[ill] * self.k**1.5
            A = fft.fht(a, self.dln, mu=ell + 0.5, offset=self.offsets[ill], bias=0.)
            sout = np.exp(self.offsets[ill]) / self.kpad[::-1]
            s.append(sout[sl])
            xi.append((-1.)**(ell // 2) / (2. * np.pi)**1.5 * A[sl] * sout[sl]**(-1.5))
        return np.array(s), np.array(xi)


def bruteforce_correlation(kin, power, s, ells):
    """theories/galaxy_clustering/base.py:163-168: direct int dlnk k^3 P_ell j_ell(k s) with trapezoidal weights."""
    lnk = np.log(kin)
    weights = np.concatenate([[lnk[1] - lnk[0]], lnk[2:] - lnk[:-2], [lnk[-1] - lnk[-2]]]) / 2.
    corr = []
    for ill, ell in enumerate(ells):
        tmp = np.sum(kin**3 * power[ill] * weights * special.spherical_jn(ell, np.asarray(s)[:, None] * kin), axis=-1)
        corr.append((-1)**(ell // 2) / (2. * np.pi**2) * tmp)
    return np.array(corr)


def get_corr(power, kin, s, ells, k=None, fftlog=None, interp_order=1):
    """theories/galaxy_clustering/base.py:62-77, 127-136; ``interp_order`` 1 (linear) or 3 (numpy backend: scipy cubic, jax.py:263-265), base.py:54-57."""
    if k is None: k = np.logspace(-4., 3., 2048)
    mask = k > kin[-1]
    logk_high = np.log10(k[mask] / kin[-1])
    damp_high = np.exp(-(k[mask] / kin[-1] - 1.)**2 / (2. * (10.)**2))
    k_mid = k[~mask]
    if fftlog is None: fftlog = FFTLogPowerToCorrelation(k, ell=ells, q=0, lowring=True)
    tmp = []
    for pk in power:
        slope_high = (pk[-1] - pk[-2]) / np.log10(kin[-1] / kin[-2])
        interp = interp1d(np.log10(k_mid), np.log10(kin), pk, method={1: 1, 3: 'cubic'}[interp_order])
        tmp.append(np.concatenate([interp, (pk[-1] + slope_high * logk_high) * damp_high], axis=-1))
    ss, corr = fftlog(np.vstack(tmp))
    return np.array([np.interp(s, sss, cc) for sss, cc in zip(ss, corr)])


def bao_damped_power(k, mu, wmu_ell, k_t, pk_dd, pknow_dd, f, qpar=1., qper=1., b1=1., sigmas=0., sigmapar=9., sigmaper=6., mode='', smoothing_radius=15., model='standard'):
    """bao.py:117-151: model 'standard' (Chen 2023) or the 'fix-damping' / 'move-all' / 'fog-damping' family (137-150); ``f`` already includes dbeta (bao.py:119)."""
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    logkt = np.log10(k_t)
    pknowap = interp1d(np.log10(kap), logkt, pknow_dd, method='cubic')
    pkap = interp1d(np.log10(kap), logkt, pk_dd, method='cubic')
    kk = k[:, None]
    if model != 'standard':
        kd, mud = (kk, mu) if 'fix-damping' in model else (kap, muap)                                   # 137-138
        sigma_nl2 = kd**2 * (sigmapar**2 * mud**2 + sigmaper**2 * (1. - mud**2))
        damped_wiggles = (pkap - pknowap) / pknowap * np.exp(-sigma_nl2 / 2.)                           # 140
        ks, mus = (kap, muap) if 'move-all' in model else (kk, mu)                                      # 141-142
        pknow = interp1d(np.log10(ks * np.ones_like(kap)), logkt, pknow_dd, method='cubic')
        fog = 1. / (1. + (sigmas * ks * mus)**2 / 2.)**2.
        sk = np.exp(-1. / 2. * (ks * smoothing_radius)**2) if mode == 'reciso' else 0.
        pksmooth = (b1 + f * mus**2 * (1 - sk))**2 * pknow
        pkmu = pksmooth * fog * (1. + damped_wiggles) if 'fog-damping' in model else pksmooth * (fog + damped_wiggles)   # 147-150
        return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)
    pkwap = pkap - pknowap
    sigma_nl2ap = kap**2 * (sigmapar**2 * muap**2 + sigmaper**2 * (1. - muap**2))
    sk = 0.
    if mode == 'reciso': sk = np.exp(-1. / 2. * (kk * smoothing_radius)**2)
    Cap = (b1 + f * muap**2 * (1 - sk))**2 * np.exp(-sigma_nl2ap / 2.)
    fog = 1. / (1. + (sigmas * kk * mu)**2 / 2.)**2.
    B = (b1 + f * mu**2 * (1 - sk))**2 * fog
    pknow = interp1d(np.log10(kk), logkt, pknow_dd, method='cubic')
    pkmu = B * pknow + Cap * pkwap
    return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)


def bao_resummation_scales(k_t, pknow_dd, rs_drag, mode='', smoothing_radius=15.):
    """ResummedPowerSpectrumWiggles.calculate, bao.py:186-199: sigma_dd^2, sigma_nl^2, sigma_x^2, sigma_sn^2 (Simpson integrals over the template knots)."""
    from scipy import special, integrate
    j0 = special.jn(0, rs_drag * k_t)
    sk = np.exp(-1. / 2. * (k_t * smoothing_radius)**2) if mode else 0.
    skc = 1. - sk
    sigma_sn2 = 1. / smoothing_radius / 6 / np.pi**(3. / 2.)
    sigma_nl2 = 1. / (3. * np.pi**2) * integrate.simpson((1. - j0) * pknow_dd, x=k_t)
    sigma_dd2 = 1. / (3. * np.pi**2) * integrate.simpson((1. - j0) * skc**2 * pknow_dd, x=k_t)
    sigma_x2 = 1. / (3. * np.pi**2) * integrate.simpson((1. - j0) * skc * pknow_dd, x=k_t) if mode == 'reciso' else 0.
    return sigma_dd2, sigma_nl2, sigma_x2, sigma_sn2


def bao_resummed_power(k, mu, wmu_ell, k_t, pk_dd, pknow_dd, f, scales, shotnoise=0., qpar=1., qper=1., b1=1., sigmas=0., d=1., mode='', smoothing_radius=15., model='standard'):
    """ResummedBAOWigglesPowerSpectrumMultipoles.calculate (bao.py:246-266) with ResummedPowerSpectrumWiggles.wiggles (201-222); ``f`` includes dbeta."""
    sigma_dd2, sigma_nl2, sigma_x2, sigma_sn2 = scales
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    logkt = np.log10(k_t)
    pknowap = interp1d(np.log10(kap), logkt, pknow_dd, method='cubic')
    wig = interp1d(np.log10(kap), logkt, pk_dd, method='cubic') - pknowap
    ksq = (1 + f * (f + 2) * muap**2) * kap**2                                                      # 204
    sdd2 = sigma_dd2 + shotnoise * sigma_sn2 / b1**2
    sk = np.exp(-1. / 2. * (kap * smoothing_radius)**2)
    skc = 1. - sk
    if mode == 'reciso':                                                                            # 211-216
        res = (b1 + f * muap**2 * skc - sk)**2 * np.exp(-1. / 2. * ksq * d**2 * sdd2)
        sigma_ds2 = (1. + f * muap**2) * sdd2 + f * (1. + f) * muap**2 * sigma_x2
        res = res + 2. * (b1 + f * muap**2 * skc - sk) * (1 + f * muap**2) * sk * np.exp(-1. / 2. * ksq * d**2 * sigma_ds2)
        sigma_ss2 = sdd2 + f**2 * muap**2 * sigma_nl2 + 2 * f * muap**2 * sigma_x2
        res = res + (1 + f * muap**2)**2 * sk**2 * np.exp(-1. / 2. * ksq * d**2 * sigma_ss2)
    else:                                                                                           # 209-210, 217-218
        res = (b1 + f * muap**2)**2 * np.exp(-1. / 2. * ksq * d**2 * sdd2)
    damped_wiggles = res * wig / pknowap                                                            # 251
    kk = k[:, None]
    ks, mus = (kap, muap) if 'move-all' in model else (kk, mu)
    pknow = interp1d(np.log10(ks * np.ones_like(kap)), logkt, pknow_dd, method='cubic')
    fog = 1. / (1. + (sigmas * ks * mus)**2 / 2.)**2.
    sks = np.exp(-1. / 2. * (ks * smoothing_radius)**2) if mode == 'reciso' else 0.
    pksmooth = (b1 + f * mus**2 * (1 - sks))**2 * pknow
    pkmu = pksmooth * fog * (1. + damped_wiggles) if 'fog-damping' in model else pksmooth * (fog + damped_wiggles)
    return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)


def bao_flexible_power(k, mu, wmu_ell, ells, k_t, pk_dd, pknow_dd, f, ml_matrix, ml_values, qpar=1., qper=1., b1=1., mode='', smoothing_radius=15., model='standard'):
    """FlexibleBAOWigglesPowerSpectrumMultipoles.calculate / get_wiggles (bao.py:360-383): ``ml_matrix`` [n_ell, n_k, n_ml] kernels, ``ml_values`` [n_ml]; ``f`` includes dbeta."""
    from scipy import special
    jac, kap, muap = ap_k_mu(k, mu, qpar=qpar, qper=qper)
    logkt = np.log10(k_t)
    pknowap = interp1d(np.log10(kap), logkt, pknow_dd, method='cubic')
    wiggles = interp1d(np.log10(kap), logkt, pk_dd, method='cubic') - pknowap
    damped_wiggles = 0.
    for ill, ell in enumerate(ells):
        mult = ml_matrix[ill].dot(ml_values)
        if ell == 0: mult = mult + 1.
        damped_wiggles = damped_wiggles + wiggles * mult[:, None] * special.legendre(ell)(mu)
    damped_wiggles = damped_wiggles / pknowap
    kk = k[:, None]
    ks, mus = (kap, muap) if 'move-all' in model else (kk, mu)
    pknow = interp1d(np.log10(ks * np.ones_like(kap)), logkt, pknow_dd, method='cubic')
    sk = np.exp(-1. / 2. * (ks * smoothing_radius)**2) if mode == 'reciso' else 0.
    pkmu = (b1 + f * mus**2 * (1 - sk))**2 * pknow * (1. + damped_wiggles)
    return np.sum(pkmu * wmu_ell[:, None, :], axis=-1)


# ----------------------------------------------------------------------------------------------
# a5 (velocileptors part): table-level bias combination                      full_shape.py:1182-1186, 1300-1313, 1577-1599, 1479-1488
# ----------------------------------------------------------------------------------------------
def velocileptors_pars(params, sigma8, f, basis='physical', model='rept', snd=1., fsat=1., sigv=1.):
    """The 11 'pars' (b1, b2, bs, b3, alpha0, alpha2, alpha4, alpha6, sn0, sn2, sn4) fed to the table combination.

    ``basis='physical'``: full_shape.py:1300-1307 (LPT) / 1577-1592 (REPT: Eulerian b1 = 1 + b1L, b2 = 8/21 b1L + b2L);
    ``model='rept'`` additionally applies the co-evolution shift of combine_bias_terms_poles (1479-1488)."""
    if basis == 'physical':
        b1L, b2L, bsL, b3L = params['b1p'] / sigma8 - 1., params['b2p'] / sigma8**2, params['bsp'] / sigma8**2, params['b3p'] / sigma8**3
        pars = [1. + b1L, 8. / 21. * b1L + b2L, bsL, b3L] if model == 'rept' else [b1L, b2L, bsL, b3L]
        pars += [(1 + b1L)**2 * params['alpha0p'], f * (1 + b1L) * (params['alpha0p'] + params['alpha2p']),
                 f * (f * params['alpha2p'] + (1 + b1L) * params['alpha4p']), f**2 * params['alpha4p']]
        pars += [params['sn{:d}p'.format(i)] * snd * (fsat if i > 0 else 1.) * sigv**i for i in [0, 2, 4]]
    else:
        pars = [params[name] for name in ['b1', 'b2', 'bs', 'b3', 'alpha0', 'alpha2', 'alpha4', 'alpha6', 'sn0', 'sn2', 'sn4']]
    if model == 'rept':   # full_shape.py:1481-1485
        pars = list(pars)
        b1 = pars[0]
        pars[2] = pars[2] - (2 / 7) * (b1 - 1.)
        pars[3] = 3 * pars[3] + (b1 - 1.)
    return pars


def tablevel_combine_bias_terms_poles(pktable, pars, nd=1e-4):
    """full_shape.py:1182-1186: sum over the 19 bias monomials, in this order."""
    b1, b2, bs, b3, alpha0, alpha2, alpha4, alpha6, sn0, sn2, sn4 = pars
    bias_monomials = np.array([1, b1, b1**2, b2, b1 * b2, b2**2, bs, b1 * bs, b2 * bs, bs**2, b3, b1 * b3, alpha0, alpha2, alpha4, alpha6, sn0 / nd, sn2 / nd, sn4 / nd])
    return np.sum(pktable * bias_monomials, axis=-1)


# ----------------------------------------------------------------------------------------------
# a12: emulator forward pass.  THIRD-PARTY in the reference (cosmoprimo.emulators.tools, un-vendored, no pinned version): PARITY UNPINNED.
# Restated from the data layouts the reference writes: Taylor -- emulators/__init__.py:471-507 (center, powers, derivatives already divided by
# the factorials); MLP -- emulators/conversion.py:20-35, 63-96 (min-max x-scaler, dense layers '(v @ kernel) + bias', silu / relu / tanh,
# inverse min-max y-scaler).  Pins: Taylor reproduces a polynomial calculator exactly (emulators/tests/test_taylor.py:99-104 checks the centre).
# ----------------------------------------------------------------------------------------------
def taylor_predict(x, center, powers, derivatives):
    """y = sum_t derivatives[t] prod_p (x_p - c_p)^powers[t, p]; x [..., P] -> y [..., *yshape]."""
    dx = np.asarray(x, dtype='f8') - center
    monomials = np.prod(dx[..., None, :]**powers, axis=-1)          # [..., n_terms]
    return np.tensordot(monomials, derivatives, axes=([-1], [0]))


def mlp_predict(x, xlimits, layers, activation, ylimits=None):
    """x [..., P]; xlimits [P, 2]; layers: list of (kernel [in, out], bias [out]); ylimits [..., 2] broadcastable to the output."""
    v = (np.asarray(x, dtype='f8') - xlimits[..., 0]) / (xlimits[..., 1] - xlimits[..., 0])       # conversion.py:75-77
    for ilayer, (kernel, bias) in enumerate(layers):
        v = v.dot(kernel) + bias                                                                   # conversion.py:25
        if ilayer < len(layers) - 1:                                                               # conversion.py:27-34
            if activation == 'silu': v = v / (1. + np.exp(-v))
            elif activation == 'relu': v = np.maximum(v, 0.)
            elif activation == 'tanh': v = np.tanh(v)
            else: raise ValueError(activation)
    if ylimits is not None:
        v = v * (ylimits[..., 1] - ylimits[..., 0]) + ylimits[..., 0]                              # conversion.py:79 (inverse)
    return v


def stacked_mlp_predict(X, params, xlimits, layers, activation, ylimits, amplitude_power=0):
    """One engine of the jaxeffort layout (emulators/conversion.py:52-98): every ``kernel [n_z, n_ell, in, out]`` / ``bias [n_z, n_ell, out]`` holds one network per
    (z, ell) (``merge_operations``, 58-66); the layer expression ``(v[..., None, :] @ kernel)[..., 0, :] + bias`` (25) runs them all by broadcasting.
    ``X``: dict of the inputs; ``xlimits [P, 2]``: the one min-max scaler of the engine (75); ``ylimits [n_z, n_ell, n_m, n_k, 2]``: inverse min-max scaler (79);
    ``amplitude_power`` 1 ('11', 'ct') / 2 ('loop') / 0 ('st'): the inverse of conversion.py:88-92, ``v * (exp(logA) * 1e-10)**power``, applied last.
    Returns ``[n_z, n_ell, n_m, n_k]``."""
    v = (np.array([X[name] for name in params], dtype='f8') - xlimits[..., 0]) / (xlimits[..., 1] - xlimits[..., 0])
    for ilayer, (kernel, bias) in enumerate(layers):
        v = np.matmul(v[..., None, :], kernel)[..., 0, :] + bias
        if ilayer < len(layers) - 1:
            if activation == 'silu': v = v / (1. + np.exp(-v))
            elif activation == 'relu': v = np.maximum(v, 0.)
            elif activation == 'tanh': v = np.tanh(v)
            else: raise ValueError(activation)
    v = v.reshape(ylimits.shape[:-1])
    v = v * (ylimits[..., 1] - ylimits[..., 0]) + ylimits[..., 0]
    if amplitude_power == 1: v = v * np.exp(X['logA']) * 1e-10
    elif amplitude_power == 2: v = v * (np.exp(X['logA']) * 1e-10)**2
    return v


def jaxeffort_pktable(components, zgrid=None, z=None):
    """``pktable`` of the emulated REPT node from the outputs ``[n_z, n_ell, n_m, n_k]`` of the engines '11', 'loop', 'ct', 'st' (in this order):
    concatenation along the monomial axis and ``moveaxis(..., [0, -1], [-1, 1])`` -> ``[n_ell, n_k, 19, n_z]`` (conversion.py:50-51), then -- ``z`` given and not the emulated grid
    itself -- the selection / blend ``_emulator_initialize`` inserts (full_shape.py:1416-1443): ``iz = searchsorted(zgrid, z, 'right') - 1``, the two bracketing redshifts kept,
    ``pktable[..., iz] (1 - wz) + pktable[..., iz + 1] wz`` with ``wz = z - zgrid[iz]`` exactly as the reference writes it (a difference of redshifts, not a fraction of the
    interval).  Returns ``[n_ell, n_k, 19, len(z)]`` (``[n_ell, n_k, 19, n_z]`` without ``z``)."""
    pktable = np.moveaxis(np.concatenate(list(components), axis=-2), [0, -1], [-1, 1])
    if z is None: return pktable
    z, zgrid = np.atleast_1d(np.asarray(z, dtype='f8')), np.asarray(zgrid, dtype='f8')
    if z.shape == zgrid.shape and np.allclose(z, zgrid): return pktable
    if np.any((z < zgrid[0]) | (z > zgrid[-1])): raise ValueError('z outside of the emulated range')
    iz = np.searchsorted(zgrid, z, side='right') - 1
    izp1 = np.minimum(iz + 1, len(zgrid) - 1)
    keepiz = np.unique(np.concatenate([iz, izp1], axis=0))
    kept = zgrid[keepiz]
    pktable = pktable[..., keepiz]           # (the reference slices the stacked weights instead: full_shape.py:1439-1442 -- the same numbers)
    iz = np.searchsorted(keepiz, iz, side='right') - 1
    wz = z - kept[iz]
    return pktable[..., iz] * (1 - wz) + pktable[..., iz + 1] * wz


# ----------------------------------------------------------------------------------------------
# a13: Fisher algebra                         fisher.py:731-750 (Gaussian finalize), 216-257 (LikelihoodFisher), 50-53 (FisherGaussianLikelihood)
# The reference's driver (Differentiation + mpi scatter) needs mpi4py: not runnable here.  The algebra below is a line-by-line restatement;
# it is pinned by finite differences of the reference-pinned likelihood (tests/test_oracle.py fixtures) in tests/test_fisher.py.
# ----------------------------------------------------------------------------------------------
def fisher_gaussian(flatdiff, flatderiv, precision):
    """flatdiff [n], flatderiv [P, n] = d(flatdiff)/d(theta) -> offset, gradient, hessian (fisher.py:739-748; NOTE offset = -d P d, no 1/2: line 746)."""
    if precision.ndim == 1:
        diffp, derivp = flatdiff * precision, flatderiv * precision
    else:
        diffp, derivp = flatdiff.dot(precision), flatderiv.dot(precision)
    return -diffp.dot(flatdiff.T), -derivp.dot(flatdiff.T), -derivp.dot(flatderiv.T)


def fisher_mean_chi2min(center, offset, gradient, hessian):
    """LikelihoodFisher.mean / chi2min, fisher.py:216-232."""
    solve = np.linalg.solve(hessian, gradient)
    flatdiff = -solve
    return center - solve, -2. * (offset + gradient.dot(flatdiff) + 0.5 * flatdiff.dot(hessian).dot(flatdiff))


# ----------------------------------------------------------------------------------------------
# f2: MLP emulator training.  THIRD-PARTY in the reference (cosmoprimo.emulators.tools.MLPEmulatorEngine; desilike/emulators/__init__.py:510-533 only wraps it):
# PARITY UNPINNED.  Restated as the published algorithm: mean squared error of the scaled outputs, back-propagation through the layers of emulators/conversion.py:20-35,
# Adam (Kingma & Ba 2015) with bias correction.  Pins: gradient against finite differences (tests/test_oracle_emulator.py), the HIP trainer against this restatement
# step by step (tests/test_gpu_mlp_train.py).
# ----------------------------------------------------------------------------------------------
def _mlp_act(z, activation):
    if activation == 'silu': return z / (1. + np.exp(-z))
    if activation == 'relu': return np.maximum(z, 0.)
    if activation == 'tanh': return np.tanh(z)
    raise ValueError(activation)


def _mlp_act_prime(z, activation):
    if activation == 'silu':
        s = 1. / (1. + np.exp(-z))
        return s * (1. + z * (1. - s))
    if activation == 'relu': return (z > 0.).astype('f8')
    if activation == 'tanh': return 1. - np.tanh(z)**2
    raise ValueError(activation)


def mlp_loss_and_grad(layers, x, y, activation='silu'):
    """Mean squared error of the network output (last layer linear) and its gradient: list of (dkernel, dbias)."""
    zs, acts, a = [], [x], x
    for il, (kernel, bias) in enumerate(layers):
        z = a.dot(kernel) + bias
        zs.append(z)
        a = _mlp_act(z, activation) if il < len(layers) - 1 else z
        acts.append(a)
    diff = a - y
    loss = np.mean(diff**2)
    delta = 2. * diff / diff.size
    grads = [None] * len(layers)
    for il in range(len(layers) - 1, -1, -1):
        grads[il] = (acts[il].T.dot(delta), delta.sum(axis=0))
        if il > 0: delta = delta.dot(layers[il][0].T) * _mlp_act_prime(zs[il - 1], activation)
    return loss, grads


def mlp_adam(layers, x, y, batch, nsteps, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, activation='silu'):
    """``nsteps`` Adam steps on consecutive chunks of ``batch`` samples (step t uses chunk t mod (S // batch)); returns (layers, losses)."""
    layers = [(kernel.copy(), bias.copy()) for kernel, bias in layers]
    moments = [[np.zeros_like(kernel), np.zeros_like(bias), np.zeros_like(kernel), np.zeros_like(bias)] for kernel, bias in layers]
    chunks, losses = len(x) // batch, []
    for t in range(1, nsteps + 1):
        c = (t - 1) % chunks
        loss, grads = mlp_loss_and_grad(layers, x[c * batch:(c + 1) * batch], y[c * batch:(c + 1) * batch], activation)
        losses.append(loss)
        c1, c2 = 1. - beta1**t, 1. - beta2**t
        for il, (gk, gb) in enumerate(grads):
            for ip, g in enumerate([gk, gb]):
                m = moments[il][ip] = beta1 * moments[il][ip] + (1. - beta1) * g
                v = moments[il][2 + ip] = beta2 * moments[il][2 + ip] + (1. - beta2) * g * g
                layers[il][ip][...] -= lr * (m / c1) / (np.sqrt(v / c2) + eps)
    return layers, np.array(losses)


# ----------------------------------------------------------------------------------------------
# f4: Gaussian covariance of power-spectrum / correlation-function multipoles
#     observables/galaxy_clustering/covariance.py:14-41 (integral_legendre_product), 274-456 (ObservablesCovarianceMatrix.run / _run)
# Pinned on the reference's own output: tests/golden/covariance.npz (tests/golden/make_covariance_fixture.py), tests/test_covariance.py.
# ----------------------------------------------------------------------------------------------
def integral_legendre_product(ells, mu_range=(-1., 1.)):
    """covariance.py:14-41: integral over mu of the product of the Legendre polynomials of orders ``ells``."""
    poly = special.legendre(0)
    for ell in ells:
        poly = poly * special.legendre(ell)
    integ = poly.integ()
    return integ(mu_range[-1]) - integ(mu_range[0])


def gaussian_covariance_block(obs1, obs2, theory1, theory2, resolution=1):
    """covariance.py:355-456 (``_run``) for one pair of observables of the same tracer.  ``obs``: dict(kind='pk' | 'xi', ells, edges=[per multipole: [n, 2]],
    volume, shotnoise); ``theory``: dict(k [n_k], ells, power [n_ell, n_k]).  Returns the block [n1, n2]."""
    if obs1['kind'] == 'pk' and obs2['kind'] == 'xi':                         # covariance.py:420-421
        return gaussian_covariance_block(obs2, obs1, theory2, theory1, resolution=resolution).T
    volume = min(obs1['volume'], obs2['volume'])                              # BaseFootprint.__and__ (covariance.py:99-101)
    shotnoise = [obs1['shotnoise'], obs2['shotnoise']]
    theories = [theory1, theory2]

    def pk(it, k, ell):                                                       # covariance.py:361-371
        theory = theories[it]
        ill = list(theory['ells']).index(ell)
        return np.interp(k, theory['k'], theory['power'][ill] + (ell == 0) * shotnoise[it])

    def sigma_k(ell1, ell2, k, remove_zero_lag=False):                        # covariance.py:373-382
        toret = 0.
        for la in theories[0]['ells']:
            for lb in theories[1]['ells']:
                zero_lag = remove_zero_lag * (la == 0) * (lb == 0) * shotnoise[0] * shotnoise[1]
                toret = toret + (pk(0, k, la) * pk(1, k, lb) - zero_lag) * integral_legendre_product((la, lb, ell1, ell2))
        return (2 * ell1 + 1) * (2 * ell2 + 1) / volume * toret

    def bin_volume(edges):                                                    # covariance.py:384-388
        return 4. / 3. * np.pi * (edges[1]**3 - edges[0]**3)

    def integ_points(edges):                                                  # covariance.py:392-393
        return np.linspace(edges[0], edges[1], resolution + 2)[1:-1]

    def weights_trapz(x):                                                     # utils.py:614-622
        return np.concatenate([[x[1] - x[0]], x[2:] - x[:-2], [x[-1] - x[-2]]]) / 2.

    cache = {}
    kinds = (obs1['kind'], obs2['kind'])

    def bin_cov(ells, ibins):
        bins = [np.asarray(o['edges'][list(o['ells']).index(ell)][ibin]) for o, ell, ibin in zip((obs1, obs2), ells, ibins)]
        if kinds == ('pk', 'pk'):                                             # covariance.py:397-406
            inter = (max(bins[0][0], bins[1][0]), min(bins[0][1], bins[1][1]))
            if inter[0] >= inter[1]: return 0.
            k = integ_points(inter)
            return (2. * np.pi)**3 * bin_volume(inter) / (bin_volume(bins[0]) * bin_volume(bins[1])) * np.sum(k**2 * sigma_k(ells[0], ells[1], k)) / np.sum(k**2)
        if kinds == ('xi', 'pk'):                                             # covariance.py:408-416
            s, k = integ_points(bins[0]), integ_points(bins[1])
            weights = np.sum(s[:, None]**2 * special.spherical_jn(ells[0], s[:, None] * k), axis=0) / np.sum(s**2)
            return np.sign(1j**ells[0]).real * np.sum(k**2 * sigma_k(ells[0], ells[1], k) * weights) / np.sum(k**2)
        # xi x xi: covariance.py:423-446
        if 'k' not in cache:
            ks = [theory['k'] for theory in theories]
            k = np.unique(np.concatenate(ks))
            cache['k'] = k[(k >= max(kk.min() for kk in ks)) & (k <= min(kk.max() for kk in ks))]
        k = cache['k']
        if ells not in cache:
            cache[ells] = sigma_k(ells[0], ells[1], k, remove_zero_lag=True) * (4. * np.pi * k**2 * weights_trapz(k))
        ss = [integ_points(b) for b in bins]
        weights = np.prod([np.sum(s[:, None]**2 * special.spherical_jn(ell, s[:, None] * k), axis=0) / np.sum(s**2) for s, ell in zip(ss, ells)], axis=0)
        toret = np.sign(1j**sum(ells)).real / (2. * np.pi)**3 * np.sum(cache[ells] * weights)
        inter = (max(bins[0][0], bins[1][0]), min(bins[0][1], bins[1][1]))
        if inter[0] >= inter[1]: return toret
        sn = integral_legendre_product((0, 0) + tuple(ells)) * shotnoise[0] * shotnoise[1] * (2 * ells[0] + 1) * (2 * ells[1] + 1) / volume
        return toret + np.sign(1j**sum(ells)).real * bin_volume(inter) / (bin_volume(bins[0]) * bin_volume(bins[1])) * sn

    rows = []
    for ill1, ell1 in enumerate(obs1['ells']):
        row = []
        for ill2, ell2 in enumerate(obs2['ells']):
            n1, n2 = len(obs1['edges'][ill1]), len(obs2['edges'][ill2])
            row.append(np.array([[bin_cov((ell1, ell2), (i1, i2)) for i2 in range(n2)] for i1 in range(n1)], dtype='f8'))
        rows.append(row)
    return np.block(rows)


def gaussian_covariance(observables, theories, resolution=1):
    """covariance.py:343-353: blocks for every pair of observables, the diagonal blocks symmetrised."""
    nobs = len(observables)
    blocks = [[None] * nobs for _ in range(nobs)]
    for io1 in range(nobs):
        for io2 in range(io1 + 1):
            c = gaussian_covariance_block(observables[io1], observables[io2], theories[io1], theories[io2], resolution=resolution)
            if io1 == io2: blocks[io1][io2] = (c + c.T) / 2.
            else: blocks[io1][io2], blocks[io2][io1] = c, c.T
    return np.block(blocks)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Blocked Metropolis-Hastings sampler (desilike/samplers/mcmc.py).  Two sources of random draws:
#   MHNumpyDraws   the reference's own sequence of numpy RandomState calls (mcmc.py:107-112, 150-183): with the same seed the restatement reproduces the
#                  reference's chain bit for bit (tests/golden/mh_*.npz, generated by the reference's classes);
#   MHPhiloxDraws  the counter-based draws of the device sampler (desilike_amd/csrc/dl_mh.h): pure functions of (seed, chain, proposer call).
# ---------------------------------------------------------------------------------------------------------------------------------------------
def mh_format_blocks(blocks, oversample_factors=None):
    """mcmc.py:247-256: starts of the blocks, the block of every parameter, the cycler's repeated parameter indices."""
    blocks = np.array(blocks, dtype='i4')
    oversample_factors = np.ones(len(blocks), dtype='i4') if oversample_factors is None else np.array(oversample_factors, dtype='i4')
    block_starts = np.insert(np.cumsum(blocks), 0, 0)
    indices_repeated = np.concatenate([np.repeat(np.arange(b) + s, o) for b, s, o in zip(blocks, block_starts, oversample_factors)])
    param_block_indices = np.concatenate([np.full(b, ib, dtype='i4') for ib, b in enumerate(blocks)])
    return blocks, oversample_factors, block_starts, indices_repeated, param_block_indices


def mh_transforms(covariance, blocks):
    """mcmc.py:298-328: per block the columns of the Cholesky factor, from the block's first row down."""
    covariance = np.array(covariance, dtype='f8')
    L = np.linalg.cholesky(covariance)
    block_starts = np.insert(np.cumsum(blocks), 0, 0)
    return [L[start:, start:start + b] for start, b in zip(block_starts, blocks)]


class MHNumpyDraws(object):
    """The reference's draws: one numpy RandomState consumed in the reference's order."""

    def __init__(self, rng, blocks, oversample_factors=None):
        from scipy.stats import special_ortho_group
        self._so = special_ortho_group
        self.rng = rng
        self.blocks, self.oversample_factors, self.block_starts, self.indices_repeated, self.param_block_indices = mh_format_blocks(blocks, oversample_factors)
        self.cycler_loop, self.cycler_indices = -1, (list(range(len(self.indices_repeated))) if len(self.indices_repeated) <= 2 else None)
        if self.cycler_indices is not None: self.cycler_indices = list(self.indices_repeated)   # mcmc.py:146-147
        self.block_loop, self.rotmat = [-1] * len(self.blocks), [None] * len(self.blocks)

    def _radius(self, b):   # mcmc.py:176-183
        if self.rng.uniform() < 0.33:
            return self.rng.standard_exponential()
        return np.sqrt(self.rng.chisquare(min(b, 2)))

    def direction(self, call):
        """(block, direction x radius) of the next proposer call (mcmc.py:150-155, 163-173, 270-277); ``call`` is not used: the draws are sequential."""
        nrep = len(self.indices_repeated)
        self.cycler_loop = (self.cycler_loop + 1) % nrep
        if self.cycler_loop == 0 and nrep > 2:
            self.cycler_indices = self.rng.permutation(self.indices_repeated)
        ib = int(self.param_block_indices[self.cycler_indices[self.cycler_loop]])
        b = int(self.blocks[ib])
        if b == 1:
            return ib, np.array([self.rng.choice([-1, 1]) * self._radius(b)])
        self.block_loop[ib] = (self.block_loop[ib] + 1) % b
        if self.block_loop[ib] == 0:
            self.rotmat[ib] = self._so.rvs(b, random_state=self.rng)
        return ib, self.rotmat[ib][:, self.block_loop[ib]] * self._radius(b)

    def exponential(self, call):
        return self.rng.standard_exponential()


def _mh_philox(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011) on python integers."""
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _mh_uniform53(hi, lo):
    return ((hi >> 5) * 67108864. + (lo >> 6)) / 9007199254740992.


class MHPhiloxDraws(object):
    """The device sampler's draws (csrc/dl_mh.h), as functions of the proposer call ``n`` of chain ``chain``."""
    PERM_A, PERM_B, RADIAL, RADIAL2, ACCEPT, ROT = 16, 17, 18, 19, 20, 21

    def __init__(self, seed, chain, blocks, oversample_factors=None):
        self.k0, self.k1, self.chain = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF, int(chain)
        self.blocks, self.oversample_factors, self.block_starts, self.indices_repeated, self.param_block_indices = mh_format_blocks(blocks, oversample_factors)
        self.rep_block = self.param_block_indices[self.indices_repeated]

    def _words(self, counter, stream):
        return _mh_philox(counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, self.chain, stream, self.k0, self.k1)

    def permutation(self, cycle):
        n = len(self.rep_block)
        if n <= 2: return list(range(n))
        ka, kb = self._words(cycle, self.PERM_A), self._words(cycle, self.PERM_B)
        bits = max(int(n - 1).bit_length(), 1)
        mask, shift = (1 << bits) - 1, (bits + 1) // 2
        out = []
        for p in range(n):
            y = p
            while True:
                for r in range(4):
                    y = (y * (ka[r] | 1) + kb[r]) & mask
                    y ^= y >> shift
                if y < n: break
            out.append(y)
        return out

    def gauss(self, m, ib, refl, element):
        w = self._words(m, self.ROT | (ib << 8) | (refl << 14) | ((element >> 1) << 20))
        rho, phi = np.sqrt(-2. * np.log1p(-_mh_uniform53(w[0], w[1]))), 2. * np.pi * _mh_uniform53(w[2], w[3])
        return rho * np.sin(phi) if element & 1 else rho * np.cos(phi)

    def rotation_column(self, m, ib, b, j):
        """Column j of the Haar rotation: Householder reflections of Gaussian vectors (Stewart 1980; scipy.stats.special_ortho_group) applied to e_j."""
        y = np.zeros(b); y[j] = 1.
        D = np.ones(b)
        for k in range(b - 2, -1, -1):
            x = np.array([self.gauss(m, ib, k, e) for e in range(b - k)])
            norm2 = np.sum(x**2)
            x0 = x[0]
            D[k] = -1. if x0 < 0. else 1.
            x[0] = x0 + D[k] * np.sqrt(norm2)
            xx = (norm2 - x0**2) + x[0]**2
            y[k:] -= 2. * x * (np.dot(x, y[k:]) / xx)
        D[b - 1] = (-1.)**(b - 1) * np.prod(D[:b - 1])
        return D * y

    def direction(self, call):
        nrep = len(self.rep_block)
        q, p = divmod(int(call), nrep)
        perm = self.permutation(q)
        ib = int(self.rep_block[perm[p]])
        b = int(self.blocks[ib])
        calls = q * int(self.blocks[ib] * self.oversample_factors[ib]) + sum(int(self.rep_block[perm[pp]]) == ib for pp in range(p))
        w = self._words(call, self.RADIAL)
        mix, e = _mh_uniform53(w[0], w[1]), -np.log1p(-_mh_uniform53(w[2], w[3]))
        if b >= 2:
            radius = e if mix < 0.33 else np.sqrt(2. * e)
            return ib, self.rotation_column(calls // b, ib, b, calls % b) * radius
        w2 = self._words(call, self.RADIAL2)
        g = np.sqrt(2. * e) * np.cos(2. * np.pi * _mh_uniform53(w2[0], w2[1]))
        radius = e if mix < 0.33 else abs(g)
        return ib, np.array([(1. if w2[2] & 1 else -1.) * radius])

    def exponential(self, call):
        w = self._words(call, self.ACCEPT)
        return -np.log1p(-_mh_uniform53(w[0], w[1]))


def mh_jump(draws, transforms, call, proposal_scale=2.4):
    """mcmc.py:290-296: the jump of proposer call ``call``: zero for the parameters of slower blocks."""
    ib, direction = draws.direction(call)
    ndim = int(draws.block_starts[-1])
    jump = np.zeros(ndim)
    jump[draws.block_starts[ib]:] += transforms[ib].dot(direction * proposal_scale)
    return jump


def mh_sample(log_prob_fn, start, draws, transforms, proposal_scale=2.4, iterations=None, ntries=None, thin_by=1, vectorize=1, max_tries=1000, start_log_prob=None):
    """MHSampler.sample without dragging (mcmc.py:45-105): ``vectorize`` proposals per try from the current state, the first accepted one is taken, the rejected
    ones before it add to the weight.  Stops after ``iterations`` + 1 accepted moves (the reference's loop) or after ``ntries`` tries (the device's unit).
    Returns (chain, weight, log_prob) of the recorded states and the final (coords, log_prob, weight)."""
    coords = np.array(start, dtype='f8')
    log_prob = float(log_prob_fn(coords[None, :])[0]) if start_log_prob is None else float(start_log_prob)
    weight, states = 1, []
    tries, it, call = 0, 0, 0

    def mh_accept(proposal_log_prob, current_log_prob, call):   # mcmc.py:107-112
        if proposal_log_prob == -np.inf: return False
        if proposal_log_prob > current_log_prob: return True
        return draws.exponential(call) > (current_log_prob - proposal_log_prob)

    while True:
        if iterations is not None and it > iterations: break
        accept = False
        for itry in range(max_tries):
            if ntries is not None and tries >= ntries: break
            proposals = coords + np.array([mh_jump(draws, transforms, call + i, proposal_scale=proposal_scale) for i in range(vectorize)])
            proposals_log_prob = np.asarray(log_prob_fn(proposals), dtype='f8')
            base = call
            call += vectorize
            tries += 1
            for i in range(vectorize):
                accept = mh_accept(proposals_log_prob[i], log_prob, base + i)
                if accept: break
                weight += 1
            if accept:
                if it > 0 and it % thin_by == 0:
                    states.append((coords, log_prob, weight))
                coords, log_prob, weight = proposals[i], float(proposals_log_prob[i]), 1
                break
        if ntries is not None and tries >= ntries and not accept: break
        if not accept:
            raise ValueError('Could not find finite log posterior after {:d} tries'.format(max_tries))
        it += 1
        if ntries is not None and tries >= ntries: break
    ndim = len(coords)
    chain = np.array([s[0] for s in states]).reshape(len(states), ndim)
    return chain, np.array([s[2] for s in states], dtype='i8'), np.array([s[1] for s in states], dtype='f8'), (coords, log_prob, weight)
