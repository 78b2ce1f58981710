// dl_tns.h -- one-loop tables of the TNS model (Taruya, Nishimichi & Saito 2010) as the reference computes them (full_shape.py:688-971), for batches of points.
//
// What the reference does per point (tns_pt, full_shape.py:749-833): for every table wavenumber k (n11 = 1.6 x the theory's k) a double sum over the template's own
// 500 wavenumbers q (trapezoid) and 10 cosines mu (Gauss-Legendre) of ~34 integrands built from the mode-coupling kernels F2 / G2, the tidal kernel S, the A / B
// polynomial kernels, the template P(q) and P(|k - q|) -- the latter by LINEAR interpolation of the template.  Everything except the two spectra is geometry: it does
// not depend on the point.  So
//     table_t (k) = sum_{mu, q} C_t[k, mu, q] P(q) P_lin(|k - q|)  (27 bilinear tables)  +  sum_q L_u[k, q] P(q)  (12 linear ones)  +  const x sum_q jq P(q)^2,
// with P_lin(|k - q|) = w0[k, mu, q] P(j0) + w1 P(j0 + 1).  The geometry (C, L, j0, w0, w1) is computed ONCE on the device when the context is created
// (dl_tns_geometry_*_kernel); per evaluation the bilinear part is a batched GEMM per k,  [points x (mu, q)] . [(mu, q) x 32],  whose left operand
// G = P(q) (w0 P(j0) + w1 P(j0 + 1)) is formed in registers from the points' templates held in LDS (dl_tns_loop_kernel: fp64 MFMA, 32 points x one k per workgroup,
// the coefficients stream from L2: every workgroup of one k runs on the same XCD).  The 29 tables of a point then go through the not-a-knot spline in log10 k, the
// AP distortion, the damping, the mu^2n polynomials of the A / B terms and the Legendre projection (dl_tns_assemble_kernel) into the same power rows the other
// theory kernels write.
#pragma once
#include <stdint.h>
#include <math.h>

#include "dl_fullshape.h"

#define DL_TNS_NCOL 32        // bilinear tables per k (27 used)
#define DL_TNS_NLIN 16        // linear tables per k (12 used)
#define DL_TNS_NTAB 29        // tables per k, in the order of full_shape.py:882-883
#define DL_TNS_NREC 48        // sums per (point, k) written by the loop kernel: 32 bilinear, 16 linear
#define DL_TNS_PTS 32         // points per workgroup of the loop kernel
#define DL_TNS_MAX_MU 16

// columns of the bilinear coefficient matrix
enum { DL_TC_B2D = 0, DL_TC_BS2D, DL_TC_B2T, DL_TC_BS2T, DL_TC_B22, DL_TC_B2S2, DL_TC_BS22, DL_TC_22DD, DL_TC_22DT, DL_TC_22TT, DL_TC_TA0 = 10, DL_TC_B0 = 15 };
// columns of the linear coefficient matrix: P(k) by interpolation; sigma_3^2; P13 kernels (density, velocity); the four distinct A-term kernels of tns_kernels;
// the four distinct A-term kernels multiplying P(k) P(|k - q|) (folded over the interpolation)
enum { DL_TL_PK = 0, DL_TL_SIG3, DL_TL_13D, DL_TL_13T, DL_TL_KA0 = 4, DL_TL_EA0 = 8 };

struct DlTnsDev {
    int32_t n11, n_q, nqp, n_mu;     // table wavenumbers; template wavenumbers (nqp: rounded up to a multiple of 4); loop cosines
    int32_t K, fog, Kp, pad1;        // K = n_mu * n_q pairs (mu, q), Kp: rounded up to whole rounds of 16 pairs (zero coefficients); fog: 0 lorentzian, 1 gaussian (full_shape.py:870-873)
    double k11_0, inv_dk11;          // k11 = linspace: interval index of the spline evaluation
    double sumw;                     // sum of the cosine weights
    const double* k11;               // [n11]
    const double* x11;               // [n11] log10(k11)
    const double* q;                 // [nqp] template wavenumbers (padding: last value)
    const double* jq;                // [nqp] q^2 wq / (4 pi^2) (padding: 0)
    const double* mus;               // [n_mu] loop cosines, then [n_mu] weights
    const int32_t* geomj;            // [n11][Kp][16][2] per lane of a 16-point tile: LDS byte offsets of P(j0) (interval of |k - q| in the template's wavenumbers) and of P(q)
    const double* geomw;             // [n11][Kp][2] w0, w1
    const double* coef;              // [n11][Kp][16][2]: columns c and 16 + c interleaved
    const double* lin;               // [n11][nqp][16]
    const double* spT;               // [n11][n11] transposed operator y -> second derivatives of the not-a-knot spline on x11
};

// ---- geometry of one (k, mu, q): everything of tns_pt's get_terms that does not depend on the spectra ----------------------------------------------
struct DlTnsGeom {
    double c[27];     // bilinear coefficients (times wmu jq)
    double ca[5];     // A-term kernels multiplying P(k) P(|k - q|) (times wmu jq)
    double sig3;      // sigma_3^2 integrand (times wmu jq)
    double r;         // |k - q|
};

DL_HD void dl_tns_geometry(double k, double q, double jq, double mu, double wmu, DlTnsGeom& g) {
    // full_shape.py:766-779
    const double kdq = k * q * mu;
    const double kq2 = k * k - 2. * kdq + q * q;
    const double qdkq = kdq - q * q;
    const double c2 = qdkq * qdkq / (q * q * kq2);
    const double half = 0.5 * qdkq * (1. / (q * q) + 1. / kq2);
    const double F2d = 5. / 7. + half + 2. / 7. * c2;
    const double F2t = 3. / 7. + half + 4. / 7. * c2;
    const double S = c2 - 1. / 3.;
    const double D = 2. / 7. * (mu * mu - 1.);
    const double base = wmu * jq;
    g.r = sqrt(kq2);
    g.c[DL_TC_B2D] = base * F2d;
    g.c[DL_TC_BS2D] = base * F2d * S;
    g.c[DL_TC_B2T] = base * F2t;
    g.c[DL_TC_BS2T] = base * F2t * S;
    g.c[DL_TC_B22] = 0.5 * base;
    g.c[DL_TC_B2S2] = 0.5 * base * S;
    g.c[DL_TC_BS22] = 0.5 * base * S * S;
    g.c[DL_TC_22DD] = 2. * base * F2d * F2d;
    g.c[DL_TC_22DT] = 2. * base * F2d * F2t;
    g.c[DL_TC_22TT] = 2. * base * F2t * F2t;
    g.sig3 = base * 105. / 16. * (D * S + 8. / 63.);
    // A term, full_shape.py:797-810
    const double x = q / k, xmu = kq2 / (k * k);
    const double x2 = x * x, x3 = x2 * x, mu2 = mu * mu, mu4 = mu2 * mu2;
    const double fa = base / x2 / (xmu * xmu);
    const double kA0 = -x3 / 7. * (mu + 6. * mu2 * mu + x2 * mu * (-3. + 10. * mu2) + x * (-3. + mu2 - 12. * mu4));
    const double kA1 = x2 * x2 / 14. * (mu2 - 1.) * (-1. + 7. * x * mu - 6. * mu2);
    const double kA2 = x3 / 14. * (x2 * mu * (13. - 41. * mu2) - 4. * (mu + 6. * mu2 * mu) + x * (5. + 9. * mu2 + 42. * mu4));
    const double kA4 = x3 / 14. * (1. - 7. * x * mu + 6. * mu2) * (-2. * mu + x * (-1. + 3. * mu2));
    g.ca[0] = fa * kA0; g.ca[1] = fa * kA1; g.ca[2] = fa * kA2; g.ca[3] = fa * kA1; g.ca[4] = fa * kA4;
    const double t0 = 1. / 7. * (mu + x - 2. * x * mu2) * (3. * x + 7. * mu - 10. * x * mu2);
    const double t1 = x / 14. * (mu2 - 1.) * (3. * x + 7. * mu - 10. * x * mu2);
    const double t2 = 1. / 14. * (28. * mu2 + x * mu * (25. - 81. * mu2) + x2 * (1. - 27. * mu2 + 54. * mu4));
    const double t3 = x / 14. * (1. - mu2) * (x - 7. * mu + 6. * x * mu2);
    const double t4 = 1. / 14. * (x - 7. * mu + 6. * x * mu2) * (-2. * mu - x + 3. * x * mu2);
    g.c[DL_TC_TA0 + 0] = fa * t0; g.c[DL_TC_TA0 + 1] = fa * t1; g.c[DL_TC_TA0 + 2] = fa * t2; g.c[DL_TC_TA0 + 3] = fa * t3; g.c[DL_TC_TA0 + 4] = fa * t4;
    // B term, full_shape.py:812-826 (n, a, b as commented there)
    const double fb = base / (x2 * xmu);
    const double m21 = mu2 - 1.;
    double* b = g.c + DL_TC_B0;
    b[0] = fb * x2 * m21 / 2.;
    b[1] = fb * 3. * x2 * m21 * m21 / 8.;
    b[2] = fb * 3. * x2 * x2 * m21 * m21 / xmu / 8.;
    b[3] = fb * 5. * x2 * x2 * m21 * m21 * m21 / xmu / 16.;
    b[4] = fb * x * (x + 2. * mu - 3. * x * mu2) / 2.;
    b[5] = fb * -3. * x * m21 * (-x - 2. * mu + 5. * x * mu2) / 4.;
    b[6] = fb * 3. * x2 * m21 * (-2. + x2 + 6. * x * mu - 5. * x2 * mu2) / xmu / 4.;
    b[7] = fb * -3. * x2 * m21 * m21 * (6. - 5. * x2 - 30. * x * mu + 35. * x2 * mu2) / xmu / 16.;
    b[8] = fb * x * (4. * mu * (3. - 5. * mu2) + x * (3. - 30. * mu2 + 35. * mu4)) / 8.;
    b[9] = fb * x * (-8. * mu + x * (-12. + 36. * mu2 + 12. * x * mu * (3. - 5. * mu2) + x2 * (3. - 30. * mu2 + 35. * mu4))) / xmu / 8.;
    b[10] = fb * 3. * x * m21 * (-8. * mu + x * (-12. + 60. * mu2 + 20. * x * mu * (3. - 7. * mu2) + 5. * x2 * (1. - 14. * mu2 + 21. * mu4))) / xmu / 16.;
    b[11] = fb * x * (8. * mu * (-3. + 5. * mu2) - 6. * x * (3. - 30. * mu2 + 35. * mu4) + 6. * x2 * mu * (15. - 70. * mu2 + 63. * mu4)
                      + x3 * (5. - 21. * mu2 * (5. - 15. * mu2 + 11. * mu4))) / xmu / 16.;
}

// Linear interpolation weights of numpy.interp(r, q, ., left = 0, right = 0): r in [q[j], q[j + 1]] -> (j, w0, w1); outside the table both weights are zero.
DL_HD void dl_tns_interp_weights(const double* q, int n_q, double r, int& j, double& w0, double& w1) {
    if (!(r >= q[0]) || !(r <= q[n_q - 1])) { j = 0; w0 = 0.; w1 = 0.; return; }
    int lo = 0, hi = n_q - 1;                 // q[lo] <= r <= q[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (q[mid] <= r) lo = mid; else hi = mid; }
    j = lo;
    w1 = (r - q[lo]) / (q[lo + 1] - q[lo]);
    w0 = 1. - w1;
}

// Angle-integrated kernels of tns_kernels (full_shape.py:688-746) at x = q / k, WITHOUT the jq factor: F3-type (density, velocity) and the four distinct A kernels.
DL_HD void dl_tns_kernels13(double x, double& ff, double& gg) {
    const double x2 = x * x, x4 = x2 * x2;
    const double dx = x - 1.;
    if (fabs(dx) < 0.01) {
        ff = -11. / 126. + dx / 126. - 29. / 252. * dx * dx;
        gg = -3. / 14. - 5. / 42. * dx - 1. / 84. * dx * dx;
    } else if (x > 10.) {
        ff = -61. / 630. + 2. / 105. / x2 - 10. / 1323. / x4;
        gg = -3. / 10. + 26. / 245. / x2 - 38. / 2205. / x4;
    } else {
        const double lg = 2. * log(fabs((x - 1.) / (x + 1.)));
        const double cube = (1. / x - x) * (1. / x - x) * (1. / x - x);
        ff = (6. / x2 - 79. + 50. * x2 - 21. * x4 + 0.75 * cube * (2. + 7. * x2) * lg) / 504.;
        gg = (6. / x2 - 41. + 2. * x2 - 3. * x4 + 0.75 * cube * (2. + x2) * lg) / 168.;
    }
    ff /= x2; gg /= x2;
}

DL_HD void dl_tns_kernels_a(double x, double* ka /* [4]: kernels 0, 1 (= 3), 2, 4 */) {
    const double x2 = x * x, x4 = x2 * x2, x6 = x4 * x2, x8 = x4 * x4;
    if (x < 1e-4) {
        ka[0] = 8. * x8 / 735. + 24. * x6 / 245. - 24. * x4 / 35. + 8. * x2 / 7. - 2. / 3.;
        ka[1] = -16. * x8 / 8085. - 16. * x6 / 735. + 48. * x4 / 245. - 16. * x2 / 35.;
        ka[2] = 32. * x8 / 1617. + 128. * x6 / 735. - 288. * x4 / 245. + 64. * x2 / 35. - 4. / 3.;
        ka[3] = 24. * x8 / 2695. + 8. * x6 / 105. - 24. * x4 / 49. + 24. * x2 / 35. - 2. / 3.;
    } else if (x > 1e2) {
        ka[0] = 2. / 105. - 24. / (245. * x2) - 8. / (735. * x4) - 8. / (2695. * x6) - 8. / (7007. * x8);
        ka[1] = -16. / 35. + 48. / (245. * x2) - 16. / (735. * x4) - 16. / (8085. * x6) - 16. / (35035. * x8);
        ka[2] = -44. / 105. - 32. / (735. * x4) - 64. / (8085. * x6) - 96. / (35035. * x8);
        ka[3] = -46. / 105. + 24. / (245. * x2) - 8. / (245. * x4) - 8. / (1617. * x6) - 8. / (5005. * x8);
    } else {
        const double lx = fabs(x - 1.) > 1e-16 ? log(fabs((x + 1.) / (x - 1.))) : 0.;
        const double x3 = x2 * x, d2 = x2 - 1., d3 = d2 * d2 * d2;
        ka[0] = -1. / 84. / x * (2. * x * (19. - 24. * x2 + 9. * x4) - 9. * d3 * lx);
        ka[1] = 1. / 112. / x3 * (2. * x * (x2 + 1.) * (3. - 14. * x2 + 3. * x4) - 3. * d3 * d2 * lx);
        ka[2] = 1. / 336. / x3 * (2. * x * (9. - 185. * x2 + 159. * x4 - 63. * x6) + 9. * d3 * (7. * x2 + 1.) * lx);
        ka[3] = 1. / 336. / x3 * (2. * x * (9. - 109. * x2 + 63. * x4 - 27. * x6) + 9. * d3 * (3. * x2 + 1.) * lx);
    }
    for (int i = 0; i < 4; ++i) ka[i] /= x2;
}

// ---- the 29 table entries of one (point, k) from the bilinear sums S [32], the linear sums Lv [16] and qq = sum_q jq P(q)^2 (full_shape.py:829-847) ----
DL_HD double dl_tns_table_entry(int r, const double* S, const double* Lv, double qq, double sumw) {
    const double pk = Lv[DL_TL_PK];
    switch (r) {
        case 0: return pk;                                                                  // pk11
        case 1: return pk + S[DL_TC_22DD] + 2. * Lv[DL_TL_13D] * pk;                        // pk_dd
        case 2: return S[DL_TC_B2D];
        case 3: return S[DL_TC_BS2D];
        case 4: return Lv[DL_TL_SIG3] * pk;                                                 // pk_sig3sq
        case 5: return S[DL_TC_B22] - 0.5 * sumw * qq;
        case 6: return S[DL_TC_B2S2] - 0.5 * sumw * (2. / 3.) * qq;
        case 7: return S[DL_TC_BS22] - 0.5 * sumw * (4. / 9.) * qq;
        case 8: return pk + S[DL_TC_22DT] + (Lv[DL_TL_13D] + Lv[DL_TL_13T]) * pk;           // pk_dt: P13 = (P13_dd + P13_tt) / 2
        case 9: return S[DL_TC_B2T];
        case 10: return S[DL_TC_BS2T];
        case 11: return pk + S[DL_TC_22TT] + 2. * Lv[DL_TL_13T] * pk;                       // pk_tt
        default: break;
    }
    if (r < 17) {   // A_i = sum (kernel_A P(k) + kernel_tA P(q)) P(|k - q|) ... + P(k) sum kernel_a P(q)
        const int i = r - 12;
        const int u = (i == 0) ? 0 : (i == 1 || i == 3) ? 1 : (i == 2) ? 2 : 3;
        return S[DL_TC_TA0 + i] + pk * (Lv[DL_TL_EA0 + u] + Lv[DL_TL_KA0 + u]);
    }
    if (r < 29) return S[DL_TC_B0 + (r - 17)];
    return 0.;
}

// ---- coefficients of the 29 (+ pk11 again) tables in the five mu'^2n polynomials of P(k, mu) (full_shape.py:889-899, 957-971) -------------------------
//   P(k, mu) = jac damping sum_n mu'^2n Q_n(k'),   Q_n = sum_r cvec[n][r] table_r
DL_HD double dl_tns_combine_coef(int n, int r, double f, double b1, double b2, double bs, double b3) {
    const double bs2 = bs - 4. / 7. * (b1 - 1.), b3nl = b3 + 32. / 315. * (b1 - 1.);
    const double f2 = f * f, f3 = f2 * f, f4 = f2 * f2;
    if (n == 0) {
        switch (r) {
            case 1: return b1 * b1;                       // pk_dd
            case 2: return 2. * b1 * b2;                  // pk_b2d
            case 3: return 2. * b1 * bs2;                 // pk_bs2d
            case 4: return 2. * b1 * b3nl + b3nl;         // pk_sig3sq (twice in full_shape.py:965-968, the second time without f mu^2: as coded)
            case 5: return b2 * b2;
            case 6: return 2. * b2 * bs2;
            case 7: return bs2 * bs2;
            default: return 0.;
        }
    }
    if (n == 1) {
        switch (r) {
            case 8: return 2. * b1 * f;                   // pk_dt
            case 9: return b2 * f;                        // pk_b2t
            case 12: return b1 * b1 * f;                  // A0
            case 13: return b1 * f2;                      // A1
            case 17: return b1 * b1 * f2;                 // B0
            case 18: case 19: return -b1 * f3;            // B1 + B2
            case 20: return f4;                           // B3
            default: return 0.;
        }
    }
    if (n == 2) {
        switch (r) {
            case 11: return f2;                           // pk_tt
            case 14: return b1 * f2;                      // A2
            case 15: return f3;                           // A3
            case 21: return b1 * b1 * f2;                 // B4
            case 22: case 23: return -b1 * f3;            // B5 + B6
            case 24: return f4;                           // B7
            default: return 0.;
        }
    }
    if (n == 3) {
        switch (r) {
            case 16: return f3;                           // A4
            case 25: case 26: return -b1 * f3;            // B8 + B9
            case 27: return f4;                           // B10
            default: return 0.;
        }
    }
    if (n == 4) return r == 28 ? f4 : 0.;                 // B11
    return r == 0 ? 1. : 0.;                              // n = 5: pk11 (counter terms)
}

// ---- assembly, per theory wavenumber ik: the polynomials Q_n (rows of Q, second derivatives in M, row stride ldq) at the AP-distorted (k, mu), damping, projection,
//      shot noise and EFT-like terms (full_shape.py:865-899, 957-971, 628-634): out [n_ell][n_kin] (+ the projected linear spectrum, monopole, at out [n_in + ik] when
//      counter terms need it).  murec [n_mu][8]: factorap, mu'^2, jac w_ell [5], jac w_(ell = 0);  sc: qper, sigmav, sn0 / nd.  Shared by dl_tns_assemble_kernel and the
//      CPU emulation. ----
DL_HD void dl_tns_eval_k(const DlObsDev& o, int fog, double k11_0, double inv_dk11, const double* x11, int n11, int ldq, int nq, const double* Q, const double* M,
                         const double* murec, const double* sc, const double* th, int ik, double* out) {
    const double sigmav = sc[1], sn0nd = sc[2];
    double pl[DL_MAX_ELL] = {0., 0., 0., 0., 0.}, dd0 = 0.;
    const double kq = o.kin[ik] / sc[0];
    for (int m = 0; m < o.n_mu; ++m) {
        const double kap = kq * murec[8 * m], m2 = murec[8 * m + 1];
        int i = (int)floor((kap - k11_0) * inv_dk11);
        i = i < 0 ? 0 : (i > n11 - 2 ? n11 - 2 : i);
        const double x = log10(kap), xl = x11[i], xr = x11[i + 1], h = xr - xl;
        const double a = (xr - x) / h, bb = (x - xl) / h;
        const double ca = (a * a * a - a) * h * h / 6., cb = (bb * bb * bb - bb) * h * h / 6.;
        double v[6];
        for (int n = 0; n < nq; ++n) {
            const double* Qn = Q + (size_t)n * ldq; const double* Mn = M + (size_t)n * ldq;
            v[n] = a * Qn[i] + bb * Qn[i + 1] + ca * Mn[i] + cb * Mn[i + 1];
        }
        const double sk = sigmav * kap, s2 = sk * sk * m2;   // (sigmav kap muap)^2
        const double damp = fog == 0 ? 1. / ((1. + s2 / 2.) * (1. + s2 / 2.)) : exp(-s2);   // full_shape.py:870-873
        const double pkmu = damp * (v[0] + m2 * (v[1] + m2 * (v[2] + m2 * (v[3] + m2 * v[4]))));
        for (int l = 0; l < DL_MAX_ELL; ++l) pl[l] = fma(murec[8 * m + 2 + l], pkmu, pl[l]);
        if (nq == 6) dd0 = fma(murec[8 * m + 7], damp * v[5], dd0);
    }
    for (int l = 0; l < o.n_ell; ++l) {
        double val = pl[l] + sn0nd;                         // full_shape.py:961: on EVERY multipole
        const size_t ix = (size_t)l * o.n_kin + ik;
        for (int c = 0; c < o.n_ct; ++c) val += o.ct_matrix[ix * o.n_ct + c] * 0.5 * (dl_get(o.ct_in[c][0], th) + dl_get(o.ct_in[c][1], th)) * dd0;   // full_shape.py:630, 633
        for (int c = 0; c < o.n_sn; ++c) val += o.sn_matrix[ix * o.n_sn + c] * dl_get(o.sn_in[c], th) / o.nd;                                         // full_shape.py:631, 634
        out[ix] = val;
    }
    if (nq == 6) out[o.n_in + ik] = dd0;
}

// per-mu records of the assembly (murec, see dl_tns_eval_k)
DL_HD void dl_tns_mu_record(const DlObsDev& o, double qpar, double qper, int m, double* murec) {
    const double jac = 1. / (qpar * qper * qper);
    const double mu = o.mu[m], rq = qper / qpar;
    const double x = 1. + mu * mu * (rq * rq - 1.);       // factorap^2 (tgc/base.py:216-222)
    murec[8 * m] = sqrt(x);
    murec[8 * m + 1] = mu * mu * rq * rq / x;
    for (int l = 0; l < DL_MAX_ELL; ++l) murec[8 * m + 2 + l] = l < o.n_ell ? jac * o.wmu[l * o.n_mu + m] : 0.;
    murec[8 * m + 7] = o.ell0 >= 0 ? jac * o.wmu[o.ell0 * o.n_mu + m] : 0.;
}

// LDS of the assembly kernel (dl_tns.hip): Q [16][ldq] | M [16][ldq] | per point: cvec [6][32] | mu records [DL_MAX_MU][8] | scalars [8] | out [n_in + n_kin]
DL_HD int dl_tns_ldq(int n11) { return n11 | 1; }
DL_HD size_t dl_tns_assemble_point_doubles(int n_in, int n_kin) { return 6 * 32 + (size_t)8 * DL_MAX_MU + 8 + n_in + n_kin; }
DL_HD size_t dl_tns_assemble_doubles(int n11, int n_in, int n_kin, int ppw) { return (size_t)32 * dl_tns_ldq(n11) + ppw * dl_tns_assemble_point_doubles(n_in, n_kin); }

// host side (dl_tns.hip)
#ifdef __HIPCC__
struct DlTnsPlan;
DlTnsPlan* dl_tns_create(const double* k11, int n11, const double* q, int n_q, const double* mus, const double* wmus, int n_mu, int fog, const char** err);
void dl_tns_destroy(DlTnsPlan* plan);
size_t dl_tns_bytes(const DlTnsPlan* plan);
// power rows of B points of one observable (theory kind 4); obs.tns_plan holds the plan
void dl_launch_tns(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power, hipStream_t stream);
// diagnostics / parity: the 29 raw tables [B, 29, n11] of the points
int dl_tns_tables(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* tables_dev, hipStream_t stream);
#endif
