// dl_fullshape.h -- per-point arithmetic of the full-shape theory kernel.
//
// One workgroup evaluates one (parameter point, observable): ShapeFit template rescaling at the
// N_t knots, not-a-knot cubic-spline moments by a segmented Thomas sweep, AP remap + spline evaluation
// at (k, mu'), Gauss-Legendre projection onto multipoles, tracer bias combination.
// The body is written as barrier-separated *phases*, each a function of (tid, nthreads) acting on
// workgroup-shared arrays, so that the HIP kernel (dl_kernels.hip) and the CPU emulation used by the
// `not gpu` tests (tests/csrc/emulate_fullshape.cpp) run literally the same code.
//
// Reference arithmetic restated here (paths relative to /root/reference/desilike):
//   phase0: APEffect.calculate + ap_k_mu          theories/galaxy_clustering/base.py:211-223, 341-353
//   phase1: ShapeFitPowerSpectrumTemplate.calculate theories/galaxy_clustering/power_template.py:747-761
//   phase2: interp1d(..., 'cubic') knots -> spline  jax.py:263-265 (scipy not-a-knot; here: moment form)
//   phase3: KaiserPowerSpectrumMultipoles.calculate theories/galaxy_clustering/full_shape.py:488-500,
//           to_poles tgc/base.py:206-208, KaiserTracer...calculate full_shape.py:545-550, EFT add-on 628-634
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DL_HD __host__ __device__ __forceinline__
#else
#define DL_HD inline
#endif

#define DL_MAX_ELL 5
#define DL_MAX_MU 32
#define DL_MAX_EFT 8
#define DL_FS_THREADS 256

struct DlInput {
    int32_t col;    // >= 0: column of theta; < 0: use `value`
    int32_t pad;
    double value;
};

DL_HD double dl_get(const DlInput& in, const double* th) { return in.col >= 0 ? th[in.col] : in.value; }

struct DlObsDev {
    int32_t theory, templ, apmode, transform;
    int32_t n_ell, n_kin, n_mu, n_t;
    int32_t n_in, ell0, n_ct, n_sn;
    int32_t seg_len, seg_warm, n_seg, fixed_spline;
    int64_t col_offset;  // first column of this observable in a row of the (concatenated) power buffer
    double eta, f_fid, a, nd, x0, inv_hx;
    double end0a, end0b, end1a, end1b;  // not-a-knot end relations: M[0] = end0a M[1] + end0b M[2]; M[n-1] = end1a M[n-2] + end1b M[n-3]
    DlInput qpar, qper, qiso, qap, df, dm, dn, sigpar, sigper, b1X, b1Y, sn0;
    DlInput ct_in[DL_MAX_EFT][2];
    DlInput sn_in[DL_MAX_EFT];
    const double *kin, *lkin, *mu, *wmu;          // [n_kin], log10(kin) [n_kin], [n_mu], [n_ell * n_mu]
    const double *x_t, *pk_fid, *sf_th, *sf_lg;   // log10(k_t), fiducial P, tanh(a ln(k/kp)), ln(k/kp): all [n_t]
    const double *ih;                              // 1 / (x_t[j+1] - x_t[j]) [n_t - 1]
    const double *sp_A, *sp_nC, *sp_inv;           // Thomas sweep coefficients of the reduced system [n_t - 2]
    const double *M_fixed;                         // moments of the fiducial table (fixed templates) [n_t]
    const double *ct_matrix, *sn_matrix;           // [n_ell, n_kin, n_ct], [n_ell, n_kin, n_sn]
};

// Layout of the small per-point scratch `pt` (doubles) in workgroup-shared memory
enum {
    DL_PT_QPAR = 0, DL_PT_QPER, DL_PT_JAC, DL_PT_F, DL_PT_B1X, DL_PT_B1Y, DL_PT_SN0ND, DL_PT_DAMP,
    DL_PT_LQ = 8,                        // log10(F_m / qper)
    DL_PT_MUP2 = DL_PT_LQ + DL_MAX_MU,   // mu'^2
    DL_PT_FAC = DL_PT_MUP2 + DL_MAX_MU,  // F_m
    DL_PT_SD = DL_PT_FAC + DL_MAX_MU,    // sigmapar^2 mu'^2 + sigmaper^2 (1 - mu'^2)
    DL_PT_CT = DL_PT_SD + DL_MAX_MU,     // 0.5 (ctX + ctY)
    DL_PT_SN = DL_PT_CT + DL_MAX_EFT,    // sn / nd
    DL_PT_SIZE = DL_PT_SN + DL_MAX_EFT
};

struct DlFsShared {
    double* y;   // [n_t] template power at the knots
    double* M;   // [n_t] spline moments (second derivatives); M[0], M[n_t-1] are formed on the fly
    double* z;   // [n_t] forward-sweep scratch
    double* pt;  // [DL_PT_SIZE]
};

DL_HD size_t dl_fs_shared_doubles(int n_t) { return 3 * (size_t)n_t + DL_PT_SIZE; }

DL_HD void dl_ap_qparqper(const DlObsDev& o, const double* th, double& qpar, double& qper) {
    // theories/galaxy_clustering/base.py:341-350
    switch (o.apmode) {
        case 1: qpar = qper = dl_get(o.qiso, th); break;
        case 2: { double qap = dl_get(o.qap, th); qpar = pow(qap, 1. - o.eta); qper = pow(qap, -o.eta); break; }
        case 3: { double qiso = dl_get(o.qiso, th), qap = dl_get(o.qap, th); qpar = qiso * pow(qap, 1. - o.eta); qper = qiso * pow(qap, -o.eta); break; }
        default: qpar = dl_get(o.qpar, th); qper = dl_get(o.qper, th);
    }
}

// phase 0 + 1 (no barrier needed between them): per-point scalars, per-mu AP factors, template at the knots
DL_HD void dl_fs_phase01(int tid, int nthr, const DlObsDev& o, const double* th, const DlFsShared& s) {
    if (tid < o.n_mu || tid == 0) {
        double qpar, qper;
        dl_ap_qparqper(o, th, qpar, qper);
        double sigpar = dl_get(o.sigpar, th), sigper = dl_get(o.sigper, th);
        if (tid < o.n_mu) {
            // ap_k_mu, tgc/base.py:216-222: factorap = sqrt(1 + mu^2 (1/qap^2 - 1)); muap = mu / qap / factorap
            double qap = qpar / qper;
            double mu = o.mu[tid];
            double fac = sqrt(1. + mu * mu * (1. / (qap * qap) - 1.));
            double mup = mu / qap / fac;
            s.pt[DL_PT_FAC + tid] = fac;
            s.pt[DL_PT_LQ + tid] = log10(fac / qper);  // log10(kap) = log10(k) + log10(factorap / qper)
            s.pt[DL_PT_MUP2 + tid] = mup * mup;
            // full_shape.py:492: sigmapar^2 muap^2 + sigmaper^2 (1 - muap^2)
            s.pt[DL_PT_SD + tid] = sigpar * sigpar * (mup * mup) + sigper * sigper * (1. - mup * mup);
        }
        if (tid == 0) {
            s.pt[DL_PT_QPAR] = qpar;
            s.pt[DL_PT_QPER] = qper;
            s.pt[DL_PT_JAC] = 1. / (qpar * qper * qper);           // tgc/base.py:217
            s.pt[DL_PT_F] = o.f_fid * dl_get(o.df, th);            // power_template.py:757
            s.pt[DL_PT_B1X] = dl_get(o.b1X, th);
            s.pt[DL_PT_B1Y] = dl_get(o.b1Y, th);
            s.pt[DL_PT_SN0ND] = dl_get(o.sn0, th) / o.nd;          // full_shape.py:549
            s.pt[DL_PT_DAMP] = (sigpar != 0. || sigper != 0.) ? 1. : 0.;
            for (int c = 0; c < o.n_ct; ++c)                       // full_shape.py:630
                s.pt[DL_PT_CT + c] = 0.5 * (dl_get(o.ct_in[c][0], th) + dl_get(o.ct_in[c][1], th));
            for (int c = 0; c < o.n_sn; ++c)                       // full_shape.py:631
                s.pt[DL_PT_SN + c] = dl_get(o.sn_in[c], th) / o.nd;
        }
    }
    if (o.templ == 1) {
        // power_template.py:749: exp(dm / a * tanh(a * log(k / kp)) + dn * log(k / kp))
        double dm_a = dl_get(o.dm, th) / o.a, dn = dl_get(o.dn, th);
        for (int j = tid; j < o.n_t; j += nthr) s.y[j] = o.pk_fid[j] * exp(dm_a * o.sf_th[j] + dn * o.sf_lg[j]);
    } else {
        for (int j = tid; j < o.n_t; j += nthr) { s.y[j] = o.pk_fid[j]; s.M[j] = o.M_fixed[j]; }
    }
}

// phase 2a: right-hand side of the reduced (n_t - 2 unknowns) not-a-knot system, pre-multiplied by the pivots
DL_HD void dl_fs_phase2a(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    if (o.fixed_spline) return;
    int m = o.n_t - 2;
    for (int i = tid; i < m; i += nthr) {
        double r = 6. * ((s.y[i + 2] - s.y[i + 1]) * o.ih[i + 1] - (s.y[i + 1] - s.y[i]) * o.ih[i]);
        s.M[i + 1] = r * o.sp_inv[i];
    }
}

// phase 2b: forward sweep z_i = A_i z_{i-1} + B_i, segmented with an exponentially-decaying warm-up
DL_HD void dl_fs_phase2b(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    if (o.fixed_spline || tid >= o.n_seg) return;
    int m = o.n_t - 2;
    int start = tid * o.seg_len, end = start + o.seg_len;
    if (end > m) end = m;
    if (start >= m) return;
    int i0 = start - o.seg_warm;
    if (i0 < 0) i0 = 0;
    double zz = 0.;
    for (int i = i0; i < end; ++i) {
        zz = fma(o.sp_A[i], zz, s.M[i + 1]);
        if (i >= start) s.z[i] = zz;
    }
}

// phase 2c: backward sweep u_i = z_i - c'_i u_{i+1}; u_i = M[i + 1]
DL_HD void dl_fs_phase2c(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    if (o.fixed_spline || tid >= o.n_seg) return;
    int m = o.n_t - 2;
    int start = tid * o.seg_len, end = start + o.seg_len;
    if (end > m) end = m;
    if (start >= m) return;
    int i1 = end - 1 + o.seg_warm;
    if (i1 > m - 1) i1 = m - 1;
    double uu = 0.;
    for (int i = i1; i >= start; --i) {
        uu = fma(o.sp_nC[i], uu, s.z[i]);
        if (i < end) s.M[i + 1] = uu;
    }
}

// spline value at abscissa x (log10 k'), extrapolating with the end pieces like scipy's fill_value='extrapolate'
DL_HD double dl_spline_eval(const DlObsDev& o, const DlFsShared& s, double x) {
    int n = o.n_t;
    int j = (int)floor((x - o.x0) * o.inv_hx);
    if (j < 0) j = 0;
    if (j > n - 2) j = n - 2;
    while (j > 0 && x < o.x_t[j]) --j;
    while (j < n - 2 && x >= o.x_t[j + 1]) ++j;
    double xl = o.x_t[j], xr = o.x_t[j + 1];
    double ihj = o.ih[j];
    double h = xr - xl;
    double a = (xr - x) * ihj, b = (x - xl) * ihj;
    double Ml = (j == 0) ? (o.end0a * s.M[1] + o.end0b * s.M[2]) : s.M[j];
    double Mr = (j == n - 2) ? (o.end1a * s.M[n - 2] + o.end1b * s.M[n - 3]) : s.M[j + 1];
    return a * s.y[j] + b * s.y[j + 1] + ((a * a * a - a) * Ml + (b * b * b - b) * Mr) * (h * h * (1. / 6.));
}

// phase 3: (k, mu) evaluation, multipole projection, tracer combination; writes power (and tables)
DL_HD void dl_fs_phase3(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, double* power_row, double* tables_row) {
    const double jac = s.pt[DL_PT_JAC], f = s.pt[DL_PT_F], qper = s.pt[DL_PT_QPER];
    const double b1X = s.pt[DL_PT_B1X], b1Y = s.pt[DL_PT_B1Y], sn0nd = s.pt[DL_PT_SN0ND];
    const bool damp = s.pt[DL_PT_DAMP] != 0.;
    const int n_ell = o.n_ell, n_mu = o.n_mu, n_kin = o.n_kin;
    for (int i = tid; i < n_kin; i += nthr) {
        double lk = o.lkin[i], kk = o.kin[i];
        double dd[DL_MAX_ELL], dt[DL_MAX_ELL], tt[DL_MAX_ELL];
#pragma unroll
        for (int l = 0; l < DL_MAX_ELL; ++l) dd[l] = dt[l] = tt[l] = 0.;
        for (int m = 0; m < n_mu; ++m) {
            double T = jac * dl_spline_eval(o, s, lk + s.pt[DL_PT_LQ + m]);
            if (damp) {
                double kap = kk / qper * s.pt[DL_PT_FAC + m];   // tgc/base.py:220
                T *= exp(-(kap * kap * s.pt[DL_PT_SD + m]) / 2.);  // full_shape.py:492-493
            }
            double fm2 = f * s.pt[DL_PT_MUP2 + m];
            double Tdt = fm2 * T, Ttt = fm2 * fm2 * T;
#pragma unroll
            for (int l = 0; l < DL_MAX_ELL; ++l) {
                if (l < n_ell) {
                    double w = o.wmu[l * n_mu + m];
                    dd[l] = fma(w, T, dd[l]);
                    dt[l] = fma(w, Tdt, dt[l]);
                    tt[l] = fma(w, Ttt, tt[l]);
                }
            }
        }
        double dd0 = 0.;
#pragma unroll
        for (int l = 0; l < DL_MAX_ELL; ++l)
            if (l == o.ell0) dd0 = dd[l];
#pragma unroll
        for (int l = 0; l < DL_MAX_ELL; ++l) {
            if (l < n_ell) {
                // full_shape.py:550
                double p = b1X * b1Y * dd[l] + (b1X + b1Y) * dt[l] + tt[l] + (l == o.ell0 ? sn0nd : 0.);
                if (o.n_ct > 0) {  // full_shape.py:633
                    double acc = 0.;
                    for (int c = 0; c < o.n_ct; ++c) acc += o.ct_matrix[((size_t)l * n_kin + i) * o.n_ct + c] * s.pt[DL_PT_CT + c];
                    p += acc * dd0;
                }
                if (o.n_sn > 0) {  // full_shape.py:634
                    double acc = 0.;
                    for (int c = 0; c < o.n_sn; ++c) acc += o.sn_matrix[((size_t)l * n_kin + i) * o.n_sn + c] * s.pt[DL_PT_SN + c];
                    p += acc;
                }
                power_row[(size_t)l * n_kin + i] = p;
                if (tables_row) {
                    tables_row[(size_t)(0 * n_ell + l) * n_kin + i] = dd[l];
                    tables_row[(size_t)(1 * n_ell + l) * n_kin + i] = dt[l];
                    tables_row[(size_t)(2 * n_ell + l) * n_kin + i] = tt[l];
                }
            }
        }
    }
}
