// dl_fullshape_grad.h -- analytic gradient of the Gaussian log-likelihood through the Kaiser full-shape theory (the value_and_grad the
// reference's HMC / NUTS samplers take from jax, desilike/samplers/hmc.py:194, samplers/nuts.py:205).
//
// logL = -1/2 |d~|^2, d~ = W~ p(theta) + b~  =>  d logL / d theta_i = Y . dp / d theta_i  with  Y = -W~^T d~  (one extra GEMM).
// The theory vector of an observable is  P_l(k) = sum_m Omega_l(m) S(log10 k + lq_m) + delta_{l, 0} sn0 / nd  (dl_fullshape.h), with
//   Omega_l(m) = jac w_l(m) (b1X + f mu'^2_m)(b1Y + f mu'^2_m),  S = not-a-knot spline of the template  y_j = P_fid(k_j) exp(dm / a tanh_j + dn ln_j),
//   x_m = 1 + mu^2 (qper^2 / qpar^2 - 1),  mu'^2_m = mu^2 (qper / qpar)^2 / x_m,  lq_m = log10(x_m) / 2 - log10(qper),  jac = 1 / (qpar qper^2).
// Its derivatives with respect to the PHYSICAL inputs (qpar, qper, f, b1X, b1Y, sn0, dm, dn) need, per (k, mu) evaluation, the spline value T, its slope T'
// (the interval's cubic differentiated), and the spline of the template's dm- (and dn-) derivative  y_j tanh_j / a  (y_j ln_j): the spline is LINEAR in its
// data, so d S / d dm is the spline of d y / d dm -- built by the same convolution, in the same LDS, after the pass that used the template's own spline.  Nothing of dp / d theta is written to memory: each
// thread contracts its wavenumbers with Y on the fly,  G_phys += Z (alpha_phys(m) T + beta_phys(m) T'),  Z = sum_l Y_l(k) w_l(m),  and the workgroup reduces the
// eight sums in a fixed order.  The chain rule to the sampled parameters (AP modes, df, tracer namespaces, several observables sharing a column) and the prior's
// gradient are applied by dl_grad_finalize_kernel (dl_kernels.hip).
// Scope: Kaiser tracers without counter terms, uniform knots (convolution path) or a fixed template, no damping (sigmapar = sigmaper = 0), no observable
// transform; everything else reports "not applicable" and the caller falls back to central differences (desilike_amd/fisher.py).
#pragma once
#include "dl_fullshape.h"

enum { DL_G_QPAR = 0, DL_G_QPER, DL_G_F, DL_G_B1X, DL_G_B1Y, DL_G_SN0, DL_G_DM, DL_G_DN, DL_NPHYS };

// per-mu record of the gradient phase: w_l (5), alpha / beta of qpar, qper, alpha of f, b1X, b1Y, gamma = jac bias
#define DL_GW 16
enum { DL_GW_W = 0, DL_GW_AQPAR = 5, DL_GW_BQPAR, DL_GW_AQPER, DL_GW_BQPER, DL_GW_AF, DL_GW_AB1X, DL_GW_AB1Y, DL_GW_G, DL_GW_LQH };

// gradient weights of mu node m (c: the node's forward quantities after parts A and B)
DL_HD void dl_fs_grad_weights(const DlObsDev& o, int m, const DlMuCarry& c, double* gw) {
    const double ln10 = 2.302585092994046;
    double* r = gw + (size_t)m * DL_GW;
    for (int l = 0; l < DL_MAX_ELL; ++l) r[DL_GW_W + l] = l < o.n_ell ? c.w[l] : 0.;
    const double mu2 = c.mu * c.mu;
    const double fm2 = c.f * c.mup2;
    const double A = c.b1X + fm2, Bb = c.b1Y + fm2, bias = A * Bb;
    const double dmup2_diq2 = mu2 * (1. - mu2) / (c.x * c.x);            // mu'^2 = mu^2 iq2 / x, x = 1 + mu^2 (iq2 - 1)
    const double dlq_diq2 = mu2 / (2. * c.x * ln10);
    const double diq2_dqpar = -2. * c.iq2 / c.qpar, diq2_dqper = 2. * c.iq2 / c.qper;
    const double dbias_dmup2 = c.f * (A + Bb);
    r[DL_GW_AQPAR] = (-c.jac / c.qpar) * bias + c.jac * dbias_dmup2 * dmup2_diq2 * diq2_dqpar;
    r[DL_GW_BQPAR] = c.jac * bias * dlq_diq2 * diq2_dqpar;
    r[DL_GW_AQPER] = (-2. * c.jac / c.qper) * bias + c.jac * dbias_dmup2 * dmup2_diq2 * diq2_dqper;
    r[DL_GW_BQPER] = c.jac * bias * (dlq_diq2 * diq2_dqper - 1. / (c.qper * ln10));
    r[DL_GW_AF] = c.jac * c.mup2 * (A + Bb);
    r[DL_GW_AB1X] = c.jac * Bb;
    r[DL_GW_AB1Y] = c.jac * A;
    r[DL_GW_G] = c.jac * bias;
    r[DL_GW_LQH] = c.lq * o.inv_hx;
}

DL_HD void dl_fs_grad_weights_pad(const DlObsDev& o, double* gw) {   // zero records up to a multiple of four nodes (one thread)
    for (int mm = o.n_mu; mm < ((o.n_mu + 3) & ~3); ++mm)
        for (int q = 0; q < DL_GW; ++q) gw[(size_t)mm * DL_GW + q] = 0.;
}

// template derivative data at the knots, IN PLACE of the template (the interval polynomials of the template are built by then): which = 0: d y / d dm = y tanh / a
// (s.y holds the template), which = 1: d y / d dn = y ln (s.y holds d y / d dm: the template is formed again)
DL_HD void dl_fs_grad_knots(int tid, int nthr, const DlObsDev& o, const double* th, const DlFsShared& s, int which) {
    const int n_t = o.n_t;
    if (which == 0) {
        for (int j = tid; j < n_t; j += nthr) s.y[j] = s.y[j] * o.sf_th[j] / o.a;
    } else {
        const double dm_a = dl_get(o.dm, th) / o.a, dn = dl_get(o.dn, th);
        for (int j = tid; j < n_t; j += nthr) s.y[j] = (o.pk_fid[j] * exp(dm_a * o.sf_th[j] + dn * o.sf_lg[j])) * o.sf_lg[j];
    }
}

// value and slope (per unit of the abscissa x = log10 k) of the interval polynomial at t = (x - x0) inv_hx
DL_HD void dl_spline_eval_t2(const DlObsDev& o, const DlFsShared& s, double t, int& j, double& u, double& v, double& dv) {
    j = (int)t;
    j = j < 0 ? 0 : j;
    if (j > o.n_t - 2) j = o.n_t - 2;
    u = t - (double)j;
    const double* c = s.coef + 2 * j;
    const double* d = c + 2 * o.n_t;
    v = fma(fma(fma(d[1], u, d[0]), u, c[1]), u, c[0]);
    dv = fma(fma(3. * d[1], u, 2. * d[0]), u, c[1]) * o.inv_hx;
}

DL_HD double dl_spline_eval_ju(const DlObsDev& o, const DlFsShared& s, int j, double u) {
    const double* c = s.coef + 2 * j;
    const double* d = c + 2 * o.n_t;
    return fma(fma(fma(d[1], u, d[0]), u, c[1]), u, c[0]);
}

// Gradient phase 3.  The workgroup holds ONE spline at a time (36 KB of LDS: four workgroups per CU, like the forward kernel; two splines side by side were 80 KB --
// one workgroup per CU, four rounds at 1024 points): pass 0 with the template's spline: everything that multiplies T and T'; pass 1 with the spline of
// d template / d dm: the dm sum; pass 2 (dn sampled) with the spline of d template / d dn.
// Y: this observable's columns of -W~^T d~, [n_ell][n_kin]; acc [DL_NPHYS]: the thread's sums (added to).
// The mu node is the OUTER loop: its record (the same for every thread: broadcast reads) is held in registers while the thread walks its wavenumbers
// (two per thread at the benchmark shape), whose abscissae and Y values are loaded once, before the loop.
#define DL_GRAD_KPT 2
template <int NL>
DL_HD void dl_fs_grad_phase3(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, const double* gw, const double* __restrict__ Y, int pass, double* acc) {
    const int n_kin = o.n_kin, n_mu4 = (o.n_mu + 3) & ~3;
    for (int i0 = tid; i0 < n_kin; i0 += DL_GRAD_KPT * nthr) {
        double t0[DL_GRAD_KPT], y[DL_GRAD_KPT][NL];
#pragma unroll
        for (int q = 0; q < DL_GRAD_KPT; ++q) {
            const int i = i0 + q * nthr;
            const bool live = i < n_kin;
            const int ii = live ? i : n_kin - 1;
            t0[q] = (o.lkin[ii] - o.x0) * o.inv_hx;
#pragma unroll
            for (int l = 0; l < NL; ++l) y[q][l] = (live && l < o.n_ell) ? Y[(size_t)l * n_kin + ii] : 0.;
            if (pass == 0 && o.ell0 >= 0 && live) acc[DL_G_SN0] += Y[(size_t)o.ell0 * n_kin + ii] / o.nd;
        }
        for (int m = 0; m < n_mu4; ++m) {
            const double* r = gw + (size_t)m * DL_GW;
            double w[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) w[l] = r[DL_GW_W + l];
            const double lqh = r[DL_GW_LQH];
            if (pass == 0) {
                const double aqpar = r[DL_GW_AQPAR], bqpar = r[DL_GW_BQPAR], aqper = r[DL_GW_AQPER], bqper = r[DL_GW_BQPER], af = r[DL_GW_AF], ab1x = r[DL_GW_AB1X], ab1y = r[DL_GW_AB1Y];
#pragma unroll
                for (int q = 0; q < DL_GRAD_KPT; ++q) {
                    double z = 0.;
#pragma unroll
                    for (int l = 0; l < NL; ++l) z = fma(y[q][l], w[l], z);      // (z = 0 for a wavenumber beyond the table: nothing is added)
                    int j; double u, T, dT;
                    dl_spline_eval_t2(o, s, t0[q] + lqh, j, u, T, dT);
                    acc[DL_G_QPAR] = fma(z, fma(aqpar, T, bqpar * dT), acc[DL_G_QPAR]);
                    acc[DL_G_QPER] = fma(z, fma(aqper, T, bqper * dT), acc[DL_G_QPER]);
                    acc[DL_G_F] = fma(z * af, T, acc[DL_G_F]);
                    acc[DL_G_B1X] = fma(z * ab1x, T, acc[DL_G_B1X]);
                    acc[DL_G_B1Y] = fma(z * ab1y, T, acc[DL_G_B1Y]);
                }
            } else {
                const double gg = r[DL_GW_G];
                double sum = 0.;
#pragma unroll
                for (int q = 0; q < DL_GRAD_KPT; ++q) {
                    double z = 0.;
#pragma unroll
                    for (int l = 0; l < NL; ++l) z = fma(y[q][l], w[l], z);
                    sum = fma(z * gg, dl_spline_eval_t(o, s, t0[q] + lqh), sum);
                }
                acc[pass == 1 ? DL_G_DM : DL_G_DN] += sum;
            }
        }
    }
}

// fixed-order reduction of the threads' sums: red [nthr][DL_NPHYS] -> out [DL_NPHYS]; thread p < DL_NPHYS sums column p in groups of 16 threads
DL_HD void dl_fs_grad_reduce(int tid, int nthr, const double* red, double* out) {
    if (tid >= DL_NPHYS) return;
    double total = 0.;
    for (int g0 = 0; g0 < nthr; g0 += 16) {
        double part = 0.;
        for (int t = g0; t < g0 + 16 && t < nthr; ++t) part += red[(size_t)t * DL_NPHYS + tid];
        total += part;
    }
    out[tid] = total;
}

// Chain rule from one observable's physical inputs to the theta columns: grad [P] += J^T g (dl_ap_qparqper, power_template.py:757, full_shape.py:88-128)
DL_HD void dl_fs_grad_chain(const DlObsDev& o, const double* th, const double* g, double* grad) {
    double qpar, qper;
    dl_ap_qparqper(o, th, qpar, qper);
    switch (o.apmode) {
        case 1: if (o.qiso.col >= 0) grad[o.qiso.col] += g[DL_G_QPAR] + g[DL_G_QPER]; break;
        case 2: if (o.qap.col >= 0) { const double qap = dl_get(o.qap, th); grad[o.qap.col] += (g[DL_G_QPAR] * (1. - o.eta) * qpar - g[DL_G_QPER] * o.eta * qper) / qap; } break;
        case 3: {
            const double qiso = dl_get(o.qiso, th), qap = dl_get(o.qap, th);
            if (o.qiso.col >= 0) grad[o.qiso.col] += (g[DL_G_QPAR] * qpar + g[DL_G_QPER] * qper) / qiso;
            if (o.qap.col >= 0) grad[o.qap.col] += (g[DL_G_QPAR] * (1. - o.eta) * qpar - g[DL_G_QPER] * o.eta * qper) / qap;
            break;
        }
        default:
            if (o.qpar.col >= 0) grad[o.qpar.col] += g[DL_G_QPAR];
            if (o.qper.col >= 0) grad[o.qper.col] += g[DL_G_QPER];
    }
    if (o.df.col >= 0) grad[o.df.col] += g[DL_G_F] * o.f_fid;
    if (o.b1X.col >= 0) grad[o.b1X.col] += g[DL_G_B1X];
    if (o.b1Y.col >= 0) grad[o.b1Y.col] += g[DL_G_B1Y];
    if (o.sn0.col >= 0) grad[o.sn0.col] += g[DL_G_SN0];
    if (o.templ == 1) {
        if (o.dm.col >= 0) grad[o.dm.col] += g[DL_G_DM];
        if (o.dn.col >= 0) grad[o.dn.col] += g[DL_G_DN];
    }
}

// can the analytic gradient be formed for this observable?
DL_HD bool dl_fs_grad_applicable(const DlObsDev& o) {
    const bool fast = o.uniform_knots && (o.toeplitz || o.fixed_spline);
    const bool damping = o.sigpar.col >= 0 || o.sigper.col >= 0 || o.sigpar.value != 0. || o.sigper.value != 0.;
    return o.theory == 0 && o.templ <= 1 && fast && !damping && o.n_ct == 0 && o.n_sn == 0 && o.n_pass == 0 && o.n_var == 0 && !o.damping_fid;   // (templates: fixed / ShapeFit)
}

// LDS doubles of the gradient workgroup: the forward layout (fast), the per-mu records, the reduction scratch (one row of sums per wavefront; the host emulation
// keeps one per thread: dl_fs_grad_reduce)
DL_HD size_t dl_fs_grad_shared_doubles(const DlObsDev& o, bool per_thread_scratch = false) {
    return dl_fs_shared_doubles_obs(o, true) + (size_t)DL_MAX_MU * DL_GW + (size_t)(per_thread_scratch ? DL_FS_THREADS : DL_FS_THREADS / 64) * DL_NPHYS + DL_NPHYS;
}
