// emulate.cpp -- TEST INFRASTRUCTURE: runs the theory kernel's phase functions (desilike_amd/csrc/dl_fullshape.h)
// and the host-side constant folding (dl_host.hpp) on the CPU, emulating one workgroup by looping tid over
// [0, 256) for each barrier-separated phase.  Lets the `not gpu` test-suite catch logic errors in the device
// arithmetic without a GPU.  It is NOT a fallback: nothing in desilike_amd/ links or loads it.
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../desilike_amd/csrc/dl_host.hpp"
#include "../../desilike_amd/csrc/dl_fullshape_grad.h"

static std::string g_err;

extern "C" {

dl_config* emu_config_new(void) { return new dl_config(); }
void emu_config_free(dl_config* cfg) { delete cfg; }
int emu_config_set_f64(dl_config* cfg, const char* key, const double* data, int64_t n) { cfg->f64[key] = std::vector<double>(data, data + n); return 0; }
int emu_config_set_i32(dl_config* cfg, const char* key, const int32_t* data, int64_t n) { cfg->i32[key] = std::vector<int32_t>(data, data + n); return 0; }
const char* emu_last_error(void) { return g_err.c_str(); }

extern "C" int emu_tns_tables(const double* k11, int n11, const double* q, int n_q, const double* mus, const double* wmus, int n_mu, const double* pk, double* tables);

// TNS one-loop theory of ONE point: the template, the 29 tables by the device's geometry / table functions run sequentially (emu_tns_tables), then what
// dl_tns_assemble_kernel does -- combination into the polynomials Q_n, second derivatives through the same not-a-knot operator the plan builds, dl_tns_eval_k
static void run_point_tns(const DlObsDev& o, const double* th, double* prow) {
    const DlObsHost& oh = *static_cast<const DlObsHost*>(o.tns_plan);
    const int n11 = (int)oh.tns_k11.size(), n_q = o.n_t, nq = o.n_ct > 0 ? 6 : 5, ldq = dl_tns_ldq(n11);
    std::vector<double> pk(n_q), tables((size_t)DL_TNS_NTAB * n11);
    const double dm_a = dl_get(o.dm, th) / o.a, dn = dl_get(o.dn, th);
    for (int j = 0; j < n_q; ++j) pk[j] = o.templ == 1 ? o.pk_fid[j] * std::exp(dm_a * o.sf_th[j] + dn * o.sf_lg[j]) : o.pk_fid[j];
    emu_tns_tables(oh.tns_k11.data(), n11, oh.tns_kt.data(), n_q, oh.tns_mu.data(), oh.tns_wmu.data(), (int)oh.tns_mu.size(), pk.data(), tables.data());
    double qpar, qper;
    dl_ap_qparqper(o, th, qpar, qper);
    const double f = o.f_fid * dl_get(o.df, th);
    std::vector<double> Q((size_t)6 * ldq, 0.), M((size_t)6 * ldq, 0.), murec(8 * DL_MAX_MU, 0.), x11(n11);
    for (int n = 0; n < nq; ++n)
        for (int i = 0; i < n11; ++i) {
            double sum = 0.;
            for (int r = 0; r < DL_TNS_NTAB; ++r) sum = std::fma(dl_tns_combine_coef(n, r, f, dl_get(o.b1X, th), dl_get(o.b2, th), dl_get(o.bs, th), dl_get(o.b3, th)), tables[(size_t)r * n11 + i], sum);
            Q[(size_t)n * ldq + i] = sum;
        }
    for (int i = 0; i < n11; ++i) x11[i] = std::log10(oh.tns_k11[i]);
    DlSplineSetup sp;
    std::string serr;
    dl_spline_setup(x11, sp, serr);
    for (int n = 0; n < nq; ++n) {
        std::vector<double> y(Q.begin() + (size_t)n * ldq, Q.begin() + (size_t)n * ldq + n11), Mv;
        dl_spline_moments_serial(y, sp, Mv);
        for (int i = 0; i < n11; ++i) M[(size_t)n * ldq + i] = Mv[i];
    }
    for (int m = 0; m < o.n_mu; ++m) dl_tns_mu_record(o, qpar, qper, m, murec.data());
    const double sc[4] = {qper, dl_get(o.sigmav, th), dl_get(o.sn0, th) / o.nd, 0.};
    std::vector<double> out((size_t)o.n_in + o.n_kin, 0.);
    const double inv_dk11 = (n11 - 1) / (oh.tns_k11[n11 - 1] - oh.tns_k11[0]);
    for (int ik = 0; ik < o.n_kin; ++ik) dl_tns_eval_k(o, oh.tns_fog, oh.tns_k11[0], inv_dk11, x11.data(), n11, ldq, nq, Q.data(), M.data(), murec.data(), sc, th, ik, out.data());
    for (int idx = 0; idx < o.n_in; ++idx) prow[idx] = out[idx];
}

static void run_point(const DlObsDev& o, const double* th, double* prow, double* trow) {
    if (o.theory == 4) { run_point_tns(o, th, prow); return; }
    if (o.theory == 3) {   // emulated theory
        std::vector<double> lds(dl_emu_shared_doubles_obs(o));
        dl_emu_point(o, th, lds.data(), prow, o.n_in + o.n_pass);
        return;
    }
    if (o.theory == 5) {   // PNG theory: the phases of dl_png_kernel, the two spline builds through the same phase functions
        std::vector<double> lds(dl_png_shared_doubles(o.n_t, o.n_in), 0.);
        const int nthr = DL_FS_THREADS;
        const bool toep = o.toeplitz;
        DlFsShared s = dl_fs_shared_carve(lds.data(), o.n_t, o.n_in, -1, toep);
        double* tabs = s.coef;
        double* murec = lds.data() + dl_fs_shared_doubles(o.n_t, o.n_in);
        double* sc = murec + 8 * DL_PNG_MAX_MU;
        auto build = [&]() {
            if (toep) {
                for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2_fir(tid, nthr, o, s);
            } else {
                for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2a(tid, nthr, o, s);
                for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2b_dot(tid, nthr, o, s);
                for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2b(tid, nthr, o, s);
                for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2c_dot(tid, nthr, o, s);
                for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2c(tid, nthr, o, s);
            }
        };
        for (int tid = 0; tid < nthr; ++tid) dl_png_setup(tid, nthr, o, th, murec, sc);
        for (int tid = 0; tid < nthr; ++tid) dl_png_knots(tid, nthr, o, th, s, true);
        build();
        for (int tid = 0; tid < nthr; ++tid) dl_png_keep_spline(tid, nthr, o, s, toep, tabs, tabs + o.n_t);
        for (int tid = 0; tid < nthr; ++tid) dl_png_knots(tid, nthr, o, th, s, false);
        build();
        for (int tid = 0; tid < nthr; ++tid) dl_png_keep_spline(tid, nthr, o, s, toep, tabs + 2 * o.n_t, tabs + 3 * o.n_t);
        std::vector<double> out(o.n_in);
        for (int tid = 0; tid < nthr; ++tid) dl_png_eval(tid, nthr, o, tabs, murec, sc, out.data());
        for (int idx = 0; idx < o.n_in; ++idx) prow[idx] = out[idx];
        return;
    }
    if (o.theory == 2) {   // BAO wiggle model
        std::vector<double> lds(dl_bao_shared_doubles(o.n_in));
        const int nthr = DL_FS_THREADS;
        for (int tid = 0; tid < nthr; ++tid) dl_bao_phaseA(tid, nthr, o, th, lds.data());
        for (int tid = 0; tid < nthr; ++tid) dl_bao_phaseB(tid, nthr, o, lds.data());
        for (int tid = 0; tid < nthr; ++tid) dl_store_with_pass(tid, nthr, o, th, lds.data() + DL_BAO_PT, prow);
        return;
    }
    std::vector<double> lds(dl_fs_shared_doubles(o.n_t, o.n_in));
    const bool toep = o.toeplitz && !o.fixed_spline;
    DlFsShared s = dl_fs_shared_carve(lds.data(), o.n_t, o.n_in, -1, toep);
    const int nthr = DL_FS_THREADS;
    const bool fast = !(trow || !o.uniform_knots || !(o.toeplitz || o.fixed_spline));   // same dispatch as dl_launch_fullshape
    if (fast) {
        // mirrors dl_fullshape_kernel<FAST = true>: threads 0 .. KT-1 build the spline, the last wave runs the per-mu chain, one part per phase
        const int KT = DL_FS_KT;
        std::vector<DlMuCarry> carry(nthr);
        auto mu_lane = [&](int tid) { return tid >= KT && tid - KT < o.n_mu; };
        for (int tid = 0; tid < nthr; ++tid) {
            if (tid >= KT) dl_fs_mu_partA(o, th, mu_lane(tid) ? tid - KT : 0, carry[tid]);
            else dl_fs_knots(tid, KT, o, th, s);
            if (o.fixed_spline && tid >= KT) {
                dl_fs_mu_partB(carry[tid]);
                if (mu_lane(tid)) dl_fs_mu_partC(o, s, tid - KT, carry[tid]);
                if (tid == nthr - 1) dl_fs_scalars(o, th, s, carry[tid]);
            }
        }
        if (toep) {
            for (int tid = 0; tid < nthr; ++tid) {
                if (tid >= KT) dl_fs_mu_partB(carry[tid]);
                else dl_fs_phase2_fir(tid, KT, o, s);
            }
            for (int tid = 0; tid < nthr; ++tid) {
                if (tid >= KT) {
                    if (mu_lane(tid)) dl_fs_mu_partC(o, s, tid - KT, carry[tid]);
                    if (tid == nthr - 1) dl_fs_scalars(o, th, s, carry[tid]);
                } else {
                    double dlt_pref[DL_TOEP_PREF];
                    for (int it = 0; it < DL_TOEP_PREF; ++it) dlt_pref[it] = (tid + it * KT < o.n_t - 1) ? o.dlt[tid + it * KT] : 0.;
                    dl_fs_phase2d_toep(tid, KT, o, s, dlt_pref);
                }
            }
        }
    } else {
        for (int tid = 0; tid < nthr; ++tid) dl_fs_phase01(tid, nthr, o, th, s);
        if (toep) {
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2_fir(tid, nthr, o, s);
            for (int tid = 0; tid < nthr; ++tid) {
                double dlt_pref[DL_TOEP_PREF];
                for (int it = 0; it < DL_TOEP_PREF; ++it) dlt_pref[it] = (tid + it * nthr < o.n_t - 1) ? o.dlt[tid + it * nthr] : 0.;
                dl_fs_phase2d_toep(tid, nthr, o, s, dlt_pref);
            }
        } else if (!o.fixed_spline) {
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2a(tid, nthr, o, s);
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2b_dot(tid, nthr, o, s);
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2b(tid, nthr, o, s);
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2c_dot(tid, nthr, o, s);
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2c(tid, nthr, o, s);
            for (int tid = 0; tid < nthr; ++tid) dl_fs_phase2d(tid, nthr, o, s);
        }
    }
    for (int tid = 0; tid < nthr; ++tid) {
        bool nl3 = o.n_ell <= 3, eft = o.n_ct > 0 || o.n_sn > 0;   // same dispatch as dl_launch_fullshape
        if (trow || !o.uniform_knots || !(o.toeplitz || o.fixed_spline)) dl_fs_phase3<false, 5, true>(tid, nthr, o, s, trow);
        else if (nl3 && !eft) dl_fs_phase3_pair<3, false>(tid, nthr, o, s);
        else if (nl3) dl_fs_phase3<true, 3, true>(tid, nthr, o, s, trow);
        else if (!eft) dl_fs_phase3_pair<5, false>(tid, nthr, o, s);
        else dl_fs_phase3<true, 5, true>(tid, nthr, o, s, trow);
    }
    for (int tid = 0; tid < nthr; ++tid) dl_fs_phase4(tid, nthr, o, s, th, prow, o.n_in);
}

// gradient workgroup (dl_fullshape_grad_kernel, dl_kernels.hip) of one (point, observable): Y = this observable's columns of -W~^T d~; gphys [DL_NPHYS]
static bool run_point_grad(const DlObsDev& o, const double* th, const double* Y, double* gphys) {
    if (!dl_fs_grad_applicable(o)) return false;
    std::vector<double> lds(dl_fs_grad_shared_doubles(o, true) + 64, 0.);
    const bool toep = o.toeplitz && !o.fixed_spline;
    DlFsShared s = dl_fs_shared_carve(lds.data(), o.n_t, o.n_in, dl_fs_n_dd0(o), toep);
    double* gw = s.pt + DL_PT_SIZE_FAST;
    double* red = gw + (size_t)DL_MAX_MU * DL_GW;
    double* out = red + (size_t)DL_FS_THREADS * DL_NPHYS;
    const int nthr = DL_FS_THREADS, KT = DL_FS_KT;
    std::vector<DlMuCarry> carry(nthr);
    auto mu_lane = [&](int tid) { return tid >= KT && tid - KT < o.n_mu; };
    for (int tid = 0; tid < nthr; ++tid) {
        if (tid >= KT) {
            dl_fs_mu_partA(o, th, mu_lane(tid) ? tid - KT : 0, carry[tid]);
            dl_fs_mu_partB(carry[tid]);
            if (mu_lane(tid)) { dl_fs_mu_partC(o, s, tid - KT, carry[tid], false); dl_fs_grad_weights(o, tid - KT, carry[tid], gw); }
            if (tid == nthr - 1) { dl_fs_scalars(o, th, s, carry[tid], false); dl_fs_grad_weights_pad(o, gw); }
        } else dl_fs_knots(tid, KT, o, th, s);
    }
    auto build = [&]() {
        for (int tid = 0; tid < KT; ++tid) dl_fs_phase2_fir(tid, KT, o, s);
        for (int tid = 0; tid < KT; ++tid) {
            double dlt_pref[DL_TOEP_PREF];
            for (int it = 0; it < DL_TOEP_PREF; ++it) dlt_pref[it] = (tid + it * KT < o.n_t - 1) ? o.dlt[tid + it * KT] : 0.;
            dl_fs_phase2d_toep(tid, KT, o, s, dlt_pref);
        }
    };
    auto contract = [&](int pass) {
        for (int tid = 0; tid < nthr; ++tid) {
            if (o.n_ell <= 3) dl_fs_grad_phase3<3>(tid, nthr, o, s, gw, Y, pass, red + (size_t)tid * DL_NPHYS);
            else dl_fs_grad_phase3<5>(tid, nthr, o, s, gw, Y, pass, red + (size_t)tid * DL_NPHYS);
        }
    };
    if (toep) build();
    std::fill(red, red + (size_t)nthr * DL_NPHYS, 0.);
    contract(0);
    if (toep && o.templ == 1) {
        for (int which = 0; which < 2; ++which) {
            if (which == 0 ? o.dm.col < 0 : o.dn.col < 0) continue;
            for (int tid = 0; tid < KT; ++tid) dl_fs_grad_knots(tid, KT, o, th, s, which);
            build();
            contract(1 + which);
        }
    }
    for (int tid = 0; tid < nthr; ++tid) dl_fs_grad_reduce(tid, nthr, red, out);
    std::copy(out, out + DL_NPHYS, gphys);
    return true;
}

// power [B, n_in], tables [B, 3, n_in] (may be null)
int emu_eval_theory(const dl_config* cfg, const double* theta, int64_t B, int iobs, double* power, double* tables) {
    int P = cfg->i("n_params", -1);
    DlArena arena;
    DlObsHost oh;
    if (!dl_build_obs(*cfg, iobs, P, oh, arena, g_err)) return 1;
    oh.rebase(arena.data.data());
    oh.dev.col_offset = 0;
    std::vector<double> rowbuf(oh.n_cols());
    for (int64_t b = 0; b < B; ++b) {
        run_point(oh.dev, theta + b * P, rowbuf.data(), tables ? tables + b * 3 * oh.dev.n_in : nullptr);
        std::copy(rowbuf.begin(), rowbuf.begin() + oh.dev.n_in, power + b * oh.dev.n_in);
    }
    return 0;
}

// whole pipeline with the same folding as dl_create: dtilde = (L^T W) power + L^T (bias - data); loglike = -1/2 |dtilde|^2
int emu_eval_batch(const dl_config* cfg, const double* theta, int64_t B, double* loglike, double* flattheory) {
    int P = cfg->i("n_params", -1), nobs = cfg->i("n_obs", -1);
    DlArena arena;
    std::vector<DlObsHost> obs(nobs);
    int n = 0, K = 0;
    std::vector<int> row0, col0;
    for (int i = 0; i < nobs; ++i) {
        if (!dl_build_obs(*cfg, i, P, obs[i], arena, g_err)) return 1;
        row0.push_back(n); col0.push_back(K);
        n += obs[i].n_out; K += obs[i].n_cols();
    }
    for (int i = 0; i < nobs; ++i) { obs[i].rebase(arena.data.data()); obs[i].dev.col_offset = col0[i]; }
    const auto& prec = cfg->F("precision");
    std::vector<double> L((size_t)n * n, 0.);
    if ((int64_t)prec.size() == (int64_t)n * n) { L = prec; if (!dl_cholesky(L, n)) { g_err = "not positive definite"; return 1; } }
    else for (int i = 0; i < n; ++i) L[(size_t)i * n + i] = std::sqrt(prec[i]);
    std::vector<double> power(K), flat(n);
    for (int64_t b = 0; b < B; ++b) {
        for (int i = 0; i < nobs; ++i) run_point(obs[i].dev, theta + b * P, power.data() + col0[i], nullptr);
        for (int i = 0; i < nobs; ++i)
            for (int r = 0; r < obs[i].n_out; ++r) {
                double sum = 0.;
                for (int k = 0; k < obs[i].n_cols(); ++k) sum += obs[i].weff[(size_t)r * obs[i].n_cols() + k] * power[col0[i] + k];
                flat[row0[i] + r] = sum + obs[i].bias[r];
            }
        if (flattheory) std::copy(flat.begin(), flat.end(), flattheory + b * n);
        double chi2 = 0.;
        for (int i = 0; i < n; ++i) {
            double d = 0.;
            int oi = 0;
            for (int j = i; j < n; ++j) {
                while (oi + 1 < nobs && j >= row0[oi + 1]) ++oi;
                d += L[(size_t)j * n + i] * (flat[j] - obs[oi].flatdata[j - row0[oi]]);
            }
            chi2 += d * d;
        }
        loglike[b] = -0.5 * chi2;
    }
    return 0;
}

// analytic gradient of the log-likelihood with the device's gradient phase functions (dl_fullshape_grad.h): loglike [B], grad [B, P]; returns 2 if not applicable
int emu_eval_grad(const dl_config* cfg, const double* theta, int64_t B, double* loglike, double* grad) {
    int P = cfg->i("n_params", -1), nobs = cfg->i("n_obs", -1);
    DlArena arena;
    std::vector<DlObsHost> obs(nobs);
    int n = 0, K = 0;
    std::vector<int> row0, col0;
    for (int i = 0; i < nobs; ++i) {
        if (!dl_build_obs(*cfg, i, P, obs[i], arena, g_err)) return 1;
        row0.push_back(n); col0.push_back(K);
        n += obs[i].n_out; K += obs[i].n_cols();
    }
    for (int i = 0; i < nobs; ++i) { obs[i].rebase(arena.data.data()); obs[i].dev.col_offset = col0[i]; }
    const auto& prec = cfg->F("precision");
    std::vector<double> L((size_t)n * n, 0.);
    if ((int64_t)prec.size() == (int64_t)n * n) { L = prec; if (!dl_cholesky(L, n)) { g_err = "not positive definite"; return 1; } }
    else for (int i = 0; i < n; ++i) L[(size_t)i * n + i] = std::sqrt(prec[i]);
    std::vector<double> power(K), flat(n), d(n), v(n), Y(K);
    for (int64_t b = 0; b < B; ++b) {
        for (int i = 0; i < nobs; ++i) run_point(obs[i].dev, theta + b * P, power.data() + col0[i], nullptr);
        for (int i = 0; i < nobs; ++i)
            for (int r = 0; r < obs[i].n_out; ++r) {
                double sum = 0.;
                for (int k = 0; k < obs[i].n_cols(); ++k) sum += obs[i].weff[(size_t)r * obs[i].n_cols() + k] * power[col0[i] + k];
                flat[row0[i] + r] = sum + obs[i].bias[r] - obs[i].flatdata[r];
            }
        double chi2 = 0.;
        for (int i = 0; i < n; ++i) { double t = 0.; for (int j = i; j < n; ++j) t += L[(size_t)j * n + i] * flat[j]; d[i] = t; chi2 += t * t; }     // d~ = L^T (flat - data)
        loglike[b] = -0.5 * chi2;
        for (int j = 0; j < n; ++j) { double t = 0.; for (int i = 0; i <= j; ++i) t += L[(size_t)j * n + i] * d[i]; v[j] = t; }                        // L d~ = precision (flat - data)
        for (int i = 0; i < nobs; ++i)
            for (int k = 0; k < obs[i].n_cols(); ++k) {
                double t = 0.;
                for (int r = 0; r < obs[i].n_out; ++r) t += obs[i].weff[(size_t)r * obs[i].n_cols() + k] * v[row0[i] + r];
                Y[col0[i] + k] = -t;
            }
        double* g = grad + b * P;
        std::fill(g, g + P, 0.);
        for (int i = 0; i < nobs; ++i) {
            double gphys[DL_NPHYS];
            if (!run_point_grad(obs[i].dev, theta + b * P, Y.data() + col0[i], gphys)) return 2;
            dl_fs_grad_chain(obs[i].dev, theta + b * P, gphys, g);
        }
    }
    return 0;
}

// TNS one-loop theory (csrc/dl_tns.h): the arithmetic of the geometry kernels, of the loop GEMM (sums over the pairs (mu, q) in the kernels' pair order) and of
// dl_tns_table_entry, run sequentially: tables [29][n11] of ONE template pk [n_q].  Returns the number of (k, mu, q) whose |k - q| falls outside the template.
int emu_tns_tables(const double* k11, int n11, const double* q, int n_q, const double* mus, const double* wmus, int n_mu, const double* pk, double* tables) {
    const double pi = 3.14159265358979323846;
    std::vector<double> jq(n_q);
    for (int j = 0; j < n_q; ++j) {
        const double wq = (j == 0 ? q[1] - q[0] : j == n_q - 1 ? q[n_q - 1] - q[n_q - 2] : q[j + 1] - q[j - 1]) / 2.;
        jq[j] = q[j] * q[j] * wq / (4. * pi * pi);
    }
    double qq = 0., sumw = 0.;
    for (int j = 0; j < n_q; ++j) qq = std::fma(jq[j] * pk[j], pk[j], qq);
    for (int m = 0; m < n_mu; ++m) sumw += wmus[m];
    int outside = 0;
    for (int ik = 0; ik < n11; ++ik) {
        const double k = k11[ik];
        double S[DL_TNS_NCOL], Lv[DL_TNS_NLIN];
        for (int i = 0; i < DL_TNS_NCOL; ++i) S[i] = 0.;
        for (int i = 0; i < DL_TNS_NLIN; ++i) Lv[i] = 0.;
        int jk; double wk0, wk1;
        dl_tns_interp_weights(q, n_q, k, jk, wk0, wk1);
        Lv[DL_TL_PK] = wk0 * pk[jk] + wk1 * pk[jk + 1];
        for (int im = 0; im < n_mu; ++im)
            for (int iq = 0; iq < n_q; ++iq) {
                DlTnsGeom g;
                dl_tns_geometry(k, q[iq], jq[iq], mus[im], wmus[im], g);
                int j; double w0, w1;
                dl_tns_interp_weights(q, n_q, g.r, j, w0, w1);
                if (w0 == 0. && w1 == 0.) ++outside;
                const double plin = std::fma(w0, pk[j], w1 * pk[j + 1]);
                const double G = pk[iq] * plin;
                for (int i = 0; i < 27; ++i) S[i] = std::fma(G, g.c[i], S[i]);
                Lv[DL_TL_SIG3] = std::fma(g.sig3, pk[iq], Lv[DL_TL_SIG3]);
                const int idx[4] = {0, 1, 2, 4};
                for (int u = 0; u < 4; ++u) Lv[DL_TL_EA0 + u] = std::fma(g.ca[idx[u]], plin, Lv[DL_TL_EA0 + u]);   // (the device folds these over the interpolation first: same sum, other order)
            }
        for (int iq = 0; iq < n_q; ++iq) {
            double ff, gg, ka[4];
            dl_tns_kernels13(q[iq] / k, ff, gg);
            dl_tns_kernels_a(q[iq] / k, ka);
            Lv[DL_TL_13D] = std::fma(2. * jq[iq] * ff, pk[iq], Lv[DL_TL_13D]);
            Lv[DL_TL_13T] = std::fma(2. * jq[iq] * gg, pk[iq], Lv[DL_TL_13T]);
            for (int u = 0; u < 4; ++u) Lv[DL_TL_KA0 + u] = std::fma(jq[iq] * ka[u], pk[iq], Lv[DL_TL_KA0 + u]);
        }
        for (int r = 0; r < DL_TNS_NTAB; ++r) tables[(size_t)r * n11 + ik] = dl_tns_table_entry(r, S, Lv, qq, sumw);
    }
    return outside;
}

// coefficients of the 29 tables (+ pk11) in the five mu'^2n polynomials of P(k, mu) and the counter-term table: cvec [6][32] (dl_tns_combine_coef)
void emu_tns_combine(double f, double b1, double b2, double bs, double b3, double* cvec) {
    for (int n = 0; n < 6; ++n)
        for (int r = 0; r < 32; ++r) cvec[n * 32 + r] = dl_tns_combine_coef(n, r, f, b1, b2, bs, b3);
}

}  // extern "C"
