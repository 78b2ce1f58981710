// dl_host.hpp -- host-side (init-time) preparation of the constants the kernels consume.
// Plain C++ (no HIP): shared by the C-ABI library (dl_api.hip) and by the CPU emulation harness of
// the `not gpu` tests.  Everything here runs once per dl_create, never per evaluation.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "dl_fullshape.h"
#include "dl_tns.h"

struct dl_config {
    std::map<std::string, std::vector<double>> f64;
    std::map<std::string, std::vector<int32_t>> i32;

    bool has_f64(const std::string& k) const { return f64.count(k) > 0; }
    bool has_i32(const std::string& k) const { return i32.count(k) > 0; }
    const std::vector<double>& F(const std::string& k) const {
        static const std::vector<double> empty;
        auto it = f64.find(k);
        return it == f64.end() ? empty : it->second;
    }
    const std::vector<int32_t>& I(const std::string& k) const {
        static const std::vector<int32_t> empty;
        auto it = i32.find(k);
        return it == i32.end() ? empty : it->second;
    }
    double f(const std::string& k, double def) const { auto& v = F(k); return v.empty() ? def : v[0]; }
    int32_t i(const std::string& k, int32_t def) const { auto& v = I(k); return v.empty() ? def : v[0]; }
};

// Arena of doubles: constants are laid out contiguously on the host, uploaded with one copy, and the
// DlObsDev pointers are rebased onto the device (or left on the host arena for the CPU emulation).
struct DlArena {
    std::vector<double> data;
    size_t push(const double* src, size_t n) {
        size_t off = data.size();
        // keep every array 16-byte aligned for vector loads
        data.insert(data.end(), src, src + n);
        if (data.size() % 2) data.push_back(0.);
        return off;
    }
    size_t push(const std::vector<double>& v) { return push(v.data(), v.size()); }
};

struct DlTnsPlan;
struct DlObsHost {
    DlObsDev dev;   // pointers are OFFSETS into arena (in doubles) until rebase()
    DlTnsPlan* tns = nullptr;   // TNS one-loop theory: geometry tables + workspace (dl_tns.hip), owned by the observable
    std::vector<double> tns_k11, tns_mu, tns_wmu, tns_kt;   // ... its table wavenumbers, loop cosines and template wavenumbers (host copies: the CPU emulation works from them)
    int tns_fog = 0;
    int n_out = 0;  // data size of this observable
    // analytic marginalisation: index of the solved parameter fed by each linear input of this observable (-1: not solved)
    int marg_sn0 = -1;
    int marg_sn[DL_MAX_EFT];
    int marg_ct[DL_MAX_EFT][2];
    std::vector<double> weff;     // [n_out, n_in] effective window (matrix / identity / row selection)
    std::vector<double> bias;     // [n_out]: W . (sn_in (x) 1) + offset[mask] - sn_out        (window.py:459-473)
    std::vector<double> flatdata; // [n_out]
    size_t off_kin, off_lkin, off_mu, off_wmu, off_xt, off_pk, off_th, off_lg, off_ih, off_dlt, off_A, off_nC, off_inv, off_gf, off_gb, off_coef, off_ct, off_sn;
    size_t off_cw, off_cn, off_pknowk, off_ml, off_pass, off_png = 0, off_band = 0;
    size_t off_eng[3][6];   // xlo, xinv, weights, center, powers, coef of each emulator engine
    size_t off_stk[3] = {0, 0, 0};   // group table, amplitude table, fragment-ordered weights of the stacked table engine
    int marg_vp[DL_N_VPARS];
    int marg_pass[DL_MAX_PASS];
    int n_cols() const { return dev.n_in + dev.n_pass; }   // columns of this observable in the theory vector / window matrix

    void rebase(const double* base) {
        dev.kin = base + off_kin; dev.lkin = base + off_lkin; dev.mu = base + off_mu; dev.wmu = base + off_wmu;
        dev.x_t = base + off_xt; dev.pk_fid = base + off_pk; dev.sf_th = base + off_th; dev.sf_lg = base + off_lg;
        dev.ih = base + off_ih; dev.dlt = base + off_dlt; dev.sp_A = base + off_A; dev.sp_nC = base + off_nC; dev.sp_inv = base + off_inv;
        dev.sp_gf = base + off_gf; dev.sp_gb = base + off_gb; dev.coef_fixed = base + off_coef;
        dev.ct_matrix = base + off_ct; dev.sn_matrix = base + off_sn;
        dev.coef_w = base + off_cw; dev.coef_n = base + off_cn; dev.pknow_k = base + off_pknowk; dev.ml_tab = base + off_ml; dev.pass_tab = base + off_pass; dev.png_alpha = base + off_png; dev.band_tab = base + off_band;
        for (int e = 0; e < 3; ++e) {
            dev.eng[e].xlo = base + off_eng[e][0]; dev.eng[e].xinv = base + off_eng[e][1]; dev.eng[e].weights = base + off_eng[e][2];
            dev.eng[e].center = base + off_eng[e][3]; dev.eng[e].powers = base + off_eng[e][4]; dev.eng[e].coef = base + off_eng[e][5];
        }
        dev.stk.table = base + off_stk[0]; dev.stk.scale = base + off_stk[1]; dev.stk.wfrag = base + off_stk[2];
    }
};

// Not-a-knot cubic spline in moment form on knots x[n]: reduced tridiagonal system for M[1..n-2]
// (the two not-a-knot conditions substituted into the first / last interior rows), factorised once.
// Per evaluation only the right-hand side changes (dl_fs_phase2a-c).
struct DlSplineSetup {
    std::vector<double> ih, A, nC, inv;
    double end0a, end0b, end1a, end1b;
    int warm;  // warm-up length after which a truncated sweep is exact to < 1e-19 relative (fp64 eps = 1.1e-16)
};

inline bool dl_spline_setup(const std::vector<double>& x, DlSplineSetup& s, std::string& err) {
    int n = (int)x.size();
    if (n < 5) { err = "spline needs at least 5 knots"; return false; }
    std::vector<double> h(n - 1);
    s.ih.resize(n - 1);
    for (int j = 0; j < n - 1; ++j) {
        h[j] = x[j + 1] - x[j];
        if (!(h[j] > 0.)) { err = "template knots must be strictly increasing"; return false; }
        s.ih[j] = 1. / h[j];
    }
    int m = n - 2;
    std::vector<double> lo(m), di(m), up(m);
    for (int i = 0; i < m; ++i) {  // reduced row i <-> original interior row i + 1
        lo[i] = h[i];
        di[i] = 2. * (h[i] + h[i + 1]);
        up[i] = h[i + 1];
    }
    // M[0] = (1 + h0/h1) M[1] - (h0/h1) M[2]
    s.end0a = 1. + h[0] / h[1];
    s.end0b = -h[0] / h[1];
    di[0] += h[0] * s.end0a;
    up[0] += h[0] * s.end0b;
    lo[0] = 0.;
    // M[n-1] = (1 + h[n-2]/h[n-3]) M[n-2] - (h[n-2]/h[n-3]) M[n-3]
    s.end1a = 1. + h[n - 2] / h[n - 3];
    s.end1b = -h[n - 2] / h[n - 3];
    di[m - 1] += h[n - 2] * s.end1a;
    lo[m - 1] += h[n - 2] * s.end1b;
    up[m - 1] = 0.;
    s.A.resize(m); s.nC.resize(m); s.inv.resize(m);
    double cprev = 0.;
    double amax = 0.;
    for (int i = 0; i < m; ++i) {
        double den = di[i] - lo[i] * cprev;
        if (!(std::fabs(den) > 0.)) { err = "singular spline system"; return false; }
        s.inv[i] = 1. / den;
        s.A[i] = -lo[i] * s.inv[i];
        cprev = up[i] * s.inv[i];
        s.nC[i] = -cprev;
        amax = std::fmax(amax, std::fmax(std::fabs(s.A[i]), std::fabs(cprev)));
    }
    if (amax < 0.9 && amax > 0.) s.warm = (int)std::ceil(std::log(1e-19) / std::log(amax)) + 1;
    else s.warm = m;  // no exploitable decay: every segment sweeps from the boundary (serial)
    if (s.warm > m) s.warm = m;
    return true;
}

inline void dl_spline_moments_serial(const std::vector<double>& y, const DlSplineSetup& s, std::vector<double>& M) {
    int n = (int)y.size(), m = n - 2;
    std::vector<double> z(m);
    M.assign(n, 0.);
    double zz = 0.;
    for (int i = 0; i < m; ++i) {
        double r = 6. * ((y[i + 2] - y[i + 1]) * s.ih[i + 1] - (y[i + 1] - y[i]) * s.ih[i]);
        zz = std::fma(s.A[i], zz, r * s.inv[i]);
        z[i] = zz;
    }
    double uu = 0.;
    for (int i = m - 1; i >= 0; --i) {
        uu = std::fma(s.nC[i], uu, z[i]);
        M[i + 1] = uu;
    }
    M[0] = s.end0a * M[1] + s.end0b * M[2];
    M[n - 1] = s.end1a * M[n - 2] + s.end1b * M[n - 3];
}

inline DlInput dl_input_from(const dl_config& cfg, const std::string& key, double def) {
    DlInput in;
    in.col = -1; in.pad = 0; in.value = def;
    const auto& v = cfg.F(key);
    if (v.size() >= 2) { in.col = (int32_t)std::lround(v[0]); in.value = v[1]; }
    return in;
}

// Cholesky factor of a symmetric positive-definite matrix: P = L L^T (lower), in place (upper part zeroed)
inline bool dl_cholesky(std::vector<double>& P, int n) {
    for (int j = 0; j < n; ++j) {
        double d = P[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= P[(size_t)j * n + k] * P[(size_t)j * n + k];
        if (!(d > 0.)) return false;
        d = std::sqrt(d);
        P[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double sum = P[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) sum -= P[(size_t)i * n + k] * P[(size_t)j * n + k];
            P[(size_t)i * n + j] = sum / d;
        }
        for (int i = 0; i < j; ++i) P[(size_t)i * n + j] = 0.;
    }
    return true;
}

// window part shared by every theory kind: effective matrix (matrix / identity / row selection) and additive bias (window.py:445-473)
inline bool dl_build_window(const dl_config& cfg, const std::string& p, DlObsHost& oh, std::string& err) {
    DlObsDev& d = oh.dev;
    const auto& wm = cfg.F(p + "wmatrix");
    const auto& kmask = cfg.I(p + "kmask");
    const auto& offset = cfg.F(p + "offset");
    const auto& snin_w = cfg.F(p + "shotnoise_in");
    const auto& snout_w = cfg.F(p + "shotnoise_out");
    const auto& flatdata = cfg.F(p + "flatdata");
    const int n_cols = d.n_in + d.n_pass;
    if (wm.empty() && d.n_pass > 0) { err = p + "pass-through parameters need an explicit wmatrix with n_in + n_pass columns"; return false; }
    int n_rows = wm.empty() ? d.n_in : (int)(wm.size() / n_cols);
    if (!wm.empty() && (size_t)n_rows * n_cols != wm.size()) { err = p + "wmatrix size is not a multiple of the theory vector size"; return false; }
    oh.n_out = kmask.empty() ? n_rows : (int)kmask.size();
    if ((int)flatdata.size() != oh.n_out) { err = p + "flatdata size does not match the window output size"; return false; }
    if (!offset.empty() && (int)offset.size() != n_rows) { err = p + "offset size mismatch"; return false; }
    if (!snout_w.empty() && (int)snout_w.size() != oh.n_out) { err = p + "shotnoise_out size mismatch"; return false; }
    if (!snin_w.empty() && (int)snin_w.size() != d.n_ell) { err = p + "shotnoise_in size mismatch"; return false; }
    oh.weff.assign((size_t)oh.n_out * n_cols, 0.);
    oh.bias.assign(oh.n_out, 0.);
    oh.flatdata = flatdata;
    for (int r = 0; r < oh.n_out; ++r) {
        int src = kmask.empty() ? r : kmask[r];
        if (src < 0 || src >= n_rows) { err = p + "kmask entry out of range"; return false; }
        double* row = &oh.weff[(size_t)r * n_cols];
        if (wm.empty()) row[src] = 1.;
        else std::memcpy(row, &wm[(size_t)src * n_cols], sizeof(double) * n_cols);
        double b = 0.;
        if (!snin_w.empty())
            for (int l = 0; l < d.n_ell; ++l) {
                if (snin_w[l] == 0.) continue;
                double sum = 0.;
                for (int i = 0; i < d.n_kin; ++i) sum += row[(size_t)l * d.n_kin + i];
                b += sum * snin_w[l];
            }
        if (!offset.empty()) b += offset[src];
        if (!snout_w.empty()) b -= snout_w[r];
        oh.bias[r] = b;
    }
    return true;
}

// Emulated theory (kind 3): emulator engines (MLP / Taylor) + bias monomials; the window matrix has n_basis * n_mono (+ n_pass) columns.
inline bool dl_build_emulated_obs(const dl_config& cfg, const std::string& p, int n_params, DlObsHost& oh, DlArena& arena, std::string& err) {
    DlObsDev& d = oh.dev;
    d.transform = cfg.i(p + "transform", 0);
    const auto& xin = cfg.F(p + "in.x");
    d.n_x = (int)(xin.size() / 2);
    if (d.n_x < 1 || d.n_x > DL_MAX_X) { err = p + "in.x: between 1 and 16 emulator inputs supported"; return false; }
    for (int c = 0; c < d.n_x; ++c) {
        d.x_in[c].col = (int32_t)std::lround(xin[2 * c]); d.x_in[c].pad = 0; d.x_in[c].value = xin[2 * c + 1];
        if (d.x_in[c].col >= n_params) { err = p + "in.x: theta column out of range"; return false; }
    }
    d.mono_mode = cfg.i(p + "mono_mode", 0);
    d.n_mono = d.mono_mode == 0 ? 1 : DL_N_MONO;
    const auto& vpin = cfg.F(p + "in.vp");
    const auto& mvp = cfg.I(p + "marg.vp");
    const double vdef[DL_N_VPARS] = {1., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
    for (int c = 0; c < DL_N_VPARS; ++c) {
        d.vp_in[c].col = -1; d.vp_in[c].pad = 0; d.vp_in[c].value = vdef[c];
        if (2 * c + 1 < (int)vpin.size()) { d.vp_in[c].col = (int32_t)std::lround(vpin[2 * c]); d.vp_in[c].value = vpin[2 * c + 1]; }
        if (d.vp_in[c].col >= n_params) { err = p + "in.vp: theta column out of range"; return false; }
        oh.marg_vp[c] = (c < (int)mvp.size()) ? mvp[c] : -1;
        d.vp_slot[c] = -1;
        if (oh.marg_vp[c] >= 0 && c < 4) { err = p + "only alpha* and sn* can be solved analytically (full_shape.py:1226)"; return false; }
    }
    const auto& vconst = cfg.F(p + "vconst");
    d.snd = vconst.size() > 0 ? vconst[0] : 1.; d.fsat = vconst.size() > 1 ? vconst[1] : 1.; d.sigv = vconst.size() > 2 ? vconst[2] : 1.;
    d.nd = vconst.size() > 3 ? vconst[3] : 1e-4;
    for (int e = 0; e < 3; ++e) {
        std::string q = p + "emu" + std::to_string(e) + ".";
        DlObsDev::Engine& en = d.eng[e];
        std::memset(&en, 0, sizeof(en));
        en.type = cfg.i(q + "type", -1);
        en.cst = cfg.f(q + "const", e == 1 ? 1. : 0.);
        std::vector<double> xlo(1, 0.), xinv(1, 0.), weights(1, 0.), center(1, 0.), powers(1, 0.), coef(1, 0.);
        if (e == 0 && en.type < 0) { err = p + "emu0 (the table engine) is required"; return false; }
        if (en.type == 0) {
            const auto& xl = cfg.F(q + "xlimits");
            const auto& widths = cfg.I(q + "widths");
            const auto& w = cfg.F(q + "weights");
            const auto& yl = cfg.F(q + "ylimits");
            en.act = cfg.i(q + "act", 0);
            en.n_layers = (int)widths.size() - 1;
            if ((int)xl.size() != 2 * d.n_x || en.n_layers < 1 || en.n_layers > DL_MAX_LAYERS || widths[0] != d.n_x) { err = q + "inconsistent MLP description"; return false; }
            size_t expect = 0;
            for (int l = 0; l <= en.n_layers; ++l) {
                en.widths[l] = widths[l];
                if (widths[l] < 1 || widths[l] > DL_MAX_WIDTH) { err = q + "layer widths must be in [1, 256]"; return false; }
                if (l) expect += (size_t)widths[l - 1] * widths[l] + widths[l];
            }
            if (w.size() != expect) { err = q + "weights size does not match the layer widths"; return false; }
            xlo.assign(d.n_x, 0.); xinv.assign(d.n_x, 0.);
            for (int c = 0; c < d.n_x; ++c) { xlo[c] = xl[2 * c]; xinv[c] = 1. / (xl[2 * c + 1] - xl[2 * c]); }
            weights = w;
            if (e != 0) {
                if (widths[en.n_layers] != 1 || yl.size() != 2) { err = q + "scalar engines need one output and ylimits"; return false; }
                en.ylo = yl[0]; en.yscale = yl[1] - yl[0];
            } else d.n_basis = widths[en.n_layers] + 1;    // last hidden layer + the bias row of the folded final layer
        } else if (en.type == 2) {
            // stacked table engine (include/desilike_amd.h; emulators/conversion.py:44-98): networks with the same hidden layers, in groups that feed ranges of bias monomials
            if (e != 0) { err = q + "only the table engine can be a stack of networks"; return false; }
            const auto& xl = cfg.F(q + "xlimits");
            const auto& widths = cfg.I(q + "widths");
            const auto& w = cfg.F(q + "weights");
            const auto& groups = cfg.I(q + "groups");
            const auto& scale = cfg.F(q + "scale");
            en.act = cfg.i(q + "act", 0);
            en.n_layers = (int)widths.size() - 1;
            if ((int)xl.size() != 2 * d.n_x || en.n_layers < 1 || en.n_layers > DL_MAX_LAYERS || widths[0] != d.n_x) { err = q + "inconsistent description of the stacked networks"; return false; }
            size_t per = 0;
            for (int l = 0; l <= en.n_layers; ++l) {
                en.widths[l] = widths[l];
                if (widths[l] < 1 || widths[l] > 128) { err = q + "stacked networks: layer widths must be in [1, 128]"; return false; }
                if (l) per += (size_t)widths[l - 1] * widths[l] + widths[l];
            }
            const int H = widths[en.n_layers], ng = (int)(groups.size() / 4);
            if (d.mono_mode == 0) { err = q + "stacked networks feed the velocileptors bias monomials (mono_mode 1 .. 4)"; return false; }
            if (ng < 1 || groups.size() != (size_t)ng * 4 || scale.size() != (size_t)ng * (d.n_x + 1)) { err = q + "groups i32[n_groups * 4] and scale f64[n_groups * (n_x + 1)] are required"; return false; }
            int n_trunks = 0;
            for (int gi = 0; gi < ng; ++gi) n_trunks = std::max(n_trunks, groups[4 * gi + 1]);
            // (no network survives -- every table a constant, e.g. a tracer of the 'st' tables alone: the weights array is then a placeholder of any size: ADVICE r5)
            if (per == 0 || (n_trunks > 0 && w.size() != per * (size_t)n_trunks)) { err = q + "weights size does not match widths x number of networks"; return false; }
            std::vector<double> table;
            std::vector<double> amp;
            int col = 0, kq = 0, max_k = 1, covered[DL_N_MONO] = {0};
            for (int gi = 0; gi < ng; ++gi) {
                const int tb = groups[4 * gi], te = groups[4 * gi + 1], m0 = groups[4 * gi + 2], m1 = groups[4 * gi + 3];
                if (tb < 0 || te < tb || te > n_trunks || m0 < 0 || m1 <= m0 || m1 > DL_N_MONO) { err = q + "groups: (first network, one past the last, first monomial, one past the last) out of range"; return false; }
                for (int m = m0; m < m1; ++m) { if (covered[m]++) { err = q + "groups: a bias monomial belongs to one group"; return false; } }
                const int K = (te - tb) * H + 1, nm = m1 - m0;
                max_k = std::max(max_k, K);
                // device groups of at most DL_STK_MAX_MONO monomials (the accumulator tiles of the feature GEMM), the same networks
                for (int ma = m0; ma < m1; ma += DL_STK_MAX_MONO) {
                    const int mb = std::min(m1, ma + DL_STK_MAX_MONO);
                    const double rec[DL_STK_REC] = {(double)tb, (double)te, (double)ma, (double)mb, (double)col, (double)nm, (double)(ma - m0), (double)kq};
                    table.insert(table.end(), rec, rec + DL_STK_REC);
                    amp.insert(amp.end(), scale.begin() + (size_t)gi * (d.n_x + 1), scale.begin() + (size_t)(gi + 1) * (d.n_x + 1));
                    kq += (K + 7) / 8 * (mb - ma);
                }
                col += K * nm;
            }
            d.stk.n_groups = (int)(table.size() / DL_STK_REC);
            if (d.stk.n_groups > DL_STK_MAX_GROUPS) { err = q + "at most 8 groups of networks"; return false; }
            d.stk.n_trunks = n_trunks; d.stk.trunk_doubles = (int32_t)per; d.stk.max_k = max_k;
            d.stk.max_net = (max_k - 1) / H;
            d.n_basis = col;    // (columns of the theory vector: n_kin below)
            xlo.assign(d.n_x, 0.); xinv.assign(d.n_x, 0.);
            for (int c = 0; c < d.n_x; ++c) { xlo[c] = xl[2 * c]; xinv[c] = 1. / (xl[2 * c + 1] - xl[2 * c]); }
            weights = w;
            // the same weights in MFMA fragment order for the batched kernel (dl_emu_stacked.h): a B-operand load is then base + lane + immediate, zero padding instead of predicates
            std::vector<double> wfrag;
            for (int t = 0; t < n_trunks; ++t) {
                const double* wl = w.data() + (size_t)t * per;
                for (int l = 0; l < en.n_layers; ++l) {
                    const int nin = widths[l], nout = widths[l + 1], ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16;
                    for (int tile = 0; tile < tiles; ++tile)
                        for (int u = 0; u < ksteps; ++u)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int k = 4 * u + (lane >> 4), oc = 16 * tile + (lane & 15);
                                wfrag.push_back(k < nin && oc < nout ? wl[(size_t)k * nout + oc] : 0.);
                            }
                    for (int oc = 0; oc < 16 * tiles; ++oc) wfrag.push_back(oc < nout ? wl[(size_t)nin * nout + oc] : 0.);
                    wl += (size_t)nin * nout + nout;
                }
                if (t == 0) d.stk.frag_doubles = (int32_t)wfrag.size();
            }
            if (wfrag.empty()) wfrag.assign(2, 0.);
            oh.off_stk[0] = arena.push(table); oh.off_stk[1] = arena.push(amp); oh.off_stk[2] = arena.push(wfrag);
        } else if (en.type == 1) {
            const auto& ce = cfg.F(q + "center");
            const auto& po = cfg.I(q + "powers");
            const auto& co = cfg.F(q + "coef");
            en.n_terms = (int)(po.size() / d.n_x);
            if ((int)ce.size() != d.n_x || en.n_terms < 1 || en.n_terms > DL_MAX_WIDTH || (size_t)en.n_terms * d.n_x != po.size()) { err = q + "inconsistent Taylor description (<= 256 terms)"; return false; }
            center = ce;
            powers.assign(po.begin(), po.end());
            if (e != 0) {
                if ((int)co.size() != en.n_terms) { err = q + "coef size must match powers"; return false; }
                coef = co;
            } else d.n_basis = en.n_terms;
        }
        oh.off_eng[e][0] = arena.push(xlo); oh.off_eng[e][1] = arena.push(xinv); oh.off_eng[e][2] = arena.push(weights);
        oh.off_eng[e][3] = arena.push(center); oh.off_eng[e][4] = arena.push(powers); oh.off_eng[e][5] = arena.push(coef);
    }
    d.n_ell = 1; d.n_kin = d.eng[0].type == 2 ? d.n_basis : d.n_basis * d.n_mono; d.n_in = d.n_kin; d.ell0 = -1;
    d.n_mu = 0; d.n_t = 0;
    const auto& pin = cfg.F(p + "in.pass");
    d.n_pass = (int)(pin.size() / 2);
    if (d.n_pass > DL_MAX_PASS) { err = p + "at most 32 pass-through parameters supported"; return false; }
    const auto& mpass = cfg.I(p + "marg.pass");
    for (int c = 0; c < DL_MAX_PASS; ++c) oh.marg_pass[c] = (c < (int)mpass.size()) ? mpass[c] : -1;
    {
        std::vector<double> pass_tab(std::max<size_t>(pin.size(), 2), 0.);
        for (int c = 0; c < d.n_pass; ++c) {
            pass_tab[2 * c] = (double)std::lround(pin[2 * c]); pass_tab[2 * c + 1] = pin[2 * c + 1];
            if ((int)pass_tab[2 * c] >= n_params) { err = p + "in.pass: theta column out of range"; return false; }
        }
        oh.off_pass = arena.push(pass_tab);
        oh.off_png = oh.off_pass; oh.off_band = oh.off_pass;
    }
    for (int c = 0; c < DL_MAX_EFT; ++c) { oh.marg_sn[c] = -1; oh.marg_ct[c][0] = oh.marg_ct[c][1] = -1; d.marg_ct_slot[c][0] = d.marg_ct_slot[c][1] = -1; }
    oh.marg_sn0 = -1;
    // unused arrays still need valid offsets
    std::vector<double> dummy(2, 0.);
    size_t off = arena.push(dummy);
    oh.off_kin = oh.off_lkin = oh.off_mu = oh.off_wmu = oh.off_xt = oh.off_pk = oh.off_th = oh.off_lg = oh.off_ih = oh.off_dlt = oh.off_A = oh.off_nC = oh.off_inv = off;
    oh.off_gf = oh.off_gb = oh.off_coef = oh.off_ct = oh.off_sn = oh.off_cw = oh.off_cn = oh.off_pknowk = oh.off_ml = off;
    return dl_build_window(cfg, p, oh, err);
}

// Build the constants of observable `iobs` from the key/value store.
inline bool dl_build_obs(const dl_config& cfg, int iobs, int n_params, DlObsHost& oh, DlArena& arena, std::string& err) {
    std::string p = "obs" + std::to_string(iobs) + ".";
    DlObsDev& d = oh.dev;
    std::memset(&d, 0, sizeof(d));
    d.theory = cfg.i(p + "theory", 0);
    d.templ = cfg.i(p + "template", 0);
    d.apmode = cfg.i(p + "apmode", 0);
    d.transform = cfg.i(p + "transform", 0);
    d.damping_fid = cfg.i(p + "damping_fid", 0);
    if (d.theory == 3) return dl_build_emulated_obs(cfg, p, n_params, oh, arena, err);
    const auto& ells = cfg.I(p + "ells_in");
    const auto& kin = cfg.F(p + "kin");
    const auto& mu = cfg.F(p + "mu");
    const auto& wmu = cfg.F(p + "wmu_ell");
    const auto& k_t = cfg.F(p + "k_t");
    const auto& pk = cfg.F(p + "pk_dd_fid");
    d.n_ell = (int)ells.size(); d.n_kin = (int)kin.size(); d.n_mu = (int)mu.size(); d.n_t = (int)k_t.size();
    d.n_in = d.n_ell * d.n_kin;
    if (d.n_ell < 1 || d.n_ell > DL_MAX_ELL) { err = p + "ells_in: between 1 and 5 multipoles supported"; return false; }
    if (d.n_mu < 1 || d.n_mu > (d.theory == 5 ? DL_PNG_MAX_MU : DL_MAX_MU)) { err = p + "mu: between 1 and 32 nodes supported (48 by the PNG theory)"; return false; }
    if (d.n_kin < 1) { err = p + "kin missing"; return false; }
    if ((int)wmu.size() != d.n_ell * d.n_mu) { err = p + "wmu_ell must have n_ell * n_mu entries"; return false; }
    if ((int)pk.size() != d.n_t) { err = p + "pk_dd_fid and k_t sizes differ"; return false; }
    d.ell0 = -1;
    for (int l = 0; l < d.n_ell; ++l) if (ells[l] == 0) d.ell0 = l;
    d.eta = cfg.f(p + "eta", 1. / 3.);
    d.f_fid = cfg.f(p + "f_fid", 1.);
    d.a = cfg.f(p + "a", 0.6);
    double kp = cfg.f(p + "kp", 0.03);
    d.nd = cfg.f(p + "nd", 1e-4);
    struct { const char* name; DlInput* in; double def; } inputs[] = {
        {"qpar", &d.qpar, 1.}, {"qper", &d.qper, 1.}, {"qiso", &d.qiso, 1.}, {"qap", &d.qap, 1.}, {"df", &d.df, 1.}, {"dm", &d.dm, 0.}, {"dn", &d.dn, 0.},
        {"sigmapar", &d.sigpar, 0.}, {"sigmaper", &d.sigper, 0.}, {"b1X", &d.b1X, 1.}, {"b1Y", &d.b1Y, 1.}, {"sn0", &d.sn0, 0.},
        {"dbeta", &d.dbeta, 1.}, {"sigmas", &d.sigmas, 0.}, {"dres", &d.dres, 1.},
        {"sigmav", &d.sigmav, 0.}, {"b2", &d.b2, 0.}, {"bs", &d.bs, 0.}, {"b3", &d.b3, 0.},
        {"bv", &d.bv, 1.}, {"sigmau", &d.sigmau, 0.},
        {"m", &d.to_m, 0.6}, {"n", &d.to_n, 0.9}, {"qto", &d.qto, 1.}, {"dpto", &d.dpto, 1.},
        {"fnl_loc", &d.fnl, 0.}, {"pX", &d.pX, 1.}, {"pY", &d.pY, 1.}, {"bphiX", &d.bphiX, 1.}, {"bphiY", &d.bphiY, 1.}, {"sigmasY", &d.sigmasY, 0.}};
    for (auto& it : inputs) {
        *it.in = dl_input_from(cfg, p + "in." + it.name, it.def);
        if (it.in->col >= n_params) { err = p + "in." + it.name + ": theta column out of range"; return false; }
    }
    std::vector<double> pass_tab;
    {
        const auto& pin = cfg.F(p + "in.pass");
        d.n_pass = (int)(pin.size() / 2);
        if (d.n_pass > DL_MAX_PASS) { err = p + "at most 32 pass-through (broadband) parameters supported"; return false; }
        const auto& mpass = cfg.I(p + "marg.pass");
        for (int c = 0; c < DL_MAX_PASS; ++c) oh.marg_pass[c] = (c < (int)mpass.size()) ? mpass[c] : -1;
        pass_tab.assign(std::max<size_t>(pin.size(), 2), 0.);
        for (int c = 0; c < d.n_pass; ++c) {
            pass_tab[2 * c] = (double)std::lround(pin[2 * c]); pass_tab[2 * c + 1] = pin[2 * c + 1];
            if ((int)pass_tab[2 * c] >= n_params) { err = p + "in.pass: theta column out of range"; return false; }
        }
        d.bao_mode = cfg.i(p + "bao_mode", 0);
        d.smoothing_radius = cfg.f(p + "smoothing_radius", 15.);
        const auto& res = cfg.F(p + "resummed");   // sigma_dd^2, sigma_nl^2, sigma_x^2, shotnoise * sigma_sn^2 (bao.py:186-199)
        for (int q = 0; q < 4; ++q) d.res_sig[q] = q < (int)res.size() ? res[q] : 0.;
    }
    if (d.templ == 2) {
        const double kto = cfg.f(p + "kto_fid", 0.), pkto = cfg.f(p + "pkto_fid", 0.);
        if (!(kto > 0.) || !(pkto > 0.)) { err = p + "turn-over template: kto_fid and pkto_fid must be positive"; return false; }
        if (d.theory != 0 && d.theory != 1) { err = p + "turn-over template: Kaiser / EFT-like Kaiser / Simple theories only"; return false; }
        d.lkto_fid = std::log10(kto); d.lpkto_fid = std::log(pkto);
    } else if (d.templ == 3) {
        const auto& bin = cfg.F(p + "in.band");
        const auto& btab = cfg.F(p + "band_templates");
        d.n_band = (int)(bin.size() / 2);
        if (d.n_band < 1 || d.n_band > DL_MAX_BAND) { err = p + "band template: 1 .. 16 bands (in.band)"; return false; }
        if ((int)btab.size() != d.n_band * d.n_t) { err = p + "band template: band_templates must be [n_band, n_t]"; return false; }
        if (d.theory != 0 && d.theory != 1) { err = p + "band template: Kaiser / EFT-like Kaiser / Simple theories only"; return false; }
        for (int i = 0; i < d.n_band; ++i) {
            d.band_in[i].col = (int)std::lround(bin[2 * i]); d.band_in[i].value = bin[2 * i + 1];
            if (d.band_in[i].col >= n_params) { err = p + "in.band: theta column out of range"; return false; }
        }
    } else if (d.templ < 0 || d.templ > 3) { err = p + "unknown template kind"; return false; }
    // template knots in log10 k (full_shape.py:498: interp1d(log10(kap), log10(k11), pk11))
    std::vector<double> x_t(d.n_t), sf_th(d.n_t), sf_lg(d.n_t), lkin(d.n_kin);
    for (int j = 0; j < d.n_t; ++j) {
        x_t[j] = std::log10(k_t[j]);
        sf_lg[j] = std::log(k_t[j] / kp);               // power_template.py:749
        sf_th[j] = std::tanh(d.a * std::log(k_t[j] / kp));
    }
    for (int i = 0; i < d.n_kin; ++i) lkin[i] = std::log10(kin[i]);
    DlSplineSetup sp;
    if (!dl_spline_setup(x_t, sp, err)) { err = p + err; return false; }
    d.end0a = sp.end0a; d.end0b = sp.end0b; d.end1a = sp.end1a; d.end1b = sp.end1b;
    d.x0 = x_t[0];
    d.inv_hx = (d.n_t - 1) / (x_t[d.n_t - 1] - x_t[0]);
    d.fixed_spline = (d.templ == 0) && d.theory != 5;   // (the PNG kernel builds its two splines per point)
    // deviation of the knot table from an exactly uniform grid in log10 k (geomspace knots: rounding only)
    std::vector<double> dlt(d.n_t);
    d.uniform_knots = 1;
    for (int j = 0; j < d.n_t; ++j) {
        dlt[j] = x_t[j] - (d.x0 + j / d.inv_hx);
        if (std::fabs(dlt[j] * d.inv_hx) > 1e-6) d.uniform_knots = 0;
    }
    // knots uniform to rounding (geomspace tables): the spline system is Toeplitz and is inverted by a convolution (dl_fs_phase2_fir);
    // DL_NO_TOEPLITZ=1 keeps the general segmented sweeps (diagnostics)
    d.toeplitz = (d.uniform_knots && d.n_t >= 4 * DL_FIR_PAD && !getenv("DL_NO_TOEPLITZ")) ? 1 : 0;
    for (int j = 0; j + 1 < d.n_t && d.toeplitz; ++j)
        if (std::fabs((x_t[j + 1] - x_t[j]) * d.inv_hx - 1.) > 1e-11) d.toeplitz = 0;
    // segmented sweeps: 64 segments, the state entering each is a dot product with products of the multipliers
    int m = d.n_t - 2;
    d.seg_warm = sp.warm;
    d.n_seg = DL_MAX_SEG;
    d.seg_len = (m + d.n_seg - 1) / d.n_seg;
    if (d.seg_warm > DL_SEG_PARTS * DL_SEG_QMAX) { err = p + "template knots too irregular for the segmented spline solve (sweep multipliers decay too slowly)"; return false; }
    std::vector<double> gf((size_t)DL_SEG_QMAX * DL_FS_THREADS, 0.), gb((size_t)DL_SEG_QMAX * DL_FS_THREADS, 0.);
    auto slot = [](int dd, int sgm) { return (size_t)(dd / DL_SEG_PARTS) * DL_FS_THREADS + (size_t)sgm * DL_SEG_PARTS + (dd % DL_SEG_PARTS); };
    for (int sgm = 0; sgm < d.n_seg; ++sgm) {
        int start = sgm * d.seg_len, end = std::min(start + d.seg_len, m);
        if (start >= m) continue;
        double g = 1.;
        for (int dd = 0; dd < d.seg_warm && start - 1 - dd >= 0; ++dd) {   // weight of B_{start-1-dd} in z_{start-1}
            gf[slot(dd, sgm)] = g;
            g *= sp.A[start - 1 - dd];
        }
        g = 1.;
        for (int dd = 0; dd < d.seg_warm && end + dd < m; ++dd) {           // weight of z_{end+dd} in u_end
            gb[slot(dd, sgm)] = g;
            g *= sp.nC[end + dd];
        }
    }
    // interval polynomials of the fiducial table itself (fixed templates skip the per-point spline build)
    std::vector<double> Mfix, coef((size_t)4 * d.n_t, 0.);
    dl_spline_moments_serial(pk, sp, Mfix);
    {
        std::vector<double> lds(dl_fs_shared_doubles(d.n_t, d.n_in), 0.);
        DlFsShared sh = dl_fs_shared_carve(lds.data(), d.n_t, d.n_in);
        for (int j = 0; j < d.n_t; ++j) { sh.y[j] = pk[j]; sh.M[j] = Mfix[j]; }
        DlObsDev tmp = d;
        tmp.ih = sp.ih.data(); tmp.x_t = x_t.data(); tmp.dlt = dlt.data();
        dl_fs_phase2d(0, 1, tmp, sh);
        for (int j = 0; j < d.n_t; ++j)   // the two planes of the shared layout -> rows [n_t][4] (what coef_fixed / coef_w / coef_n hold)
            for (int q = 0; q < 4; ++q) coef[(size_t)4 * j + q] = sh.coef[(size_t)(q >> 1) * 2 * d.n_t + 2 * j + (q & 1)];
    }
    // BAO wiggle model (bao.py:117-140): constant splines of P_now and of the wiggle P_dd - P_now, P_now at the fiducial k
    std::vector<double> coef_w(4, 0.), coef_n(4, 0.), pknow_k(2, 0.);
    if (d.theory == 2) {
        const auto& pknow = cfg.F(p + "pknow_dd_fid");
        if ((int)pknow.size() != d.n_t) { err = p + "pknow_dd_fid (no-wiggle table) is required by the BAO model"; return false; }
        auto interval_polynomials = [&](const std::vector<double>& y, std::vector<double>& out) {
            std::vector<double> Mv;
            dl_spline_moments_serial(y, sp, Mv);
            std::vector<double> lds(dl_fs_shared_doubles(d.n_t, d.n_in), 0.);
            DlFsShared sh = dl_fs_shared_carve(lds.data(), d.n_t, d.n_in);
            for (int j = 0; j < d.n_t; ++j) { sh.y[j] = y[j]; sh.M[j] = Mv[j]; }
            DlObsDev tmp = d;
            tmp.ih = sp.ih.data(); tmp.x_t = x_t.data(); tmp.dlt = dlt.data();
            dl_fs_phase2d(0, 1, tmp, sh);
            out.assign((size_t)4 * d.n_t, 0.);
            for (int j = 0; j < d.n_t; ++j)
                for (int q = 0; q < 4; ++q) out[(size_t)4 * j + q] = sh.coef[(size_t)(q >> 1) * 2 * d.n_t + 2 * j + (q & 1)];
        };
        std::vector<double> wig(d.n_t);
        for (int j = 0; j < d.n_t; ++j) wig[j] = pk[j] - pknow[j];
        interval_polynomials(wig, coef_w);
        interval_polynomials(pknow, coef_n);
        pknow_k.assign(d.n_kin, 0.);
        DlObsDev tmp = d;
        tmp.ih = sp.ih.data(); tmp.x_t = x_t.data();
        for (int i = 0; i < d.n_kin; ++i) {
            int j; double u;
            dl_spline_locate<false>(tmp, lkin[i], j, u);
            const double* c = &coef_n[(size_t)4 * j];
            pknow_k[i] = std::fma(std::fma(std::fma(c[3], u, c[2]), u, c[1]), u, c[0]);
        }
    }
    // flexible BAO wiggles: kernel matrix K [n_ml, n_kin] and Legendre table [n_ell, n_mu] travel in the slots of the (otherwise unused) EFT matrices
    const bool flexible = d.theory == 2 && cfg.has_f64(p + "ml_matrix");
    std::vector<double> ml_tab;
    if (flexible) {
        const auto& mlin = cfg.F(p + "in.ml");
        const auto& mlell = cfg.I(p + "ml_ell");
        d.n_ml = (int)mlell.size();
        if (d.n_ml > DL_MAX_ML) { err = p + "at most 40 multiplicative wiggle terms supported"; return false; }
        if ((int)mlin.size() != 2 * d.n_ml || (int)cfg.F(p + "ml_matrix").size() != d.n_ml * d.n_kin || (int)cfg.F(p + "legendre").size() != d.n_ell * d.n_mu) {
            err = p + "ml_matrix / ml_ell / in.ml / legendre sizes do not match"; return false;
        }
        for (int q = 0; q < d.n_ml; ++q) {
            const int col = (int)std::lround(mlin[2 * q]);
            if (col >= n_params || mlell[q] < 0 || mlell[q] >= d.n_ell) { err = p + "in.ml / ml_ell out of range"; return false; }
            ml_tab.push_back((double)col); ml_tab.push_back(mlin[2 * q + 1]); ml_tab.push_back((double)mlell[q]);
        }
    }
    // EFT-like terms
    const auto& ctm = flexible ? cfg.F(p + "ml_matrix") : cfg.F(p + "ct_matrix");
    const auto& snm = flexible ? cfg.F(p + "legendre") : cfg.F(p + "sn_matrix");
    d.n_ct = (flexible || ctm.empty()) ? 0 : (int)(ctm.size() / d.n_in);
    d.n_sn = (flexible || snm.empty()) ? 0 : (int)(snm.size() / d.n_in);
    if (d.n_ct > DL_MAX_EFT || d.n_sn > DL_MAX_EFT) { err = p + "at most 8 counter / stochastic terms supported"; return false; }
    if (d.n_ct > 0 && d.ell0 < 0) { err = p + "counter terms need the monopole in ells_in (full_shape.py:633)"; return false; }
    const auto& ctin = cfg.F(p + "in.ct");
    const auto& snin = cfg.F(p + "in.sn");
    if ((int)ctin.size() != d.n_ct * 4 || (int)snin.size() != d.n_sn * 2) { err = p + "in.ct / in.sn sizes do not match the matrices"; return false; }
    for (int c = 0; c < d.n_ct; ++c)
        for (int t = 0; t < 2; ++t) {
            d.ct_in[c][t].col = (int32_t)std::lround(ctin[(c * 2 + t) * 2]);
            d.ct_in[c][t].value = ctin[(c * 2 + t) * 2 + 1];
            if (d.ct_in[c][t].col >= n_params) { err = p + "in.ct: theta column out of range"; return false; }
        }
    for (int c = 0; c < d.n_sn; ++c) {
        d.sn_in[c].col = (int32_t)std::lround(snin[c * 2]);
        d.sn_in[c].value = snin[c * 2 + 1];
        if (d.sn_in[c].col >= n_params) { err = p + "in.sn: theta column out of range"; return false; }
    }
    {
        const auto& msn0 = cfg.I(p + "marg.sn0");
        const auto& msn = cfg.I(p + "marg.sn");
        const auto& mct = cfg.I(p + "marg.ct");
        oh.marg_sn0 = msn0.empty() ? -1 : msn0[0];
        for (int c = 0; c < DL_MAX_EFT; ++c) {
            oh.marg_sn[c] = (c < (int)msn.size()) ? msn[c] : -1;
            for (int t = 0; t < 2; ++t) { oh.marg_ct[c][t] = (2 * c + t < (int)mct.size()) ? mct[2 * c + t] : -1; d.marg_ct_slot[c][t] = -1; }
        }
        d.n_var = 0;
    }
    oh.off_kin = arena.push(kin); oh.off_lkin = arena.push(lkin); oh.off_mu = arena.push(mu); oh.off_wmu = arena.push(wmu);
    oh.off_xt = arena.push(x_t); oh.off_pk = arena.push(pk); oh.off_th = arena.push(sf_th); oh.off_lg = arena.push(sf_lg);
    oh.off_ih = arena.push(sp.ih); oh.off_dlt = arena.push(dlt); oh.off_A = arena.push(sp.A); oh.off_nC = arena.push(sp.nC); oh.off_inv = arena.push(sp.inv);
    oh.off_gf = arena.push(gf); oh.off_gb = arena.push(gb); oh.off_coef = arena.push(coef);
    oh.off_ct = arena.push(ctm); oh.off_sn = arena.push(snm);
    oh.off_cw = arena.push(coef_w); oh.off_cn = arena.push(coef_n); oh.off_pknowk = arena.push(pknow_k);
    std::vector<double> png_alpha(2, 0.);
    if (d.theory == 5) {   // scale-dependent bias (primordial_non_gaussianity.py:75-112)
        const auto& al = cfg.F(p + "png_alpha");
        if ((int)al.size() != d.n_t) { err = p + "png_alpha (alpha at the template knots) is required by the PNG theory"; return false; }
        if (d.n_ct > 0 || d.n_sn > 0 || d.n_pass > 0) { err = p + "counter / stochastic / pass-through terms are not part of the PNG theory"; return false; }
        png_alpha = al;
        d.png_mode = cfg.i(p + "png_mode", 1);
        d.png_vel = cfg.i(p + "png_velocity", 0);
        d.png_velfac = cfg.f(p + "png_velfac", 100.);
        const auto& nk = cfg.F(p + "png_knorm");   // normalisation wavenumber of the transfer function (methods other than 'prim'), absent: none
        if (!nk.empty()) { d.png_lg0 = std::log(nk[0] / kp); d.png_th0 = std::tanh(d.a * d.png_lg0); }
    }
    if (ml_tab.empty()) ml_tab.assign(3, 0.);
    oh.off_ml = arena.push(ml_tab);
    oh.off_pass = arena.push(pass_tab);
    oh.off_png = arena.push(png_alpha);
    {
        std::vector<double> band_tab = d.templ == 3 ? cfg.F(p + "band_templates") : std::vector<double>(2, 0.);
        oh.off_band = arena.push(band_tab);
    }

    for (int c = 0; c < DL_N_VPARS; ++c) { oh.marg_vp[c] = -1; d.vp_slot[c] = -1; }
    for (int e = 0; e < 3; ++e) { d.eng[e].type = -1; for (int q = 0; q < 6; ++q) oh.off_eng[e][q] = oh.off_kin; }
    if (d.theory == 4) {   // TNS one-loop tables: geometry on the device, once (dl_tns.h)
        const auto& k11 = cfg.F(p + "tns_k11");
        const auto& tmu = cfg.F(p + "tns_mu");
        const auto& twmu = cfg.F(p + "tns_wmu");
        if (k11.empty() || tmu.empty() || tmu.size() != twmu.size()) { err = p + "tns_k11 / tns_mu / tns_wmu are required by the TNS theory"; return false; }
        if (d.n_pass != 0) { err = p + "pass-through columns are not supported by the TNS theory"; return false; }
        if (dl_tns_assemble_doubles((int)k11.size(), d.n_in, d.n_kin, 1) * sizeof(double) > 156 * 1024) { err = p + "TNS table grid too large for the LDS of the assembly kernel (about 550 table wavenumbers)"; return false; }
        oh.tns_k11 = k11; oh.tns_mu = tmu; oh.tns_wmu = twmu; oh.tns_kt = k_t; oh.tns_fog = cfg.i(p + "tns_fog", 0);
#ifdef __HIPCC__
        const char* terr = nullptr;
        oh.tns = dl_tns_create(k11.data(), (int)k11.size(), k_t.data(), d.n_t, tmu.data(), twmu.data(), (int)tmu.size(), oh.tns_fog, &terr);
        if (!oh.tns) { err = p + (terr ? terr : "tns: plan creation failed"); return false; }
        d.tns_plan = oh.tns;
#else
        d.tns_plan = &oh;   // CPU emulation (tests/csrc/emulate.cpp): the host record itself (it must not move while the observable is in use)
#endif
    }
    return dl_build_window(cfg, p, oh, err);
}
