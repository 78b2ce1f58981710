// dl_api.hip -- C ABI (include/desilike_amd.h) over the gfx950 kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/desilike_amd.h"
#include "dl_host.hpp"
#include "dl_kernels.h"
#include "dl_ens_fold.h"
#include "dl_prior.h"

static thread_local std::string g_last_error;
void dl_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }

struct dl_ctx {
    int device = 0;
    int n_params = 0, n_obs = 0, n_data = 0;
    int n_white = 0;                 // width of the whitened residual rows (= n_data, or more when the observables' row ranges are aligned to 16: see dl_create)
    int N_pad = 0, K_pad = 0, K_live = 0, max_n_t = 0;   // K_live: columns of the theory vector before the padding to whole GEMM panels
    bool any_transform = false;
    bool priors_general = false;     // a prior of a kind beyond uniform / norm is present (dl_prior.h)
    // analytic marginalisation
    int n_solved = 0, n_var = 0;
    DlMargDev marg;
    double* tconst_dev = nullptr;    // [n_solved, N_pad]
    std::vector<DlObsHost> obs;
    std::vector<int> obs_row0;       // first data row of each observable
    // device constants
    double* arena_dev = nullptr;     // per-observable theory constants
    std::vector<DlObsDev> obs_kernarg; // observables with device pointers, passed by value to the theory kernel
    DlObsDev* obs_array_dev = nullptr; // the same structs in device memory (one theory launch for several observables)
    double* priors_dev = nullptr;    // [P, 5]
    int32_t* gemm_counters = nullptr;// [<= 2048 / 32 + 8] arrival counters of the fused chi2 GEMM finalize (zero between launches)
    int32_t* step_ready = nullptr;   // [32 + 8] arrival counters of the row blocks' producers in dl_step_kernel (zero between launches); behind gemm_counters in one allocation
    double* wt_white_dev = nullptr;  // [N_pad, K_pad]  L^T . blockdiag(W_obs)          (chi2 path)
    double* wt_frag_dev = nullptr;   // the same in MFMA fragment order [N_pad / 16][K_pad / 4][64]: k-step ks of column block nt, lane l = W~[16 nt + (l & 15)][4 ks + (l >> 4)] (dl_chi2_gemm_tile_bf)
    std::vector<uint8_t> panel_ranges;   // [N_pad / 16][2]: 128-wide K panels of wt_white with non-zero entries per 16-row column block (chi2 GEMM skips the others)
    double* bias_white_dev = nullptr;// [N_pad]         L^T . (bias - flatdata)
    double* wt_full_dev = nullptr;   // [N_pad, K_pad]  blockdiag(W_obs)                 (flattheory path)
    double* bias_full_dev = nullptr; // [N_pad]
    double* wh_dev = nullptr;        // [N_pad, N_pad]  L^T                              (transform path)
    double* bias_wh_dev = nullptr;   // [N_pad]        -L^T . flatdata
    double* flatdata_dev = nullptr;  // [N_pad]
    int32_t* transform_dev = nullptr;// [N_pad]
    // feature path of the emulated theories (dl_feature_gemm.h): every observable separable, no pass-through columns
    bool feat_ok = false;
    int64_t feat_ld = 0;             // doubles per point record (all observables)
    std::vector<double*> gfrag_dev;  // per observable: whitened folded operator in MFMA fragment order
    std::vector<int> stk_steps;      // per observable: operand steps per column block of a stacked table engine (dl_emu_stacked.h), 0: not stacked
    bool any_stacked = false;
    double* feat_ws = nullptr;       // [cap, feat_ld]
    // workspaces (grown on demand, never inside dl_eval_* once large enough)
    int64_t cap = 0;
    double* power_ws = nullptr;      // [cap, K_pad]
    double* delta_ws = nullptr;      // [cap, N_pad]
    double* flat_ws = nullptr;       // [cap, N_pad]  (transform path / flattheory staging)
    double* stencil_ws = nullptr;    // [cap, n_params] theta rows of the Fisher stencil
    // host staging for the *_host entry points
    int64_t stage_cap = 0;
    double* theta_stage = nullptr;   // device
    double* out_stage = nullptr;     // device: loglike[cap] | logprior[cap]
    int32_t* status_stage = nullptr; // device
    double* host_stage = nullptr;    // pinned host mirror: theta[cap * P] | out[3 * cap]
    hipStream_t host_stream = nullptr;   // private stream of the *_host entry points
    double* host_stage_dev = nullptr;    // the pinned buffer as the device sees it (hipHostMallocMapped): theta read and results written by the kernels themselves
    uint64_t* host_flag = nullptr;       // pinned, mapped: sequence number of the last finished *_host call, written by the device after the results
    uint64_t* host_flag_dev = nullptr;
    uint64_t host_seq = 0;
    hipEvent_t host_event = nullptr;
    // analytic gradient (dl_eval_logposterior_grad): -W~^T [K_pad, N_pad], a zero bias [K_pad], residual rows [cap, N_pad], Y [cap, K_pad], per-observable sums [cap, n_obs, 8]
    double *grad_wtT = nullptr, *grad_zero = nullptr, *grad_delta = nullptr, *grad_y = nullptr, *grad_phys = nullptr;
    int32_t* grad_status = nullptr;
    int64_t grad_cap = 0;
    // the workspaces are shared by every call on this context: a call on another stream than the previous one waits for it (event recorded on the old stream
    // at the moment of the switch: calls that stay on one stream pay nothing)
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    hipEvent_t order_event = nullptr;
    // profiling
    bool profile = false;
    static const int NPOOL = 256;            // event sets kept: dl_profile_read averages over the calls recorded since dl_profile_enable
    std::vector<int8_t> prof_phase_of;       // [NPOOL] phase that carried events in each sampled call (-1: all)
    std::vector<hipEvent_t> ev;              // [NPOOL * 6]: (start, stop) of the theory kernel, the GEMM, the finalize kernel
    int64_t prof_calls = 0;                  // profiled calls recorded
    int64_t eval_calls = 0;                  // dl_eval_batch calls since dl_profile_enable
    int prof_every = 1;                      // record events on one call out of prof_every (sampling keeps the event overhead out of the throughput)
    bool prof_rotate = false;                // single-kernel mode: only phase prof_only carries events on a sampled call
    int prof_only = 0;
    std::string last_error;
};

#define DL_HIP_CHECK(ctx, call)                                                                                  \
    do {                                                                                                         \
        hipError_t err__ = (call);                                                                               \
        if (err__ != hipSuccess) {                                                                               \
            std::string msg__ = std::string(#call) + ": " + hipGetErrorString(err__);                            \
            if (ctx) (ctx)->last_error = msg__;                                                                  \
            g_last_error = msg__;                                                                                \
            return 1;                                                                                            \
        }                                                                                                        \
    } while (0)

// Serialise the calls on one context across streams (the header's contract): everything enqueued so far on the stream of the previous call is finished before
// work enqueued from now on ``stream`` starts.
static int dl_order_streams(dl_ctx* ctx, hipStream_t stream) {
    if (ctx->has_last_stream && ctx->last_stream != stream) {
        if (!ctx->order_event) DL_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
        DL_HIP_CHECK(ctx, hipEventRecord(ctx->order_event, ctx->last_stream));
        DL_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->order_event, 0));
    }
    ctx->last_stream = stream;
    ctx->has_last_stream = true;
    return 0;
}

static int dl_fail(dl_ctx* ctx, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    g_last_error = msg;
    return 1;
}

template <typename T>
static int dl_upload(dl_ctx* ctx, T** dst, const std::vector<T>& src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    DL_HIP_CHECK(ctx, hipMalloc((void**)dst, bytes));
    if (!src.empty()) DL_HIP_CHECK(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static_assert(sizeof(DlObsDev) <= 4096, "DlObsDev travels in the kernarg segment (4 KiB)");

extern "C" {

dl_config* dl_config_new(void) { return new dl_config(); }

int dl_config_set_f64(dl_config* cfg, const char* key, const double* data, int64_t n) {
    if (!cfg || !key || (n > 0 && !data) || n < 0) { g_last_error = "dl_config_set_f64: invalid argument"; return 1; }
    cfg->f64[key] = std::vector<double>(data, data + n);
    return 0;
}

int dl_config_set_i32(dl_config* cfg, const char* key, const int32_t* data, int64_t n) {
    if (!cfg || !key || (n > 0 && !data) || n < 0) { g_last_error = "dl_config_set_i32: invalid argument"; return 1; }
    cfg->i32[key] = std::vector<int32_t>(data, data + n);
    return 0;
}

void dl_config_free(dl_config* cfg) { delete cfg; }

const char* dl_last_error(const dl_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

int dl_create(dl_ctx** out, int device, const dl_config* cfg) {
    if (!out || !cfg) { g_last_error = "dl_create: null argument"; return 1; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_last_error = "dl_create: no HIP device available (this library has no CPU fallback)";
        return 1;
    }
    if (device < 0 || device >= ndev) { g_last_error = "dl_create: device ordinal out of range"; return 1; }
    dl_ctx* ctx = new dl_ctx();
    ctx->device = device;
    auto bail = [&](const std::string& msg) { g_last_error = msg; dl_destroy(ctx); return 1; };
    if (hipSetDevice(device) != hipSuccess) return bail("dl_create: hipSetDevice failed");
    ctx->n_params = cfg->i("n_params", -1);
    ctx->n_obs = cfg->i("n_obs", -1);
    if (ctx->n_params < 1 || ctx->n_obs < 1) return bail("dl_create: n_params and n_obs must be set (>= 1)");
    const auto& priors = cfg->F("priors");
    if ((int)priors.size() != 5 * ctx->n_params) return bail("dl_create: priors must have 5 entries per parameter");
    for (int p = 0; p < ctx->n_params; ++p) {   // (kind, lo, hi, loc, scale): kinds of dl_prior.h
        const double kind = priors[5 * p];
        if (!(kind >= 0. && kind <= (double)DL_PRIOR_MAX_KIND && kind == (double)(int)kind)) return bail("dl_create: unknown prior kind (0 uniform, 1 norm, 2-9: see include/desilike_amd.h)");
        if (kind >= 1. && !(priors[5 * p + 4] > 0.)) return bail("dl_create: the scale of a prior must be positive");
        if (kind >= 2.) ctx->priors_general = true;
    }
    // ---- observables ----
    DlArena arena;
    ctx->obs.resize(ctx->n_obs);
    std::string err;
    int64_t col = 0;
    int row = 0;
    for (int i = 0; i < ctx->n_obs; ++i) {
        if (!dl_build_obs(*cfg, i, ctx->n_params, ctx->obs[i], arena, err)) return bail("dl_create: " + err);
        ctx->obs[i].dev.col_offset = col;
        col += ctx->obs[i].n_cols();
        ctx->obs_row0.push_back(row);
        row += ctx->obs[i].n_out;
        ctx->max_n_t = std::max(ctx->max_n_t, ctx->obs[i].dev.n_t);
        if (ctx->obs[i].dev.transform != 0) ctx->any_transform = true;
    }
    for (auto& ob : ctx->obs)
        if ((ob.dev.theory == 3 || ob.dev.theory == 4 ? 0 : ob.dev.theory == 5 ? dl_png_shared_doubles(ob.dev.n_t, ob.dev.n_in) : ob.dev.theory == 2 ? dl_bao_shared_doubles(ob.dev.n_in) : dl_fs_shared_doubles(ob.dev.n_t, ob.dev.n_in)) * sizeof(double) > 160 * 1024)
            return bail("dl_create: template / theory grid too large for the 160 KiB LDS");
    ctx->n_data = row;
    int n = ctx->n_data;
    int K = (int)col;
    ctx->K_live = K;
    ctx->K_pad = round_up(K, 128);   // whole 128-wide panels of the chi2 GEMM (dl_chi2_gemm.h); padding columns are zero in both operands
    // ---- precision -> Cholesky factor (likelihoods/base.py:13-17: chi2 = d P d = |L^T d|^2) ----
    const auto& prec = cfg->F("precision");
    std::vector<double> L((size_t)n * n, 0.);
    bool dense_factor = false;
    if (cfg->has_f64("precision_factor")) {
        // any factor with precision = F F^T, given by the host (possibly rank-deficient: the posterior marginalised over linear parameters with flat priors,
        // desilike_amd/likelihoods/base.py::_posterior_spec); chi2 = |F^T d|^2
        const auto& fac = cfg->F("precision_factor");
        if ((int64_t)fac.size() != (int64_t)n * n) return bail("dl_create: precision_factor must have n_data^2 entries");
        L = fac;
        dense_factor = true;
    } else if ((int64_t)prec.size() == (int64_t)n * n) {
        L = prec;
        // symmetrise (the reference uses the matrix as given in d.P.d, which only sees the symmetric part)
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) { double v = 0.5 * (L[(size_t)i * n + j] + L[(size_t)j * n + i]); L[(size_t)i * n + j] = L[(size_t)j * n + i] = v; }
        if (!dl_cholesky(L, n)) return bail("dl_create: precision matrix is not positive definite");
    } else if ((int)prec.size() == n) {
        for (int i = 0; i < n; ++i) {
            if (!(prec[i] > 0.)) return bail("dl_create: diagonal precision must be positive");
            L[(size_t)i * n + i] = std::sqrt(prec[i]);
        }
    } else return bail("dl_create: precision must have n_data^2 or n_data entries");
    // ---- position of each whitened row in the residual layout ----
    // Row i of L^T X mixes the data rows j >= i with L[j][i] != 0.  With independent observables (block-diagonal precision: SumLikelihood, joint covariances
    // without cross terms) L is block diagonal and whitened row i belongs to the observable of data row i: each observable's rows then start on a multiple of 16,
    // so that no 16-row column block of the chi2 GEMM straddles two observables (a straddling block needs the K ranges of both: it alone kept the launch at the
    // full K, profiles/r02b).  chi2 is a sum over the whitened rows: where they sit does not matter, padding rows are zero in every operand.
    std::vector<int> wr(n);
    for (int i = 0; i < n; ++i) wr[i] = i;
    ctx->n_white = n;
    if (ctx->n_obs > 1 && !dense_factor && !getenv("DL_NO_ROW_ALIGN")) {
        std::vector<int> owner(n);
        for (int o = 0; o < ctx->n_obs; ++o)
            for (int r = 0; r < ctx->obs[o].n_out; ++r) owner[ctx->obs_row0[o] + r] = o;
        bool block_diagonal = true;
        for (int j = 0; j < n && block_diagonal; ++j)
            for (int i = 0; i < j; ++i)
                if (owner[i] != owner[j] && L[(size_t)j * n + i] != 0.) { block_diagonal = false; break; }
        if (block_diagonal) {
            int pos = 0;
            for (int o = 0; o < ctx->n_obs; ++o) {
                pos = round_up(pos, 16);
                for (int r = 0; r < ctx->obs[o].n_out; ++r) wr[ctx->obs_row0[o] + r] = pos++;
            }
            ctx->n_white = pos;
        }
    }
    ctx->N_pad = round_up(ctx->n_white, 128);   // N tile of the tiled GEMM
    // ---- assemble GEMM operands ----
    size_t NK = (size_t)ctx->N_pad * ctx->K_pad;
    std::vector<double> wt_full(NK, 0.), wt_white(NK, 0.), bias_full(ctx->N_pad, 0.), bias_white(ctx->N_pad, 0.), flatdata(ctx->N_pad, 0.);
    std::vector<double> wh((size_t)ctx->N_pad * ctx->N_pad, 0.), bias_wh(ctx->N_pad, 0.);
    std::vector<int32_t> transform(ctx->N_pad, 0);
    for (int i = 0; i < ctx->n_obs; ++i) {
        const DlObsHost& oh = ctx->obs[i];
        for (int r = 0; r < oh.n_out; ++r) {
            int gr = ctx->obs_row0[i] + r;
            std::memcpy(&wt_full[(size_t)gr * ctx->K_pad + oh.dev.col_offset], &oh.weff[(size_t)r * oh.n_cols()], sizeof(double) * oh.n_cols());
            bias_full[gr] = oh.bias[r];
            flatdata[gr] = oh.flatdata[r];
            transform[gr] = oh.dev.transform;
        }
    }
    // whitened: row i of (L^T . X) = sum_{j >= i} L[j][i] X[j]
    for (int i = 0; i < n; ++i) {
        double* dst = &wt_white[(size_t)wr[i] * ctx->K_pad];
        double bsum = 0., dsum = 0.;
        for (int j = dense_factor ? 0 : i; j < n; ++j) {
            double lji = L[(size_t)j * n + i];
            if (lji == 0.) continue;
            const double* src = &wt_full[(size_t)j * ctx->K_pad];
            for (int k = 0; k < K; ++k) dst[k] += lji * src[k];
            bsum += lji * (bias_full[j] - flatdata[j]);
            dsum += lji * flatdata[j];
            wh[(size_t)wr[i] * ctx->N_pad + j] = lji;
        }
        bias_white[wr[i]] = bsum;
        bias_wh[wr[i]] = -dsum;
    }
    // ---- non-zero K panels per 16-row block of the whitened operator (block-diagonal precisions leave whole panels zero) ----
    if (ctx->K_pad / 128 <= 255 && !getenv("DL_NO_PANEL_SKIP")) {
        const int n_tiles = ctx->N_pad / 16, n_panels = ctx->K_pad / 128;
        ctx->panel_ranges.assign((size_t)2 * n_tiles, 0);
        for (int t = 0; t < n_tiles; ++t) {
            int lo = n_panels, hi = 0;
            for (int r = 16 * t; r < 16 * t + 16; ++r)
                for (int p = 0; p < n_panels; ++p) {
                    const double* seg = &wt_white[(size_t)r * ctx->K_pad + (size_t)p * 128];
                    bool nz = false;
                    for (int k = 0; k < 128 && !nz; ++k) nz = seg[k] != 0.;
                    if (nz) { lo = std::min(lo, p); hi = std::max(hi, p + 1); }
                }
            if (hi <= lo) { lo = 0; hi = 1; }   // an all-zero block (padding rows): one panel of zeros
            ctx->panel_ranges[2 * t] = (uint8_t)lo; ctx->panel_ranges[2 * t + 1] = (uint8_t)hi;
        }
    }
    // ---- feature path: per observable G[(m, j)][h] = W~[j][(h, m)] in fragment order [N_pad / 16][nb_pad / 8][19][lane = col + 16 g][e]: k = 8 q + 2 g + e ----
    std::vector<std::vector<double>> gfrag_host;
    ctx->feat_ok = ctx->n_obs > 0 && !ctx->any_transform && !getenv("DL_NO_FEATURE_PATH");
    for (int i = 0; i < ctx->n_obs && ctx->feat_ok; ++i) {
        const DlObsDev& d = ctx->obs[i].dev;
        if (d.theory != 3 || d.n_mono != DL_N_MONO || d.n_pass != 0) ctx->feat_ok = false;
        else if (d.eng[0].type == 2 && !dl_emulated_stacked_ok(d)) ctx->feat_ok = false;   // (a stack that does not fit the LDS of the batched kernel: the general path)
    }
    if (ctx->feat_ok) {
        int64_t off = 0;
        for (int i = 0; i < ctx->n_obs; ++i) {
            DlObsDev& d = ctx->obs[i].dev;
            ctx->stk_steps.push_back(0);
            if (d.eng[0].type == 2) {
                // stacked table engine: per column block, group by group: [k / 8][monomial][lane = col + 16 g][e], k = 8 q + 2 g + e < K_g, column of the theory vector col_g + k nm + mo + i
                ctx->any_stacked = true;
                const int njb = ctx->N_pad / 16, H = d.eng[0].widths[d.eng[0].n_layers];
                const double* table = arena.data.data() + ctx->obs[i].off_stk[0];
                int steps = 0;
                for (int gi = 0; gi < d.stk.n_groups; ++gi) steps += (((int)table[gi * DL_STK_REC + 1] - (int)table[gi * DL_STK_REC]) * H + 1 + 7) / 8 * ((int)table[gi * DL_STK_REC + 3] - (int)table[gi * DL_STK_REC + 2]);
                std::vector<double> gf((size_t)njb * steps * 64 * 2, 0.);
                for (int jb = 0; jb < njb; ++jb)
                    for (int gi = 0; gi < d.stk.n_groups; ++gi) {
                        const double* rec = table + (size_t)gi * DL_STK_REC;
                        const int K = ((int)rec[1] - (int)rec[0]) * H + 1, cnt = (int)rec[3] - (int)rec[2], col = (int)rec[4], nm = (int)rec[5], mo = (int)rec[6], kq = (int)rec[7];
                        for (int q = 0; q < (K + 7) / 8; ++q)
                            for (int m = 0; m < cnt; ++m)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int e = 0; e < 2; ++e) {
                                        const int j = jb * 16 + (lane & 15), h = 8 * q + 2 * (lane >> 4) + e;
                                        if (h < K) gf[(((size_t)jb * steps + kq + (size_t)q * cnt + m) * 64 + lane) * 2 + e] = wt_white[(size_t)j * ctx->K_pad + d.col_offset + col + (size_t)h * nm + mo + m];
                                    }
                    }
                ctx->stk_steps.back() = steps;
                d.nb_pad = 8; d.feat_off = off;
                off += 8;
                gfrag_host.push_back(std::move(gf));
                continue;
            }
            d.nb_pad = round_up(d.n_basis, 8);
            d.feat_off = off;
            off += d.nb_pad + (int64_t)17 * DL_FG_MONO_LD;   // records sized for the largest number of rows (1 + DL_MAX_SOLVED); n_var is fixed below
            const int nq = d.nb_pad / 8, njb = ctx->N_pad / 16;
            std::vector<double> gf((size_t)njb * nq * DL_FG_NM * 64 * 2, 0.);
            for (int jb = 0; jb < njb; ++jb)
                for (int q = 0; q < nq; ++q)
                    for (int m = 0; m < DL_FG_NM; ++m)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 2; ++e) {
                                int j = jb * 16 + (lane & 15), h = 8 * q + 2 * (lane >> 4) + e;
                                double v = (h < d.n_basis) ? wt_white[(size_t)j * ctx->K_pad + d.col_offset + (size_t)h * DL_N_MONO + m] : 0.;
                                gf[((((size_t)jb * nq + q) * DL_FG_NM + m) * 64 + lane) * 2 + e] = v;
                            }
            gfrag_host.push_back(std::move(gf));
        }
        ctx->feat_ld = off;
    }
    // ---- analytic marginalisation (likelihoods/base.py:314-413): solved parameters, constant / point-dependent derivative columns ----
    {
        const auto& kind = cfg->I("marg.kind");
        const auto& mprior = cfg->F("marg.prior");
        const auto& mx0 = cfg->F("marg.x0");
        int ns = (int)kind.size();
        if (ns > DL_MAX_SOLVED) return bail("dl_create: at most 16 analytically solved parameters");
        if ((int)mprior.size() != 2 * ns || (int)mx0.size() != ns) return bail("dl_create: marg.prior / marg.x0 sizes do not match marg.kind");
        if (ns > 0 && ctx->any_transform) return bail("dl_create: analytic marginalisation needs a theory linear in the solved parameters (no observable transform)");
        if (ns > 0 && ctx->n_white > 64 * DL_MARG_NJ) return bail("dl_create: analytic marginalisation supports up to 512 data points");
        ctx->n_solved = ns;
        std::memset(&ctx->marg, 0, sizeof(ctx->marg));
        ctx->marg.n_s = ns;
        for (int s_ = 0; s_ < ns; ++s_) {
            ctx->marg.is_marg[s_] = kind[s_] != 0;
            ctx->marg.n_marg += kind[s_] != 0;
            ctx->marg.loc[s_] = mprior[2 * s_];
            ctx->marg.prec[s_] = mprior[2 * s_ + 1];
            ctx->marg.x0[s_] = mx0[s_];
            ctx->marg.var_slot[s_] = -1;
        }
        // point-dependent columns: counter terms (derivative proportional to P_dd,l=0 of the point)
        for (auto& ob : ctx->obs)
            for (int c = 0; c < ob.dev.n_ct; ++c)
                for (int t = 0; t < 2; ++t) {
                    int sidx = ob.marg_ct[c][t];
                    if (sidx < 0) continue;
                    if (sidx >= ns) return bail("dl_create: marg index out of range");
                    if (ctx->marg.var_slot[sidx] < 0) ctx->marg.var_slot[sidx] = ctx->n_var++;
                }
        for (auto& ob : ctx->obs)
            if (ob.dev.theory == 3)
                for (int c = 0; c < DL_N_VPARS; ++c) {
                    int sidx = ob.marg_vp[c];
                    if (sidx < 0) continue;
                    if (sidx >= ns) return bail("dl_create: marg index out of range");
                    if (ctx->marg.var_slot[sidx] < 0) ctx->marg.var_slot[sidx] = ctx->n_var++;
                }
        for (auto& ob : ctx->obs) {
            ob.dev.n_var = ctx->n_var;
            if (ob.dev.theory == 3)
                for (int c = 0; c < DL_N_VPARS; ++c) ob.dev.vp_slot[c] = ob.marg_vp[c] >= 0 ? ctx->marg.var_slot[ob.marg_vp[c]] : -1;
            for (int c = 0; c < ob.dev.n_ct; ++c)
                for (int t = 0; t < 2; ++t) ob.dev.marg_ct_slot[c][t] = ob.marg_ct[c][t] >= 0 ? ctx->marg.var_slot[ob.marg_ct[c][t]] : -1;
        }
        // constant columns: Tt_s = (L^T W) dpower/dx_s with dpower/dsn0 = delta_{l0} / nd (full_shape.py:549), dpower/dsn_c = sn_matrix[:, c] / nd (634)
        std::vector<double> tconst((size_t)std::max(ns, 1) * ctx->N_pad, 0.);
        for (int s_ = 0; s_ < ns; ++s_) {
            std::vector<double> dvec(K, 0.);
            for (auto& ob : ctx->obs) {
                const DlObsDev& od = ob.dev;
                double* dst = &dvec[od.col_offset];
                if (ob.marg_sn0 == s_ && od.theory == 4)   // TNS: sn0 / nd is added to every multipole (full_shape.py:961)
                    for (int idx = 0; idx < od.n_in; ++idx) dst[idx] += 1. / od.nd;
                else if (ob.marg_sn0 == s_ && od.ell0 >= 0)
                    for (int i = 0; i < od.n_kin; ++i) dst[(size_t)od.ell0 * od.n_kin + i] += 1. / od.nd;
                for (int c = 0; c < od.n_pass; ++c)
                    if (ob.marg_pass[c] == s_) dst[od.n_in + c] += 1.;   // pass-through column: derivative = unit vector
                const auto& snm = cfg->F("obs" + std::to_string(&ob - &ctx->obs[0]) + ".sn_matrix");
                for (int c = 0; c < od.n_sn; ++c)
                    if (ob.marg_sn[c] == s_)
                        for (int idx = 0; idx < od.n_in; ++idx) dst[idx] += snm[(size_t)idx * od.n_sn + c] / od.nd;
            }
            for (int i = 0; i < ctx->n_white; ++i) {   // (whitened rows in their residual layout; alignment padding rows are zero)
                double sum = 0.;
                const double* wrow = &wt_white[(size_t)i * ctx->K_pad];
                for (int k = 0; k < K; ++k) sum += wrow[k] * dvec[k];
                tconst[(size_t)s_ * ctx->N_pad + i] = sum;
            }
        }
        if (dl_upload(ctx, &ctx->tconst_dev, tconst)) { dl_destroy(ctx); return 1; }
        ctx->marg.tconst = ctx->tconst_dev;
    }
    // ---- upload ----
    if (dl_upload(ctx, &ctx->arena_dev, arena.data)) { dl_destroy(ctx); return 1; }
    for (auto& gf : gfrag_host) {
        double* dev = nullptr;
        if (dl_upload(ctx, &dev, gf)) { dl_destroy(ctx); return 1; }
        ctx->gfrag_dev.push_back(dev);
    }
    ctx->obs_kernarg.resize(ctx->n_obs);
    for (int i = 0; i < ctx->n_obs; ++i) { ctx->obs[i].rebase(ctx->arena_dev); ctx->obs_kernarg[i] = ctx->obs[i].dev; }
    if (dl_upload(ctx, &ctx->obs_array_dev, ctx->obs_kernarg)) { dl_destroy(ctx); return 1; }
    {   // arrival counters of the fused chi2-GEMM finalize: one per 32-row block of the largest pass that takes that path (self-resetting)
        size_t nbytes = (16384 / 32 + 8 + 64) * sizeof(int32_t);
        if (hipMalloc((void**)&ctx->gemm_counters, nbytes) != hipSuccess || hipMemset(ctx->gemm_counters, 0, nbytes) != hipSuccess) { dl_fail(ctx, "dl_create: counter allocation failed"); dl_destroy(ctx); return 1; }
        ctx->step_ready = ctx->gemm_counters + (16384 / 32 + 8);
    }
    {   // W~ in MFMA fragment order for the chi2 GEMM (window.py:459-473 + likelihoods/base.py:13-17 folded: the constant operand streams straight into registers)
        std::vector<double> frag((size_t)ctx->N_pad * ctx->K_pad);
        const int nks = ctx->K_pad / 4;
        for (int nt = 0; nt < ctx->N_pad / 16; ++nt)
            for (int ks = 0; ks < nks; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    frag[((size_t)nt * nks + ks) * 64 + lane] = wt_white[(size_t)(16 * nt + (lane & 15)) * ctx->K_pad + 4 * ks + (lane >> 4)];
        if (dl_upload(ctx, &ctx->wt_frag_dev, frag)) { dl_destroy(ctx); return 1; }
    }
    if (dl_upload(ctx, &ctx->priors_dev, priors) || dl_upload(ctx, &ctx->wt_white_dev, wt_white) ||
        dl_upload(ctx, &ctx->bias_white_dev, bias_white) || dl_upload(ctx, &ctx->wt_full_dev, wt_full) || dl_upload(ctx, &ctx->bias_full_dev, bias_full) ||
        dl_upload(ctx, &ctx->wh_dev, wh) || dl_upload(ctx, &ctx->bias_wh_dev, bias_wh) || dl_upload(ctx, &ctx->flatdata_dev, flatdata) ||
        dl_upload(ctx, &ctx->transform_dev, transform)) {
        dl_destroy(ctx);
        return 1;
    }
    if (hipDeviceSynchronize() != hipSuccess) { dl_fail(ctx, "dl_create: hipDeviceSynchronize failed"); dl_destroy(ctx); return 1; }   // (null-stream memsets vs non-blocking streams: see dl_reserve)
    *out = ctx;
    return 0;
}

// workspace of the two-launch stacked engine (dl_emu_stacked_split.h): the last hidden layers of observable i's networks, [points, n_networks x H] -- the theory-vector
// workspace, which the feature path does not write (rows of (1 + n_var) K_pad doubles per point: large enough when n_networks x H fits a row)
static double* dl_stk_basis_ws(dl_ctx* ctx, int i) {
    const DlObsDev& d = ctx->obs_kernarg[i];
    if (d.eng[0].type != 2 || ctx->power_ws == nullptr) return nullptr;
    const int64_t ldk = (int64_t)d.stk.n_trunks * d.eng[0].widths[d.eng[0].n_layers];
    return ldk <= (int64_t)(1 + ctx->n_var) * ctx->K_pad ? ctx->power_ws : nullptr;
}

void dl_destroy(dl_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (double* p : {ctx->grad_wtT, ctx->grad_zero, ctx->grad_delta, ctx->grad_y, ctx->grad_phys}) if (p) (void)hipFree(p);
    if (ctx->grad_status) (void)hipFree(ctx->grad_status);
    void* ptrs[] = {ctx->arena_dev, ctx->priors_dev, ctx->wt_white_dev, ctx->wt_frag_dev, ctx->bias_white_dev, ctx->wt_full_dev, ctx->bias_full_dev, ctx->wh_dev,
                    ctx->bias_wh_dev, ctx->flatdata_dev, ctx->transform_dev, ctx->tconst_dev, ctx->power_ws, ctx->delta_ws, ctx->flat_ws, ctx->stencil_ws, ctx->theta_stage, ctx->out_stage,
                    ctx->status_stage, ctx->gemm_counters, ctx->obs_array_dev};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (double* p : ctx->gfrag_dev) if (p) (void)hipFree(p);
    for (auto& ob : ctx->obs) if (ob.tns) { dl_tns_destroy(ob.tns); ob.tns = nullptr; }
    if (ctx->feat_ws) (void)hipFree(ctx->feat_ws);
    for (hipEvent_t e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    if (ctx->host_flag) (void)hipHostFree(ctx->host_flag);
    if (ctx->host_event) (void)hipEventDestroy(ctx->host_event);
    if (ctx->host_stream) (void)hipStreamDestroy(ctx->host_stream);
    delete ctx;
}

int64_t dl_info(const dl_ctx* ctx, const char* key) {
    if (!ctx || !key) return -1;
    std::string k(key);
    if (k == "n_params") return ctx->n_params;
    if (k == "n_data") return ctx->n_data;
    if (k == "n_obs") return ctx->n_obs;
    if (k == "n_in_total") { int64_t t = 0; for (auto& o : ctx->obs) t += o.n_cols(); return t; }
    if (k == "K_pad") return ctx->K_pad;
    if (k == "N_pad") return ctx->N_pad;
    if (k == "n_solved") return ctx->n_solved;
    if (k == "device") return ctx->device;
    for (int i = 0; i < ctx->n_obs; ++i) {
        if (k == "n_in_obs" + std::to_string(i)) return ctx->obs[i].dev.n_in;
        if (k == "n_out_obs" + std::to_string(i)) return ctx->obs[i].n_out;
        if (k == "n_ell_obs" + std::to_string(i)) return ctx->obs[i].dev.n_ell;
        if (k == "n_kin_obs" + std::to_string(i)) return ctx->obs[i].dev.n_kin;
    }
    return -1;
}

static const int64_t DL_CHUNK = 32768;  // points per internal pass: bounds the workspaces to ~0.4 GB at K_pad ~ 1200

static int dl_reserve(dl_ctx* ctx, int64_t B) {
    int64_t need = std::min<int64_t>(B, DL_CHUNK);
    if (need <= ctx->cap) return 0;
    need = std::min<int64_t>(std::max<int64_t>(need, 1024), DL_CHUNK);
    if (ctx->cap > 0) DL_HIP_CHECK(ctx, hipDeviceSynchronize());   // kernels of earlier calls (any stream) may still read the workspaces about to be freed
    for (double** p : {&ctx->power_ws, &ctx->delta_ws, &ctx->flat_ws, &ctx->feat_ws, &ctx->stencil_ws}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    ctx->cap = 0;
    const size_t R = 1 + ctx->n_var;   // rows per point: power + point-dependent derivative rows (analytic marginalisation)
    DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->power_ws, (size_t)need * R * ctx->K_pad * sizeof(double)));
    // residual slabs: split-K partial sums, S * M <= max(2 M, M + 16384 + 64) rows (dl_gemm_tiled_splits)
    DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->delta_ws, std::max<size_t>(2 * (size_t)need * R, (size_t)need * R + 16384 + 2048) * ctx->N_pad * sizeof(double)));
    DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->flat_ws, (size_t)need * ctx->N_pad * sizeof(double)));
    if (ctx->feat_ok) DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->feat_ws, (size_t)need * ctx->feat_ld * sizeof(double)));
    DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->stencil_ws, (size_t)need * ctx->n_params * sizeof(double)));
    // the K padding columns of the power buffer are never written by the theory kernel and must be finite (they meet zeros of the operators); the other
    // workspaces are zeroed too: recycled device memory holds whatever the previous owner left, and 0 x NaN is NaN
    DL_HIP_CHECK(ctx, hipMemset(ctx->power_ws, 0, (size_t)need * R * ctx->K_pad * sizeof(double)));
    DL_HIP_CHECK(ctx, hipMemset(ctx->delta_ws, 0, std::max<size_t>(2 * (size_t)need * R, (size_t)need * R + 16384 + 2048) * ctx->N_pad * sizeof(double)));
    DL_HIP_CHECK(ctx, hipMemset(ctx->flat_ws, 0, (size_t)need * ctx->N_pad * sizeof(double)));
    if (ctx->feat_ok) DL_HIP_CHECK(ctx, hipMemset(ctx->feat_ws, 0, (size_t)need * ctx->feat_ld * sizeof(double)));
    DL_HIP_CHECK(ctx, hipMemset(ctx->stencil_ws, 0, (size_t)need * ctx->n_params * sizeof(double)));
    // the memsets run on the null stream and may still be in flight when this returns; the caller's stream (the private host stream, torch's side streams) is a
    // non-blocking one that does not wait for the null stream: without this synchronisation its kernels could be overtaken by the zeroing of the buffer they write
    // (seen once as a 2.6e-6 error on one row of the first large batch of a context, in a child of tests/test_gpu_switches.py running beside three others)
    DL_HIP_CHECK(ctx, hipDeviceSynchronize());
    ctx->cap = need;
    return 0;
}

static int dl_eval_impl(dl_ctx* ctx, const double* theta_dev, int64_t B, double* loglike_dev, double* logprior_dev, double* flattheory_dev, int32_t* status_dev,
                        double* solved_dev, void* hip_stream, int post_mode, double* hessian_dev = nullptr) {
    if (!ctx) { g_last_error = "dl_eval_batch: null context"; return 1; }
    if (B < 0 || (B > 0 && !theta_dev)) return dl_fail(ctx, "dl_eval_batch: invalid batch");
    if (B == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    dl_prof_events.start = dl_prof_events.stop = nullptr;   // (an earlier call that failed half-way may have left them set)
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_order_streams(ctx, stream)) return 1;
    if (dl_reserve(ctx, B)) return 1;
    const int n = ctx->n_data, P = ctx->n_params, R = 1 + ctx->n_var;
    for (int64_t b0 = 0; b0 < B; b0 += DL_CHUNK) {
        int64_t nb = std::min<int64_t>(DL_CHUNK, B - b0);
        const double* th = theta_dev + (size_t)b0 * P;
        bool prof = ctx->profile && b0 == 0 && (ctx->eval_calls % ctx->prof_every == ctx->prof_every / 2);   // (the middle call of every window: not the first call after a synchronisation)
        hipEvent_t* ev = prof ? &ctx->ev[(size_t)(ctx->prof_calls % dl_ctx::NPOOL) * 6] : nullptr;
        // events attached to the dispatch packets of the launches of phase k (0 theory, 1 GEMM, 2 finalize; dl_kernels.h): with several launches in a phase
        // (one theory launch per observable) the pair holds the LAST one
        const int64_t pc = ctx->prof_calls;
        const int only = !ctx->prof_rotate ? -1 : ctx->prof_only;   // single-kernel mode: the one phase that carries events
        if (prof) ctx->prof_phase_of[(size_t)(pc % dl_ctx::NPOOL)] = (int8_t)only;
        auto prof_phase = [&](int k) {
            const bool on = ev && k >= 0 && (only < 0 || only == k);
            dl_prof_events.start = on ? ev[2 * k] : nullptr;
            dl_prof_events.stop = on ? ev[2 * k + 1] : nullptr;
        };
        bool need_flat = ctx->any_transform || flattheory_dev != nullptr;
        // emulated (separable) theories: the theory kernel writes only the factors (basis, monomial rows), the feature GEMM turns them into residual rows
        const bool feat_path = ctx->feat_ok && !need_flat;
        const bool emu_fused_env = !dl_options().no_emu_fused;   // DL_NO_EMU_FUSED=1: theory kernel -> point records in HBM -> feature GEMM (two launches)
        const bool emu_fused = emu_fused_env || ctx->any_stacked;         // (a stacked table engine has the one-launch form only)
        const int64_t chi2_max_rows = dl_options().chi2_max_rows;   // above: split-K slabs + finalize win (measured: 4096 rows 40 vs 49 us; 1024 rows 18 vs 13 us)
        const bool chi2_path = !feat_path && !ctx->any_transform && ctx->n_solved == 0 && nb <= chi2_max_rows;
        // the chi2 GEMM consumes row block mb (32 points; the LDS-DMA GEMM: 64) on XCD mb % 8: have the theory kernel produce it there (power then waits in that XCD's L2:
        // -0.5 us per 1024 points)
        const int xcd_local = dl_options().xcd_local;   // 0: off, 1: chi2 GEMM path, 2: also the large-batch GEMM
        const bool chi2_fused_env_early = dl_options().chi2_fused;
        const bool chi2_fused_early = chi2_fused_env_early && !ctx->priors_general;
        const int xcd_block = !xcd_local ? 0 : chi2_path ? (chi2_fused_early ? 32 : dl_chi2_gemm_row_tile(nb, ctx->N_pad)) : (xcd_local > 1 && !feat_path && !ctx->any_transform && ctx->n_solved == 0 && ctx->N_pad == 128) ? 64 : 0;
        // the whole step in one launch (dl_step_kernel): plain likelihood of one Kaiser-type observable in its fast instantiation, <= 1024 points in whole groups of 256
        // OFF by default (DL_STEP_KERNEL=1 selects it): measured 31.1 us per 1024-point step against 24.4 us for the three launches -- in-kernel stamps (profiles/r05c_step_stamps.txt):
        // theory of four points under ONE workgroup barrier 14.4 us (four independent workgroups per CU: ~8.5), publish 1.1, wait for the row block 2.0 - 2.5, GEMM + finalize 11.1;
        // even with the theory phase at its stand-alone time the sum is what the three launches take: the step is bound by the lives of its workgroups, not by its launch boundaries
        const bool step_kernel_allowed = dl_options().step_kernel;
        if (step_kernel_allowed && chi2_path && !need_flat && ctx->n_obs == 1 && !ctx->priors_general && ctx->K_pad % 128 == 0 &&
            dl_step_lds_bytes(ctx->obs_kernarg[0], nb, ctx->N_pad) != 0) {
            prof_phase(0);
            dl_launch_step(ctx->obs_kernarg[0], th, P, nb, ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->bias_white_dev, ctx->delta_ws, ctx->K_pad, ctx->K_live,
                           ctx->gemm_counters, ctx->step_ready, 8, ctx->priors_dev, loglike_dev ? loglike_dev + b0 : nullptr, logprior_dev ? logprior_dev + b0 : nullptr,
                           status_dev ? status_dev + b0 : nullptr, post_mode, stream, ctx->panel_ranges.empty() ? nullptr : ctx->panel_ranges.data());
            prof_phase(-1);
            if (prof) ctx->prof_calls++;
            continue;
        }
        prof_phase(0);
        if (!(feat_path && emu_fused))
            dl_launch_fullshape(ctx->obs_kernarg.data(), ctx->n_obs, th, P, nb, ctx->power_ws, ctx->K_pad, nullptr, 0, stream, feat_path ? ctx->feat_ws : nullptr, ctx->feat_ld,
                                xcd_block, ctx->obs_array_dev);
        prof_phase(1);
        int n_slabs = 1, cps = 0;
        int64_t slab_stride = 0;
        const double* fin_bias = nullptr;   // the direct GEMM (transform path) adds its bias itself
        if (need_flat) {
            // flattheory = W . power + bias (window.py:459-473), then optional cubic transform (power_spectrum.py:402-404)
            dl_launch_window_gemm(ctx->power_ws, (int64_t)R * ctx->K_pad, ctx->wt_full_dev, ctx->K_pad, ctx->bias_full_dev, ctx->flat_ws, ctx->N_pad, nb, ctx->N_pad,
                                  ctx->N_pad, ctx->K_pad, 1, stream);   // row 0 of each point only (lda skips the derivative rows)
            if (ctx->any_transform) dl_launch_transform(ctx->flat_ws, ctx->N_pad, ctx->flatdata_dev, ctx->transform_dev, n, nb, stream);
            if (flattheory_dev)
                DL_HIP_CHECK(ctx, hipMemcpy2DAsync(flattheory_dev + (size_t)b0 * n, (size_t)n * sizeof(double), ctx->flat_ws, (size_t)ctx->N_pad * sizeof(double),
                                                   (size_t)n * sizeof(double), (size_t)nb, hipMemcpyDeviceToDevice, stream));
        }
        // plain likelihood: chi2 is additive over the columns of the whitened residual -> column-split GEMM that emits partial chi2 only
        // marginalised fits on one emulated observable: Gram-matrix epilogue in the fused kernel -- the residual rows never reach memory (DL_NO_GRAM_EPILOGUE=1: rows + Gram in the finalize)
        bool gram_done = false, finalized_in_kernel = false, stacked_rows_done = false;
        const bool gram_epilogue = !dl_options().no_gram_epilogue;
        // (also without solved parameters: X is the residual row alone, chi2 = G[0][0] and the finalize -- priors, status -- runs in the kernel's tail: one launch instead of two)
        if (feat_path && emu_fused && gram_epilogue && ctx->n_obs == 1 && ctx->N_pad == 128) {
            DlGramFinalize fin = {ctx->priors_dev, loglike_dev ? loglike_dev + b0 : nullptr, logprior_dev ? logprior_dev + b0 : nullptr, status_dev ? status_dev + b0 : nullptr,
                                  solved_dev ? solved_dev + (size_t)b0 * ctx->n_solved : nullptr, hessian_dev ? hessian_dev + (size_t)b0 * ctx->n_solved * ctx->n_solved : nullptr,
                                  post_mode, false};
            if (ctx->stk_steps[0]) {   // stacked table engine: the same finalize in the tail of dl_emulated_stacked_kernel when the rows of X fit its LDS
                prof_phase(0);      // (the chains of the two-launch form carry the events of the theory phase, the feature GEMMs those of the GEMM phase)
                const bool chains_done = dl_launch_stk_chains(ctx->obs_kernarg[0], th, P, nb, dl_stk_basis_ws(ctx, 0), stream);
                prof_phase(1);
                dl_launch_emulated_stacked(ctx->obs_kernarg[0], th, P, nb, ctx->gfrag_dev[0], ctx->delta_ws, ctx->N_pad, ctx->N_pad, 0, ctx->stk_steps[0], stream, &fin, ctx->bias_white_dev,
                                           &ctx->marg, ctx->n_white, dl_stk_basis_ws(ctx, 0), chains_done);
                gram_done = fin.done;          // (launched either way: with fin.done the outputs are written ...
                stacked_rows_done = !fin.done; //  ... without it the residual rows of the observable are in delta_ws: the general finalize follows)
            } else gram_done = dl_launch_emulated_feature_gram(ctx->obs_kernarg[0], th, P, nb, ctx->gfrag_dev[0], ctx->bias_white_dev, ctx->marg, ctx->n_white, ctx->delta_ws, stream, &fin);
            if (gram_done) fin_bias = ctx->bias_white_dev;
            finalized_in_kernel = gram_done && fin.done;
        }
        if (feat_path && !gram_done) {
            for (int i = 0; i < ctx->n_obs; ++i) {
                if (i == 0 && stacked_rows_done) continue;
                if (ctx->stk_steps[i]) dl_launch_emulated_stacked(ctx->obs_kernarg[i], th, P, nb, ctx->gfrag_dev[i], ctx->delta_ws, ctx->N_pad, ctx->N_pad, i > 0, ctx->stk_steps[i], stream, nullptr, nullptr, nullptr, 0, dl_stk_basis_ws(ctx, i));
                else if (emu_fused) dl_launch_emulated_feature(ctx->obs_kernarg[i], th, P, nb, ctx->gfrag_dev[i], ctx->delta_ws, ctx->N_pad, ctx->N_pad, i > 0, stream);
                else dl_launch_feature_gemm(ctx->feat_ws, ctx->feat_ld, ctx->obs_kernarg[i].feat_off, ctx->obs_kernarg[i].nb_pad, R, ctx->gfrag_dev[i], ctx->delta_ws, ctx->N_pad,
                                            ctx->N_pad, nb, i > 0, stream);
            }
            fin_bias = ctx->bias_white_dev;
        }
        // larger plain batches: the LDS-DMA split-K GEMM with the same partial-chi2 epilogue (no residual slab is written or read)
        int cps_probe = 0;
        const bool chi2_big_allowed = !dl_options().no_chi2_big;
        const bool chi2_big = !feat_path && !ctx->any_transform && ctx->n_solved == 0 && !chi2_path && ctx->K_pad % 32 == 0 && chi2_big_allowed &&
                              dl_gemm_tiled_splits(nb, ctx->N_pad, ctx->K_pad, &cps_probe) == 1;   // enough row tiles to fill the chip without splitting K
        int part_tiles = ctx->N_pad / 16;
        // DL_CHI2_FUSED=1: finalize inside the GEMM's last-arriving workgroups.  Off by default: measured 19.1 us (GEMM 17.1) against 17.5 us for GEMM + the
        // separate 1024-thread finalize launch -- the device-scope counter round trip and the dependent tail cost more than the launch they save.
        const bool chi2_fused_env = dl_options().chi2_fused;
        const bool chi2_fused = chi2_fused_env && !ctx->priors_general;   // (the experimental fused finalize of the GEMM knows uniform / norm priors only)
        if (chi2_path) {
            if (nb > 16384) { prof_phase(-1); return dl_fail(ctx, "dl_eval_batch: DL_CHI2_GEMM_MAX above 16384 rows"); }
            dl_launch_chi2_gemm(ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->bias_white_dev, ctx->delta_ws, nb, ctx->N_pad, ctx->K_pad,
                                chi2_fused ? ctx->gemm_counters : nullptr, th, P, ctx->priors_dev, loglike_dev ? loglike_dev + b0 : nullptr,
                                logprior_dev ? logprior_dev + b0 : nullptr, status_dev ? status_dev + b0 : nullptr, post_mode, stream,
                                ctx->panel_ranges.empty() ? nullptr : ctx->panel_ranges.data(), ctx->K_live, nullptr, 0, ctx->wt_frag_dev);
        } else if (chi2_big) {
            dl_launch_window_gemm_dma_chi2(ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->bias_white_dev, ctx->delta_ws, nb, ctx->N_pad, ctx->K_pad, stream, ctx->n_white);
            part_tiles = dl_gemm_dma_chi2_parts(ctx->N_pad);
        } else if (feat_path) {
            // residual rows already in delta_ws (one slab, bias added by the finalize kernels)
        } else if (ctx->any_transform) {
            // dtilde = L^T (flattheory - flatdata)
            dl_launch_window_gemm(ctx->flat_ws, ctx->N_pad, ctx->wh_dev, ctx->N_pad, ctx->bias_wh_dev, ctx->delta_ws, ctx->N_pad, nb, ctx->N_pad, ctx->N_pad, ctx->N_pad,
                                  1, stream);
        } else {
            // dtilde = (L^T W) . power + L^T (bias - flatdata): window convolution and precision folded in one fp64 MFMA GEMM (split-K slabs, bias added by finalize)
            n_slabs = dl_gemm_tiled_splits(nb * R, ctx->N_pad, ctx->K_pad, &cps);
            slab_stride = (int64_t)nb * R * ctx->N_pad;
            fin_bias = ctx->bias_white_dev;
            dl_launch_window_gemm_tiled(ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->delta_ws, slab_stride, ctx->N_pad, nb * R, ctx->N_pad, ctx->K_pad, n_slabs, cps,
                                        stream, ctx->n_white);
        }
        prof_phase(2);
        if (chi2_path && chi2_fused) {
            // finalize fused into the GEMM
        } else if (finalized_in_kernel) {
            // marginalised finalize in the tail of the fused emulator / feature-GEMM kernel
        } else if (chi2_path || chi2_big)
            dl_launch_finalize_part(ctx->delta_ws, part_tiles, th, P, ctx->priors_dev, nb, loglike_dev ? loglike_dev + b0 : nullptr, logprior_dev ? logprior_dev + b0 : nullptr,
                                    status_dev ? status_dev + b0 : nullptr, post_mode, stream);
        else if (ctx->n_solved > 0)
            dl_launch_finalize_marg(ctx->delta_ws, ctx->N_pad, ctx->n_white, R, n_slabs, slab_stride, fin_bias, ctx->marg, th, P, ctx->priors_dev, nb, loglike_dev ? loglike_dev + b0 : nullptr,
                                    logprior_dev ? logprior_dev + b0 : nullptr, status_dev ? status_dev + b0 : nullptr,
                                    solved_dev ? solved_dev + (size_t)b0 * ctx->n_solved : nullptr,
                                    hessian_dev ? hessian_dev + (size_t)b0 * ctx->n_solved * ctx->n_solved : nullptr, post_mode, stream, feat_path && ctx->n_obs == 1,
                                    gram_done ? ctx->delta_ws : nullptr);
        else
            dl_launch_finalize(ctx->delta_ws, ctx->N_pad, ctx->n_white, n_slabs, slab_stride, fin_bias, th, P, ctx->priors_dev, nb, loglike_dev ? loglike_dev + b0 : nullptr, logprior_dev ? logprior_dev + b0 : nullptr,
                               status_dev ? status_dev + b0 : nullptr, post_mode, stream);
        prof_phase(-1);
        if (prof) ctx->prof_calls++;
    }
    if (ctx->profile) ctx->eval_calls++;
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

int dl_eval_batch(dl_ctx* ctx, const double* theta_dev, int64_t B, double* loglike_dev, double* logprior_dev, double* flattheory_dev, int32_t* status_dev,
                  double* solved_dev, void* hip_stream) {
    return dl_eval_impl(ctx, theta_dev, B, loglike_dev, logprior_dev, flattheory_dev, status_dev, solved_dev, hip_stream, 0);
}

int dl_eval_batch_derived(dl_ctx* ctx, const double* theta_dev, int64_t B, double* loglike_dev, double* logprior_dev, int32_t* status_dev, double* solved_dev,
                          double* hessian_dev, void* hip_stream) {
    if (ctx && hessian_dev && ctx->n_solved == 0) return dl_fail(ctx, "dl_eval_batch_derived: no analytically solved parameter in this likelihood");
    return dl_eval_impl(ctx, theta_dev, B, loglike_dev, logprior_dev, nullptr, status_dev, solved_dev, hip_stream, 0, hessian_dev);
}

int dl_eval_logposterior(dl_ctx* ctx, const double* theta_dev, int64_t B, double* logposterior_dev, int32_t* status_dev, void* hip_stream) {
    if (ctx && B > 0 && !logposterior_dev) return dl_fail(ctx, "dl_eval_logposterior: null output");
    return dl_eval_impl(ctx, theta_dev, B, logposterior_dev, nullptr, nullptr, status_dev, nullptr, hip_stream, 1);
}

int dl_eval_fisher(dl_ctx* ctx, const double* centers_dev, const double* steps_dev, int64_t B, double* hessian_dev, double* gradient_dev, double* offset_dev, void* hip_stream) {
    if (!ctx) { g_last_error = "dl_eval_fisher: null context"; return 1; }
    if (B < 0 || (B > 0 && (!centers_dev || !steps_dev))) return dl_fail(ctx, "dl_eval_fisher: invalid argument");
    if (ctx->n_solved > 0) return dl_fail(ctx, "dl_eval_fisher: the context has analytically solved parameters: create it with these parameters varied (the reference does the same, fisher.py:688-695)");
    const int P = ctx->n_params, n = ctx->n_data, S = 1 + 2 * P;
    if (P > 31) return dl_fail(ctx, "dl_eval_fisher: at most 31 varied parameters");
    if (dl_fisher_waves(ctx->n_white, P, nullptr, nullptr) < 1) return dl_fail(ctx, "dl_eval_fisher: too many parameters for the LDS-resident Gram product");
    if (B == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    dl_prof_events.start = dl_prof_events.stop = nullptr;
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_order_streams(ctx, stream)) return 1;
    const int64_t per_pass = std::max<int64_t>(1, DL_CHUNK / S);   // centres per internal pass (stencil rows <= DL_CHUNK)
    if (dl_reserve(ctx, std::min<int64_t>(B, per_pass) * S)) return 1;
    for (int64_t b0 = 0; b0 < B; b0 += per_pass) {
        const int64_t nc = std::min<int64_t>(per_pass, B - b0), nb = nc * S;
        const double* steps = steps_dev + (size_t)b0 * P * 2;
        dl_launch_fisher_stencil(centers_dev + (size_t)b0 * P, steps, P, nc, ctx->stencil_ws, stream);
        const double* th = ctx->stencil_ws;
        int n_slabs = 1, cps = 0;
        int64_t slab_stride = 0;
        const double* bias = nullptr;
        const bool emu_fused_env = !dl_options().no_emu_fused;
        const bool emu_fused = emu_fused_env || ctx->any_stacked;
        if (ctx->feat_ok) {
            // emulated (separable) theories: residual rows straight from the feature GEMM
            if (!emu_fused) dl_launch_fullshape(ctx->obs_kernarg.data(), ctx->n_obs, th, P, nb, ctx->power_ws, ctx->K_pad, nullptr, 0, stream, ctx->feat_ws, ctx->feat_ld, 0);
            for (int i = 0; i < ctx->n_obs; ++i) {
                if (ctx->stk_steps[i]) dl_launch_emulated_stacked(ctx->obs_kernarg[i], th, P, nb, ctx->gfrag_dev[i], ctx->delta_ws, ctx->N_pad, ctx->N_pad, i > 0, ctx->stk_steps[i], stream, nullptr, nullptr, nullptr, 0, dl_stk_basis_ws(ctx, i));
                else if (emu_fused) dl_launch_emulated_feature(ctx->obs_kernarg[i], th, P, nb, ctx->gfrag_dev[i], ctx->delta_ws, ctx->N_pad, ctx->N_pad, i > 0, stream);
                else dl_launch_feature_gemm(ctx->feat_ws, ctx->feat_ld, ctx->obs_kernarg[i].feat_off, ctx->obs_kernarg[i].nb_pad, 1, ctx->gfrag_dev[i], ctx->delta_ws, ctx->N_pad,
                                            ctx->N_pad, nb, i > 0, stream);
            }
            bias = ctx->bias_white_dev;
        } else {
            dl_launch_fullshape(ctx->obs_kernarg.data(), ctx->n_obs, th, P, nb, ctx->power_ws, ctx->K_pad, nullptr, 0, stream, nullptr, 0, 0, ctx->obs_array_dev);
            if (ctx->any_transform) {
                // flattheory -> observable transform -> whitened residual (bias added by the direct GEMM)
                dl_launch_window_gemm(ctx->power_ws, ctx->K_pad, ctx->wt_full_dev, ctx->K_pad, ctx->bias_full_dev, ctx->flat_ws, ctx->N_pad, nb, ctx->N_pad, ctx->N_pad, ctx->K_pad, 1, stream);
                dl_launch_transform(ctx->flat_ws, ctx->N_pad, ctx->flatdata_dev, ctx->transform_dev, n, nb, stream);
                dl_launch_window_gemm(ctx->flat_ws, ctx->N_pad, ctx->wh_dev, ctx->N_pad, ctx->bias_wh_dev, ctx->delta_ws, ctx->N_pad, nb, ctx->N_pad, ctx->N_pad, ctx->N_pad, 1, stream);
            } else {
                n_slabs = dl_gemm_tiled_splits(nb, ctx->N_pad, ctx->K_pad, &cps);
                slab_stride = (int64_t)nb * ctx->N_pad;
                bias = ctx->bias_white_dev;
                dl_launch_window_gemm_tiled(ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->delta_ws, slab_stride, ctx->N_pad, nb, ctx->N_pad, ctx->K_pad, n_slabs, cps, stream, ctx->n_white);
            }
        }
        dl_launch_fisher(ctx->delta_ws, ctx->N_pad, ctx->n_white, n_slabs, slab_stride, bias, steps, P, nc, hessian_dev ? hessian_dev + (size_t)b0 * P * P : nullptr,
                         gradient_dev ? gradient_dev + (size_t)b0 * P : nullptr, offset_dev ? offset_dev + b0 : nullptr, stream);
    }
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

// log-posterior and its gradient: theory -> residual rows d~ (direct GEMM) -> chi2 + priors, Y = -d~ W~ (second GEMM) -> gradient workgroups -> chain rule.
// Returns 2 (nothing launched) when the context is outside the analytic gradient's scope (dl_fullshape_grad.h): the caller differentiates numerically.
int dl_eval_logposterior_grad(dl_ctx* ctx, const double* theta_dev, int64_t B, double* logposterior_dev, double* grad_dev, int32_t* status_dev, void* hip_stream) {
    if (!ctx) { g_last_error = "dl_eval_logposterior_grad: null context"; return 1; }
    if (B < 0 || (B > 0 && (!theta_dev || !logposterior_dev || !grad_dev))) return dl_fail(ctx, "dl_eval_logposterior_grad: invalid argument");
    if (ctx->feat_ok || ctx->any_transform || ctx->n_solved != 0 || ctx->priors_general || !dl_grad_applicable(ctx->obs_kernarg.data(), ctx->n_obs)) return 2;
    if (B == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    dl_prof_events.start = dl_prof_events.stop = nullptr;
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_order_streams(ctx, stream)) return 1;
    // per pass: at most 2048 points (the chi2 GEMM with the residual output: partial chi2 for the log-posterior AND the rows d~ for the gradient, in one launch)
    const int64_t per_pass = 2048;
    if (dl_reserve(ctx, std::min<int64_t>(B, per_pass))) return 1;
    const int P = ctx->n_params, Np = ctx->N_pad, Kp = ctx->K_pad;
    if (!ctx->grad_wtT) {
        std::vector<double> w((size_t)Np * Kp), wt((size_t)Kp * Np);
        DL_HIP_CHECK(ctx, hipMemcpy(w.data(), ctx->wt_white_dev, w.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int j = 0; j < Np; ++j) for (int k = 0; k < Kp; ++k) wt[(size_t)k * Np + j] = -w[(size_t)j * Kp + k];
        DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grad_wtT, wt.size() * sizeof(double)));
        DL_HIP_CHECK(ctx, hipMemcpy(ctx->grad_wtT, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice));
        DL_HIP_CHECK(ctx, hipDeviceSynchronize());
    }
    const int64_t need = std::min<int64_t>(B, per_pass);
    if (need > ctx->grad_cap) {
        if (ctx->grad_cap > 0) DL_HIP_CHECK(ctx, hipDeviceSynchronize());
        for (double** p : {&ctx->grad_delta, &ctx->grad_y, &ctx->grad_phys}) if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (ctx->grad_status) { (void)hipFree(ctx->grad_status); ctx->grad_status = nullptr; }
        ctx->grad_cap = 0;
        const int64_t cap = std::max<int64_t>((need + 63) / 64 * 64, 256);     // (whole 64-row tiles of the second GEMM)
        DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grad_delta, (size_t)cap * Np * sizeof(double)));
        DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grad_y, (size_t)cap * Kp * sizeof(double)));
        DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grad_phys, (size_t)cap * ctx->n_obs * 8 * sizeof(double)));
        DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grad_status, (size_t)cap * sizeof(int32_t)));
        DL_HIP_CHECK(ctx, hipMemset(ctx->grad_delta, 0, (size_t)cap * Np * sizeof(double)));
        DL_HIP_CHECK(ctx, hipDeviceSynchronize());
        ctx->grad_cap = cap;
    }
    const int xcd_local = dl_options().xcd_local;
    for (int64_t b0 = 0; b0 < B; b0 += per_pass) {
        const int64_t nb = std::min<int64_t>(per_pass, B - b0);
        const double* th = theta_dev + (size_t)b0 * P;
        int32_t* st = status_dev ? status_dev + b0 : ctx->grad_status;      // (the gradient's finalize needs the status whether the caller wants it or not)
        dl_launch_fullshape(ctx->obs_kernarg.data(), ctx->n_obs, th, P, nb, ctx->power_ws, ctx->K_pad, nullptr, 0, stream, nullptr, 0, xcd_local ? dl_chi2_gemm_row_tile(nb, Np) : 0, ctx->obs_array_dev);
        dl_launch_chi2_gemm(ctx->power_ws, Kp, ctx->wt_white_dev, Kp, ctx->bias_white_dev, ctx->delta_ws, nb, Np, Kp, nullptr, th, P, ctx->priors_dev, nullptr, nullptr, nullptr, 1, stream,
                            ctx->panel_ranges.empty() ? nullptr : ctx->panel_ranges.data(), ctx->K_live, ctx->grad_delta, Np, ctx->wt_frag_dev);
        dl_launch_finalize_part(ctx->delta_ws, Np / 16, th, P, ctx->priors_dev, nb, logposterior_dev + b0, nullptr, st, 1, stream);
        // Y = -d~ W~: [nb, N_pad] x [N_pad, K_pad] through the LDS-DMA tiled GEMM, one split (K = N_pad: 4 panels), no bias
        dl_launch_window_gemm_tiled(ctx->grad_delta, Np, ctx->grad_wtT, Np, ctx->grad_y, 0, Kp, nb, Kp, Np, 1, Np / 16, stream, 0);
        dl_launch_fullshape_grad(ctx->obs_kernarg.data(), ctx->n_obs, ctx->obs_array_dev, th, P, nb, ctx->grad_y, Kp, ctx->grad_phys, ctx->priors_dev, st, grad_dev + (size_t)b0 * P, stream);
    }
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

int dl_eval_theory(dl_ctx* ctx, const double* theta_dev, int64_t B, int32_t iobs, double* power_dev, double* tables_dev, void* hip_stream) {
    if (!ctx) { g_last_error = "dl_eval_theory: null context"; return 1; }
    if (iobs < 0 || iobs >= ctx->n_obs) return dl_fail(ctx, "dl_eval_theory: observable index out of range");
    if (B < 0 || (B > 0 && (!theta_dev || !power_dev))) return dl_fail(ctx, "dl_eval_theory: invalid argument");
    if (B == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // launch only observable iobs, writing rows of n_in doubles directly into the caller's buffers
    DlObsDev tmp = ctx->obs_kernarg[iobs];
    tmp.col_offset = 0;
    tmp.n_pass = 0;   // rows of exactly n_in doubles
    tmp.n_var = 0;    // no derivative rows: the slots that would address them are cleared (they index LDS sized by n_var)
    for (int c = 0; c < DL_N_VPARS; ++c) tmp.vp_slot[c] = -1;
    for (int c = 0; c < DL_MAX_EFT; ++c) tmp.marg_ct_slot[c][0] = tmp.marg_ct_slot[c][1] = -1;
    if (dl_order_streams(ctx, stream)) return 1;
    dl_launch_fullshape(&tmp, 1, theta_dev, ctx->n_params, B, power_dev, tmp.n_in, tables_dev, 3 * (int64_t)tmp.n_in, stream);
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

int dl_eval_tns_tables(dl_ctx* ctx, const double* theta_dev, int64_t B, int32_t iobs, double* tables_dev, void* hip_stream) {
    if (!ctx) { g_last_error = "dl_eval_tns_tables: null context"; return 1; }
    if (iobs < 0 || iobs >= ctx->n_obs) return dl_fail(ctx, "dl_eval_tns_tables: observable index out of range");
    if (ctx->obs[iobs].dev.theory != 4) return dl_fail(ctx, "dl_eval_tns_tables: the observable's theory is not the TNS model");
    if (B < 0 || (B > 0 && (!theta_dev || !tables_dev))) return dl_fail(ctx, "dl_eval_tns_tables: invalid argument");
    if (B == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_order_streams(ctx, stream)) return 1;
    if (dl_tns_tables(ctx->obs_kernarg[iobs], theta_dev, ctx->n_params, B, tables_dev, stream)) return dl_fail(ctx, std::string("dl_eval_tns_tables: ") + dl_last_error(nullptr));
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

static int dl_stage_reserve(dl_ctx* ctx, int64_t B) {
    if (!ctx->host_stream) DL_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->host_stream, hipStreamNonBlocking));
    if (B <= ctx->stage_cap) return 0;
    for (void* p : {(void*)ctx->theta_stage, (void*)ctx->out_stage}) if (p) (void)hipFree(p);
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    ctx->theta_stage = ctx->out_stage = ctx->host_stage = nullptr; ctx->status_stage = nullptr; ctx->stage_cap = 0;
    int64_t cap = std::max<int64_t>(B, 64);
    DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->theta_stage, (size_t)cap * ctx->n_params * sizeof(double)));
    DL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->out_stage, (size_t)cap * 3 * sizeof(double)));   // per call: loglike[B] | logprior[B] | status[B] (int32), one block
    DL_HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->host_stage, (size_t)cap * (ctx->n_params + 3) * sizeof(double), hipHostMallocMapped));
    DL_HIP_CHECK(ctx, hipHostGetDevicePointer((void**)&ctx->host_stage_dev, ctx->host_stage, 0));
    if (!ctx->host_flag) {
        DL_HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->host_flag, 64, hipHostMallocMapped));
        DL_HIP_CHECK(ctx, hipHostGetDevicePointer((void**)&ctx->host_flag_dev, ctx->host_flag, 0));
        *ctx->host_flag = 0;
        DL_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->host_event, hipEventDisableTiming));
    }
    ctx->stage_cap = cap;
    return 0;
}

// Completion flag of the mapped *_host path: launched after the last kernel of the call on the same (in-order) stream, so every result -- written by the kernels
// straight into the caller-visible pinned buffer -- is out before the sequence number is; the host spins on it instead of paying a stream synchronisation.
__global__ void dl_host_flag_kernel(uint64_t* flag, uint64_t seq) {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// DL_HOST_MODE: 0 staged copies + stream synchronisation (the first version), 1 mapped buffers + stream synchronisation, 2 mapped + event polling,
// 3 (default) mapped + completion flag (+ a stream query at the start of one call in eight), 4 the flag alone.  Tried and dropped (profiles/r04a_host_call_modes.txt):
// a stream synchronisation after the flag (46 us median: the runtime's own wait path), polling hipStreamQuery (41 us, long tail)
static int dl_host_mode() { return dl_options().host_mode >= 0 ? dl_options().host_mode : 3; }   // (read once: dl_options_refresh re-reads DL_HOST_MODE)

// Host-pointer evaluation (what the reference-side binding and any unmodified desilike sampler call: samplers/base.py:144-200, samplers/emcee.py:69).
// Default (mode 3): theta is copied into a pinned, device-mapped buffer which the kernels read in place; the finalize kernel writes loglike | logprior | status
// straight into the same buffer; a one-thread kernel then publishes the call's sequence number and the host spins on it -- no copy engine, no DtoH, no
// hipStreamSynchronize (the staged version: one async copy in, one out, one synchronisation: 75 - 240 us per 256-point call depending on the box).
// flattheory / solved outputs keep the staged path (large, rare).
static int dl_eval_host_impl(dl_ctx* ctx, const double* theta, int64_t B, double* loglike, double* logprior, double* flattheory, int32_t* status, double* solved,
                             double* logposterior) {
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_stage_reserve(ctx, B)) return 1;
    hipStream_t stream = ctx->host_stream;
    const int P = ctx->n_params;
    int mode = dl_host_mode();
    // the flag path never tells the runtime that a call has finished: left alone it retires ~1000 dispatches at once every ~250 calls (eight calls in a row 15 - 30 us
    // slower).  A stream query at the start of a call -- the previous one is long finished -- lets it retire them; a query costs ~6 us plus what it retires (one call
    // in 32: those calls +20 us, p99 = 1.57 x median; every call: median +7 us): one call in 8 pays it (p99 / median, measured: profiles/r04*_host_call*)
    if (mode == 3 && ctx->host_seq && (ctx->host_seq & 7) == 0) (void)hipStreamQuery(stream);
    double *flat_dev = nullptr, *solved_dev = nullptr;
    if (flattheory) DL_HIP_CHECK(ctx, hipMalloc((void**)&flat_dev, (size_t)B * ctx->n_data * sizeof(double)));
    if (solved && ctx->n_solved > 0 && hipMalloc((void**)&solved_dev, (size_t)B * ctx->n_solved * sizeof(double)) != hipSuccess) {
        if (flat_dev) (void)hipFree(flat_dev);
        return dl_fail(ctx, "dl_eval_batch_host: allocation of the solved-parameter block failed");
    }
    if (flat_dev || solved_dev) mode = 0;
    double* host_in = ctx->host_stage;
    double* host_out = ctx->host_stage + (size_t)ctx->stage_cap * P;
    std::copy(theta, theta + (size_t)B * P, host_in);
    const double* theta_dev = ctx->theta_stage;
    double* out_dev = ctx->out_stage;
    // theta: read in place from the mapped buffer by every kernel that needs it -- up to 4096 points; above, each of those reads crosses PCIe again and one
    // asynchronous copy to device memory is cheaper (8192 points: 178 -> 173 us, 32768: 590 -> 540 us; 1024: 45 against 59 us the other way)
    const bool theta_mapped = mode != 0 && B <= 4096;
    if (theta_mapped) theta_dev = ctx->host_stage_dev;
    else if (hipMemcpyAsync(ctx->theta_stage, host_in, (size_t)B * P * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) {
        if (flat_dev) (void)hipFree(flat_dev);
        if (solved_dev) (void)hipFree(solved_dev);
        return dl_fail(ctx, "dl_eval_batch_host: copy of the parameter block failed");
    }
    if (mode != 0) out_dev = ctx->host_stage_dev + (size_t)ctx->stage_cap * P;
    double* ll_dev = out_dev;
    double* lp_dev = out_dev + B;
    int32_t* st_dev = reinterpret_cast<int32_t*>(out_dev + 2 * B);
    int rc;
    if (logposterior) rc = dl_eval_logposterior(ctx, theta_dev, B, ll_dev, st_dev, stream);
    else rc = dl_eval_batch(ctx, theta_dev, B, ll_dev, lp_dev, flat_dev, st_dev, solved_dev, stream);
    if (rc == 0) {
        hipError_t e = hipSuccess;
        if (mode == 0) {
            const size_t out_bytes = (size_t)B * 2 * sizeof(double) + (size_t)B * sizeof(int32_t);
            e = hipMemcpyAsync(host_out, ctx->out_stage, out_bytes, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess && flattheory) e = hipMemcpyAsync(flattheory, flat_dev, (size_t)B * ctx->n_data * sizeof(double), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess && solved_dev) e = hipMemcpyAsync(solved, solved_dev, (size_t)B * ctx->n_solved * sizeof(double), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
        } else if (mode == 1) {
            e = hipStreamSynchronize(stream);
        } else if (mode == 2) {
            e = hipEventRecord(ctx->host_event, stream);
            while (e == hipSuccess && (e = hipEventQuery(ctx->host_event)) == hipErrorNotReady) e = hipSuccess;

        } else {   // modes 3, 4
            const uint64_t seq = ++ctx->host_seq;
            dl_host_flag_kernel<<<1, 1, 0, stream>>>(ctx->host_flag_dev, seq);
            e = hipGetLastError();
            volatile uint64_t* flag = ctx->host_flag;
            // spin on the flag for a bounded time (small batches finish within tens of microseconds: the spin is what makes the call 34 us instead of 46), then hand the
            // wait to the runtime -- a large batch must not burn a host core that the sampler's own threads (or another rank) could use
            const auto spin_start = std::chrono::steady_clock::now();
            for (uint64_t spins = 0; e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                std::this_thread::yield();
#endif
                if ((spins & 0x3ff) == 0x3ff && std::chrono::steady_clock::now() - spin_start > std::chrono::microseconds(200)) {
                    e = hipStreamSynchronize(stream);    // (also the way out when a failed kernel never writes the flag)
                    if (e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) e = hipErrorUnknown;
                    break;
                }
            }

        }
        if (e != hipSuccess) rc = dl_fail(ctx, std::string("dl_eval_batch_host: ") + hipGetErrorString(e));
    }
    if (rc == 0) {
        if (logposterior) std::copy(host_out, host_out + B, logposterior);
        if (loglike) std::copy(host_out, host_out + B, loglike);
        if (logprior) std::copy(host_out + B, host_out + 2 * B, logprior);
        if (status) { const int32_t* st = reinterpret_cast<const int32_t*>(host_out + 2 * B); std::copy(st, st + B, status); }
    }
    if (flat_dev) (void)hipFree(flat_dev);
    if (solved_dev) (void)hipFree(solved_dev);
    return rc;
}

int dl_eval_batch_host(dl_ctx* ctx, const double* theta, int64_t B, double* loglike, double* logprior, double* flattheory, int32_t* status, double* solved) {
    if (!ctx) { g_last_error = "dl_eval_batch_host: null context"; return 1; }
    if (B < 0 || (B > 0 && !theta)) return dl_fail(ctx, "dl_eval_batch_host: invalid batch");
    if (B == 0) return 0;
    return dl_eval_host_impl(ctx, theta, B, loglike, logprior, flattheory, status, solved, nullptr);
}

int dl_eval_logposterior_host(dl_ctx* ctx, const double* theta, int64_t B, double* logposterior, int32_t* status) {
    if (!ctx) { g_last_error = "dl_eval_logposterior_host: null context"; return 1; }
    if (B < 0 || (B > 0 && (!theta || !logposterior))) return dl_fail(ctx, "dl_eval_logposterior_host: invalid argument");
    if (B == 0) return 0;
    return dl_eval_host_impl(ctx, theta, B, nullptr, nullptr, nullptr, status, nullptr, logposterior);
}

int dl_eval_theory_host(dl_ctx* ctx, const double* theta, int64_t B, int32_t iobs, double* power, double* tables) {
    if (!ctx) { g_last_error = "dl_eval_theory_host: null context"; return 1; }
    if (iobs < 0 || iobs >= ctx->n_obs) return dl_fail(ctx, "dl_eval_theory_host: observable index out of range");
    if (B <= 0 || !theta || !power) return dl_fail(ctx, "dl_eval_theory_host: invalid argument");
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_stage_reserve(ctx, B)) return 1;
    hipStream_t stream = ctx->host_stream;
    const int n_in = ctx->obs[iobs].dev.n_in;
    double *pdev = nullptr, *tdev = nullptr;
    hipError_t e = hipMalloc((void**)&pdev, (size_t)B * n_in * sizeof(double));
    if (e == hipSuccess && tables) e = hipMalloc((void**)&tdev, (size_t)B * 3 * n_in * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->theta_stage, theta, (size_t)B * ctx->n_params * sizeof(double), hipMemcpyHostToDevice, stream);
    int rc = 0;
    if (e == hipSuccess) {
        rc = dl_eval_theory(ctx, ctx->theta_stage, B, iobs, pdev, tdev, stream);
        if (rc == 0) {
            e = hipMemcpyAsync(power, pdev, (size_t)B * n_in * sizeof(double), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess && tables) e = hipMemcpyAsync(tables, tdev, (size_t)B * 3 * n_in * sizeof(double), hipMemcpyDeviceToHost, stream);
        }
        hipError_t es = hipStreamSynchronize(stream);   // (also on failure: the buffers below are freed)
        if (e == hipSuccess) e = es;
    }
    if (e != hipSuccess && rc == 0) rc = dl_fail(ctx, std::string("dl_eval_theory_host: ") + hipGetErrorString(e));
    if (pdev) (void)hipFree(pdev);
    if (tdev) (void)hipFree(tdev);
    return rc;
}

int dl_profile_enable(dl_ctx* ctx, int enable) {
    if (!ctx) { g_last_error = "dl_profile_enable: null context"; return 1; }
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if ((enable & 0xffff) && ctx->ev.empty()) {
        ctx->ev.assign((size_t)dl_ctx::NPOOL * 6, nullptr);
        for (auto& e : ctx->ev) DL_HIP_CHECK(ctx, hipEventCreate(&e));
    }
    ctx->prof_phase_of.assign(dl_ctx::NPOOL, -1);
    ctx->prof_rotate = (enable & (1 << 16)) != 0;
    ctx->prof_only = (enable >> 17) & 3;
    if (ctx->prof_only > 2) ctx->prof_only = 0;
    enable &= 0xffff;
    ctx->profile = enable != 0;
    ctx->prof_every = enable > 1 ? enable : 1;
    ctx->prof_calls = 0;
    ctx->eval_calls = 0;
    return 0;
}

int dl_profile_read(dl_ctx* ctx, double* ms, int32_t n) {
    if (!ctx || !ms || n < 4) return dl_fail(ctx, "dl_profile_read: need room for 4 values");
    if (ctx->ev.empty() || ctx->prof_calls == 0) return dl_fail(ctx, "dl_profile_read: no profiled call recorded");
    int64_t ncalls = std::min<int64_t>(ctx->prof_calls, dl_ctx::NPOOL);
    // MEDIAN over the sampled calls.  The intervals are the dispatch packets' own start / stop timestamps (hipExtLaunchKernelGGL): kernel durations as
    // rocprofv3 --kernel-trace reports them; [3] = start of the first kernel to the stop of the last one.  A phase without a launch reports 0.
    std::vector<double> samples[4];
    for (int64_t c = 0; c < ncalls; ++c) {
        hipEvent_t* ev = &ctx->ev[(size_t)c * 6];
        const int only = ctx->prof_phase_of.empty() ? -1 : ctx->prof_phase_of[(size_t)c];
        int first = -1, last = -1;
        for (int k = 0; k < 3; ++k) {
            if (only >= 0 && only != k) continue;
            (void)hipEventSynchronize(ev[2 * k + 1]);
            float t = 0;
            if (hipEventElapsedTime(&t, ev[2 * k], ev[2 * k + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (first < 0) first = k;
            last = k;
            samples[k].push_back(t);
        }
        float t = 0;
        if (only < 0 && first >= 0 && hipEventElapsedTime(&t, ev[2 * first], ev[2 * last + 1]) == hipSuccess) samples[3].push_back(t);
        else (void)hipGetLastError();
    }
    for (int i = 0; i < 4; ++i) if (samples[i].empty()) samples[i].push_back(0.);
    if (n >= 9) for (int i = 0; i < 3; ++i) ms[6 + i] = (samples[i].size() == 1 && samples[i][0] == 0.) ? 0. : (double)samples[i].size();
    for (int i = 0; i < 4; ++i) {
        std::sort(samples[i].begin(), samples[i].end());
        const size_t m = samples[i].size();
        ms[i] = (m & 1) ? samples[i][m / 2] : 0.5 * (samples[i][m / 2 - 1] + samples[i][m / 2]);
    }
    if (n >= 5) ms[4] = 0.;   // (no event-record overhead to subtract any more)
    if (n >= 6) ms[5] = (double)ncalls;
    return 0;
}

}  // extern "C"

static bool dl_fold_applicable(const dl_ctx* ctx, int64_t B) {
    const int64_t chi2_max_rows = dl_options().chi2_max_rows;
    if (ctx->feat_ok || ctx->any_transform || ctx->n_solved != 0 || B > chi2_max_rows || B > DL_CHUNK || ctx->n_params > 32) return false;
    for (auto& ob : ctx->obs) {
        const DlObsDev& oh = ob.dev;
        const bool generic = !oh.uniform_knots || !(oh.toeplitz || oh.fixed_spline);
        if (oh.theory >= 2 || generic || oh.n_ct > 0 || oh.n_sn > 0 || (oh.n_ell <= 3) != (ctx->obs[0].dev.n_ell <= 3)) return false;
    }
    return ctx->n_obs >= 1 && ctx->n_obs <= 8;
}

int dl_internal_fold_info(dl_ctx* ctx, int64_t B, int* n_tiles, const double** priors) {
    if (!ctx || !n_tiles || !priors || B <= 0) return 1;
    if (!dl_fold_applicable(ctx, B)) return 2;
    *n_tiles = ctx->N_pad / 16;
    *priors = ctx->priors_dev;
    return 0;
}

int dl_internal_eval_fold(dl_ctx* ctx, const DlEnsFold& fold, int64_t B, double* part_out, hipStream_t stream) {
    if (!ctx || !part_out || B <= 0) return 1;
    if (!dl_fold_applicable(ctx, B)) return 2;
    dl_prof_events.start = dl_prof_events.stop = nullptr;
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_order_streams(ctx, stream)) return 1;
    if (dl_reserve(ctx, B)) return 1;
    const int xcd_local = dl_options().xcd_local;
    if (!dl_launch_fullshape_ens(ctx->obs_kernarg.data(), ctx->n_obs, ctx->obs_array_dev, fold, B, ctx->power_ws, ctx->K_pad, xcd_local ? dl_chi2_gemm_row_tile(B, ctx->N_pad) : 0, stream)) return 2;
    dl_launch_chi2_gemm(ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->bias_white_dev, part_out, B, ctx->N_pad, ctx->K_pad, nullptr, nullptr, ctx->n_params,
                        ctx->priors_dev, nullptr, nullptr, nullptr, 1, stream, ctx->panel_ranges.empty() ? nullptr : ctx->panel_ranges.data(), ctx->K_live, nullptr, 0, ctx->wt_frag_dev);
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

int dl_internal_eval_partials(dl_ctx* ctx, const double* theta_dev, int64_t B, const double** part, int* n_tiles, const double** priors, hipStream_t stream) {
    if (!ctx || !theta_dev || !part || !n_tiles || !priors || B <= 0) return 1;
    const int64_t chi2_max_rows = dl_options().chi2_max_rows;
    if (ctx->feat_ok || ctx->any_transform || ctx->n_solved != 0 || B > chi2_max_rows || B > DL_CHUNK) return 2;
    for (auto& ob : ctx->obs) if (ob.dev.theory == 3) return 2;
    dl_prof_events.start = dl_prof_events.stop = nullptr;
    DL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dl_order_streams(ctx, stream)) return 1;
    if (dl_reserve(ctx, B)) return 1;
    const int xcd_local = dl_options().xcd_local;
    dl_launch_fullshape(ctx->obs_kernarg.data(), ctx->n_obs, theta_dev, ctx->n_params, B, ctx->power_ws, ctx->K_pad, nullptr, 0, stream, nullptr, 0, xcd_local ? dl_chi2_gemm_row_tile(B, ctx->N_pad) : 0, ctx->obs_array_dev);
    dl_launch_chi2_gemm(ctx->power_ws, ctx->K_pad, ctx->wt_white_dev, ctx->K_pad, ctx->bias_white_dev, ctx->delta_ws, B, ctx->N_pad, ctx->K_pad, nullptr, theta_dev, ctx->n_params,
                        ctx->priors_dev, nullptr, nullptr, nullptr, 1, stream, ctx->panel_ranges.empty() ? nullptr : ctx->panel_ranges.data(), ctx->K_live, nullptr, 0, ctx->wt_frag_dev);
    *part = ctx->delta_ws;
    *n_tiles = ctx->N_pad / 16;
    *priors = ctx->priors_dev;
    DL_HIP_CHECK(ctx, hipGetLastError());
    return 0;
}

