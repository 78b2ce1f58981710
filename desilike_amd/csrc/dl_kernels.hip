// dl_kernels.hip -- gfx950 (CDNA4) kernels of the full-shape likelihood path.
//
//   dl_fullshape_kernel : one workgroup per (point, observable): template -> spline -> AP -> multipoles
//                         -> tracer combination, everything staged in LDS.
//   dl_window_gemm      : C[B, N] = A[B, K] . Wt[N, K]^T + bias, fp64 MFMA v_mfma_f64_16x16x4_f64;
//                         used for the (precision-whitened) window convolution (rows a6 + a8).
//   dl_finalize_kernel  : chi2 = |whitened residual|^2 by wavefront shuffles, priors, status (rows a8 + a9).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dl_fullshape.h"
#include <mutex>
#include "dl_kernels.h"
#include "dl_tns.h"
#include "dl_emu_batch.h"
#include "dl_emu_stacked.h"
#include "dl_emu_stacked_ov.h"
#include "dl_emu_stacked_split.h"
#include "dl_finalize_part.h"
#include "dl_marg_solve.h"
#include "dl_scalar_prefetch.h"
#include "dl_ens_fold.h"
#include "dl_fullshape_grad.h"

thread_local DlProfEvents dl_prof_events;

// ------------------------------------------------------------------------------------------------
// theory kernel
// ------------------------------------------------------------------------------------------------
// The observable's constants travel BY VALUE in the kernarg segment: every pointer in it is then known to be a
// global-memory pointer (global_load, not flat_load) and every scalar field is an SGPR, never re-read in a loop.
// DENSE (fast kernels without counter terms): 71 VGPRs and 31 KB of LDS, five workgroups per CU -- for batches that keep every CU oversubscribed (+9 % at 32768 points);
// otherwise the two-wavenumber projection loop with 122 VGPRs, four workgroups per CU (shorter workgroup life: 14.0 vs 15.5 us per launch at 1024 points).
// ... of the parts of a description the Kaiser / EFT kernels read: the head (sizes, constants, parameter slots) and the tail (table pointers).  ONE wavefront of
// the workgroup asks (all four asking: 13.6 us against 11.9 us without -- the scalar cache serves several CUs), and
// the asking wavefront must also WAIT -- the destination registers are only borrowed for the duration of the asm statement, a load still in flight afterwards
// would land in whatever the compiler keeps there by then.  The other wavefronts' own loads of the same lines then ride on the requests already in flight.
// Headline theory kernel 11.9 -> 11.05 us.  Not for descriptions read from a device array (several observables in one launch: those lines stay in L2 from launch to
// launch, asking first cost config 5 +2.2 us per launch) and without effect on the throughput-bound BAO / emulator launches.
__device__ __forceinline__ void dl_obs_prefetch(const void* p) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 64) return;
    dl_scalar_prefetch<0, (offsetof(DlObsDev, ct_in) + 63) / 64 * 64, offsetof(DlObsDev, coef_w) / 64 * 64, (sizeof(DlObsDev) + 63) / 64 * 64>(p);
}

// ---- the table of diagnostic switches (dl_kernels.h) ----
static DlOptions g_options;
static std::once_flag g_options_once;     // (several host threads -- one per chain -- may make the first call)
static void dl_options_read() {
    auto on = [](const char* name) { return std::getenv(name) != nullptr; };
    DlOptions& o = g_options;
    o.no_merged_theory = on("DL_NO_MERGED_THEORY"); o.no_emu_batch = on("DL_NO_EMU_BATCH"); o.no_fused_solve = on("DL_NO_FUSED_SOLVE"); o.no_gram_plain = on("DL_NO_GRAM_PLAIN");
    o.no_scaled_row0 = on("DL_NO_SCALED_ROW0"); o.ef_no_early_theta = on("DL_EF_NO_EARLY_THETA"); o.fm_no_lane_solve = on("DL_FM_NO_LANE_SOLVE");
    o.no_stk_split = on("DL_NO_STK_SPLIT");
    o.stk_overlap = std::getenv("DL_STK_OVERLAP") ? atoi(std::getenv("DL_STK_OVERLAP")) : 0;
    o.ens_global = on("DL_ENS_GLOBAL"); o.ens_force_comm = on("DL_ENS_FORCE_COMM"); o.ens_no_defer = on("DL_ENS_NO_DEFER"); o.ens_no_fold = on("DL_ENS_NO_FOLD");
    o.ens_stamps = on("DL_ENS_STAMPS"); o.ens_fold_stamps = on("DL_ENS_FOLD_STAMPS");
    o.cg_mt = std::getenv("DL_CG_MT") ? atoi(std::getenv("DL_CG_MT")) : 0;
    o.host_mode = std::getenv("DL_HOST_MODE") ? atoi(std::getenv("DL_HOST_MODE")) : -1;
    auto flag = [](const char* name) { const char* v = std::getenv(name); return v != nullptr && atoi(v) != 0; };
    o.no_emu_fused = on("DL_NO_EMU_FUSED"); o.no_gram_epilogue = on("DL_NO_GRAM_EPILOGUE"); o.no_chi2_big = on("DL_NO_CHI2_BIG");
    o.step_kernel = flag("DL_STEP_KERNEL"); o.chi2_fused = flag("DL_CHI2_FUSED"); o.chi2_bfrag = flag("DL_CHI2_BFRAG");
    o.xcd_local = std::getenv("DL_XCD_LOCAL") ? atoi(std::getenv("DL_XCD_LOCAL")) : 1;
    o.chi2_max_rows = std::getenv("DL_CHI2_GEMM_MAX") ? atoll(std::getenv("DL_CHI2_GEMM_MAX")) : 2048;
}
const DlOptions& dl_options() {
    std::call_once(g_options_once, dl_options_read);
    return g_options;
}
extern "C" void dl_options_refresh(void) { (void)dl_options(); dl_options_read(); }

// Workgroups are dealt round-robin to the 8 XCDs; with xblk > 0 workgroup w = xcd + 8 r handles point xblk (xcd + 8 (r / xblk)) + r % xblk (see dl_fullshape_body)
__device__ __forceinline__ int dl_fs_point_of_wg(int wg, int xblk) { return xblk ? xblk * ((wg & 7) + 8 * ((wg >> 3) / xblk)) + ((wg >> 3) % xblk) : wg; }

// SUB: the point is evaluated by a 256-thread SUB-GROUP of a larger workgroup (dl_step_kernel: four points per 1024-thread workgroup): `lds_sub`, `tid_sub`, `b_sub` name its
// share of LDS, the thread's index in the sub-group and the point; every sub-group runs the same sequence of barriers (the branches between them are uniform in the observable).
// NT: threads of the point's workgroup (DL_FS_THREADS; the wide form of small batches, dl_fullshape_wide_kernel, runs a point on 512: every phase strides by the thread count)
template <bool FAST, int NL, bool EFT, bool DENSE, bool TH_ROW = false, bool SUB = false, int NT = DL_FS_THREADS>
__device__ __forceinline__ void dl_fullshape_body(const DlObsDev& o, const double* __restrict__ theta, int n_params, double* __restrict__ power,
                                                  int64_t ld_power, double* __restrict__ tables, int64_t ld_tables, int stop_after, unsigned long long* __restrict__ stamps,
                                                  const double* th_row = nullptr,     // th_row: the point's parameters already in LDS (dl_fullshape_ens_kernel)
                                                  double* lds_sub = nullptr, int tid_sub = 0, int b_sub = 0) {
    extern __shared__ __attribute__((aligned(16))) double lds_wg[];
    double* lds = SUB ? lds_sub : lds_wg;
    // Workgroups are dealt round-robin to the 8 XCDs; the GEMM that follows runs row block mb (xblk = 32 or 64 points) on XCD mb % 8.  With xblk > 0 the points are dealt
    // so that a row block is PRODUCED on the XCD that consumes it (B a multiple of 8 xblk): workgroup w = xcd + 8 r handles point xblk (xcd + 8 (r / xblk)) + r % xblk.
    const int xblk = SUB ? 0 : (stop_after >> 8) & 0xff;
    stop_after = (SUB || xblk) ? 0 : stop_after;
    const int b = SUB ? b_sub : dl_fs_point_of_wg(blockIdx.x, xblk);
    // DL_FS_STAMPS diagnostics: s_memtime (shader clock) of thread 0 at entry, after each barrier and at exit, 8 slots per workgroup
#define DL_STAMP(slot) if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();
    DL_STAMP(0)
    if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, same on every XCD
    // FAST instantiations are only launched when the convolution path applies (or the spline is fixed): the segmented sweeps are not compiled in
    const bool toep = !o.fixed_spline && (FAST || o.toeplitz);
    const DlFsShared s = dl_fs_shared_carve(lds, o.n_t, o.n_in, dl_fs_n_dd0(o), toep);
    const double* th = TH_ROW ? th_row : theta + (size_t)b * n_params;
    const int tid = SUB ? tid_sub : (int)threadIdx.x, nthr = NT;
    if (stop_after == -1) return;  // stop_after != 0: timing diagnostics only (DL_FS_STOP), outputs are then incomplete
    // constants of the later phases are requested now: their round trip hides behind phase 0/1
    double lk_pref[DL_P3_PREF];
#pragma unroll
    for (int it = 0; it < DL_P3_PREF; ++it) lk_pref[it] = (tid + it * nthr < o.n_kin) ? o.lkin[tid + it * nthr] : 0.;
    if (FAST) {
        // Waves 0-2 (DL_FS_KT threads) build the spline: knots -> convolution -> interval polynomials.  Wave 3 runs the per-mu chain beside them, one part
        // per phase (its results are first read in phase 3), lane 63 of it keeps the per-point scalars.
        constexpr int KT = NT - 64;       // (DL_FS_KT at 256 threads)
        const bool mu_wave = tid >= KT;
        const int m = tid - KT;                                  // mu node of this lane of the mu wave
        const bool mu_lane = mu_wave && m < o.n_mu, scalar_lane = (tid == nthr - 1);
        DlMuCarry c;
        double dlt_pref[DL_TOEP_PREF];
#pragma unroll
        for (int it = 0; it < DL_TOEP_PREF; ++it) dlt_pref[it] = (toep && !mu_wave && tid + it * KT < o.n_t - 1) ? o.dlt[tid + it * KT] : 0.;
        if (mu_wave) dl_fs_mu_partA(o, th, mu_lane ? m : 0, c);
        else dl_fs_knots(tid, KT, o, th, s);
        if (o.fixed_spline && mu_wave) {
            dl_fs_mu_partB(c);
            if (mu_lane) dl_fs_mu_partC(o, s, m, c, false);
            if (scalar_lane) dl_fs_scalars(o, th, s, c, false);
        }
        __syncthreads();
        DL_STAMP(1)
        if (stop_after == 1) return;
        if (toep) {
            if (mu_wave) dl_fs_mu_partB(c);
            else dl_fs_phase2_fir(tid, KT, o, s);
            __syncthreads();
            DL_STAMP(2)
            if (stop_after == 2) return;
            if (mu_wave) {
                if (mu_lane) dl_fs_mu_partC(o, s, m, c, false);
                if (scalar_lane) dl_fs_scalars(o, th, s, c, false);
            } else dl_fs_phase2d_toep(tid, KT, o, s, dlt_pref);
            __syncthreads();
            DL_STAMP(3)
            if (stop_after == 5) return;
        }
    } else {
        dl_fs_phase01(tid, nthr, o, th, s);
        __syncthreads();
        if (stop_after == 1) return;
        if (toep) {
            double dlt_pref[DL_TOEP_PREF];
#pragma unroll
            for (int it = 0; it < DL_TOEP_PREF; ++it) dlt_pref[it] = (tid + it * nthr < o.n_t - 1) ? o.dlt[tid + it * nthr] : 0.;
            dl_fs_phase2_fir(tid, nthr, o, s);
            __syncthreads();
            if (stop_after == 2) return;
            dl_fs_phase2d_toep(tid, nthr, o, s, dlt_pref);
            __syncthreads();
            if (stop_after == 5) return;
        } else if (!o.fixed_spline) {
            dl_fs_phase2a(tid, nthr, o, s);
            __syncthreads();
            if (stop_after == 2) return;
            dl_fs_phase2b_dot(tid, nthr, o, s);
            __syncthreads();
            dl_fs_phase2b(tid, nthr, o, s);
            __syncthreads();
            if (stop_after == 3) return;
            dl_fs_phase2c_dot(tid, nthr, o, s);
            __syncthreads();
            dl_fs_phase2c(tid, nthr, o, s);
            __syncthreads();
            if (stop_after == 4) return;
            dl_fs_phase2d(tid, nthr, o, s);
            __syncthreads();
            if (stop_after == 5) return;
        }
    }
    double* prow = power + (size_t)b * (1 + o.n_var) * ld_power + o.col_offset;
    double* trow = tables ? tables + (size_t)b * ld_tables : nullptr;
    if (FAST && !EFT && !DENSE) dl_fs_phase3_pair<NL, EFT>(tid, nthr, o, s, lk_pref);   // (with counter terms the pair variant spills registers)
    else dl_fs_phase3<FAST, NL, EFT>(tid, nthr, o, s, trow, lk_pref);
    __syncthreads();
    DL_STAMP(4)
    dl_fs_phase4(tid, nthr, o, s, th, prow, ld_power);
    DL_STAMP(5)
    if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#undef DL_STAMP
}

template <bool FAST, int NL, bool EFT, bool DENSE = false>
__global__ __launch_bounds__(DL_FS_THREADS, DENSE ? 5 : 4) void dl_fullshape_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, double* __restrict__ power,
                                                                     int64_t ld_power, double* __restrict__ tables, int64_t ld_tables, int stop_after, unsigned long long* __restrict__ stamps) {
    dl_obs_prefetch((const void*)__builtin_amdgcn_kernarg_segment_ptr());
    dl_fullshape_body<FAST, NL, EFT, DENSE>(o, theta, n_params, power, ld_power, tables, ld_tables, stop_after, stamps);
}

// Small batches (B x observables <= 256: at most one 256-thread workgroup per CU, a chain of latencies on a quarter of the CU's wave slots): the same body on 512 threads per point --
// seven waves build the spline (knots and interval polynomials stride by the thread count), the evaluation phase holds ONE wavenumber per thread instead of two.  Measured (round 6,
// profiles/r06d_wide_theory.txt): 64 - 256 points 18.0 - 18.2 -> 17.3 - 17.6 us per step; 512 points 19.05 -> 19.2, 1024 points 23.6 -> 28.6 us (four 512-thread workgroups do not fit a CU).
template <int NL>
__global__ __launch_bounds__(512, 2) void dl_fullshape_wide_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, double* __restrict__ power,
                                                                 int64_t ld_power, int stop_after, unsigned long long* __restrict__ stamps) {
    dl_obs_prefetch((const void*)__builtin_amdgcn_kernarg_segment_ptr());
    dl_fullshape_body<true, NL, false, false, false, false, 512>(o, theta, n_params, power, ld_power, nullptr, 0, stop_after, stamps);
}

// ---- scale-dependent bias from local primordial non-Gaussianity (theory kind 5; primordial_non_gaussianity.py:75-112) ---------------------------------------------
//   P(k, mu) = jac fog (bX + f mu'^2) (bY + f mu'^2) P(k') + sn0 / nd,   bX = b1X + bfnlX alpha(k'),   fog = 1 / ((1 + sX^2 k'^2 mu'^2 / 2) (1 + sY^2 k'^2 mu'^2 / 2)),
// with TWO not-a-knot splines in log10 k' per point: the template and alpha = alpha_fid sqrt(norm / template factor).  One workgroup per point; the spline phases of
// the generic kernel (dl_fullshape.h) run twice on the same work area, the interval polynomials of alpha are kept beside those of the template.
// LDS: generic layout (coef [4 n_t] | work | pt) | coefA [4 n_t] | mu records [DL_PNG_MAX_MU][8] | scalars [16]

__device__ __forceinline__ void dl_png_build_moments(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, bool toep) {
    if (toep) {
        dl_fs_phase2_fir(tid, nthr, o, s);
    } else {
        dl_fs_phase2a(tid, nthr, o, s);
        __syncthreads();
        dl_fs_phase2b_dot(tid, nthr, o, s);
        __syncthreads();
        dl_fs_phase2b(tid, nthr, o, s);
        __syncthreads();
        dl_fs_phase2c_dot(tid, nthr, o, s);
        __syncthreads();
        dl_fs_phase2c(tid, nthr, o, s);
    }
    __syncthreads();
}

__global__ __launch_bounds__(DL_FS_THREADS, 2) void dl_png_kernel(DlObsDev o, const double* __restrict__ theta, int n_params, double* __restrict__ power, int64_t ld_power) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, nthr = DL_FS_THREADS, n_t = o.n_t;
    const int64_t b = blockIdx.x;
    const double* th = theta + (size_t)b * n_params;
    const bool toep = o.toeplitz;
    const DlFsShared s = dl_fs_shared_carve(lds, n_t, o.n_in, -1, toep);
    double* tabs = s.coef;                                      // alpha knots | alpha second derivatives | template knots | template second derivatives
    double* murec = lds + dl_fs_shared_doubles(n_t, o.n_in);
    double* sc = murec + 8 * DL_PNG_MAX_MU;
    dl_png_setup(tid, nthr, o, th, murec, sc);
    dl_png_knots(tid, nthr, o, th, s, true);                    // alpha at the knots, its spline
    __syncthreads();
    dl_png_build_moments(tid, nthr, o, s, toep);
    dl_png_keep_spline(tid, nthr, o, s, toep, tabs, tabs + n_t);
    __syncthreads();
    dl_png_knots(tid, nthr, o, th, s, false);                   // the template at the knots, its spline
    __syncthreads();
    dl_png_build_moments(tid, nthr, o, s, toep);
    dl_png_keep_spline(tid, nthr, o, s, toep, tabs + 2 * n_t, tabs + 3 * n_t);
    __syncthreads();
    dl_png_eval(tid, nthr, o, tabs, murec, sc, s.out);          // (s.out aliases the spline work area: every thread is past it)
    __syncthreads();
    double* prow = power + (size_t)b * (1 + o.n_var) * ld_power + o.col_offset;
    for (int idx = tid; idx < o.n_in; idx += nthr) prow[idx] = s.out[idx];
}

// Several observables in ONE launch (blockIdx.y = observable): two DlObsDev (2 x 2016 bytes) do not fit the 4 KB kernarg segment, so the structs are read from a
// device array -- uniform, read-only addresses: scalar loads all the same.  Saves one launch ramp / drain per extra observable where the step is launch-latency
// bound (two tracers x 256 walkers: 2 x 8.6 us -> one launch).
template <bool FAST, int NL, bool EFT, bool DENSE = false>
__global__ __launch_bounds__(DL_FS_THREADS, DENSE ? 5 : 4) void dl_fullshape_multi_kernel(const DlObsDev* __restrict__ obs, const double* __restrict__ theta, int n_params,
                                                                           double* __restrict__ power, int64_t ld_power, int stop_after) {
    dl_fullshape_body<FAST, NL, EFT, DENSE>(obs[blockIdx.y], theta, n_params, power, ld_power, nullptr, 0, stop_after, nullptr);
}

// ---- folded ensemble update (dl_ens_fold.h): the workgroup derives the proposal it evaluates --------------------------------------------------------------------
// Wave 0 of the workgroup of slot b: the slot's move draw (uniform over the lanes), then the two halves of the wave re-evaluate the pending accepts of the slot's own
// walker and of its partner if they wait for one (dl_ens_decide2: lane-parallel, one round of loads -- the rows either decision can select are requested in the same
// round), then lanes p < P form theta_p = c_p - (c_p - x_p) z.  The row goes to LDS (the kernel's phases read the parameters from there); observable 0's workgroup
// also publishes it with the stretch factor: the NEXT half-step's launches need them for THIS half-step's accept.
#define DL_ENS_MAXP 32
#define DL_ENS_SLOTS_PER_WG 8    // an extra workgroup (4 waves) writes the state of 8 pending slots: two per wave
#define DL_EF_STAMP(slot) if (f.stamps != nullptr && publish && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); f.stamps[(size_t)b * 8 + (slot)] = __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ void dl_ens_propose(const DlEnsFold& f, int b, bool publish, double* __restrict__ th_out, double* __restrict__ scratch) {
#pragma clang fp contract(off)   // the NumPy driver rounds after every operation
    const int lane = threadIdx.x & 63;
    DL_EF_STAMP(1)
    const int half = f.nw / 2, P = f.P;
    const DlPhilox r = dl_philox4x32((uint32_t)f.it_prop, (uint32_t)((unsigned long long)f.it_prop >> 32), (uint32_t)b, DL_ENS_STREAM_MOVE + f.half_prop, f.k0, f.k1);
    const double u = dl_uniform53(r.x[0], r.x[1]);
    const double t = (f.a - 1.) * u + 1.;
    const double zz = (t * t) / f.a;                       // z ~ g(z) on [1 / a, a] (emcee moves/stretch.py)
    const int ic = dl_ens_split_at(f.split_prop, (1 - f.half_prop) * half + (int)(r.x[2] % (uint32_t)half));
    const int is = dl_ens_split_at(f.split_prop, f.half_prop * half + b);
    // pending accepts of the slot's own walker (decision A: lanes 0-31) and of its partner (B: lanes 32-63), if they wait for one
    const int sA = dl_ens_pending_slot(f.pend, is, half), sB = dl_ens_pending_slot(f.pend, ic, half);
    const int p = lane < P ? lane : P - 1;
    DL_EF_STAMP(2)
    // (requested with the decisions' loads: whichever row a decision selects is then already on its way)
    const double x_old = f.coords[(size_t)is * P + p], c_old = f.coords[(size_t)ic * P + p];
    const double x_new = f.pend.half >= 0 ? f.pend.prop[(size_t)(sA >= 0 ? sA : 0) * P + p] : 0., c_new = f.pend.half >= 0 ? f.pend.prop[(size_t)(sB >= 0 ? sB : 0) * P + p] : 0.;
    bool ax = false, ac = false;
    if (sA >= 0 || sB >= 0) {
        const DlEnsDecision2 d = dl_ens_decide2(f.pend, f.priors, P, f.n_tiles, f.offset, f.k0, f.k1, sA, f.logp[is], sB, f.logp[ic], scratch);
        ax = d.acc[0]; ac = d.acc[1];
    }
    DL_EF_STAMP(3)
    if (lane < P) {
        const double c = ac ? c_new : c_old, x = ax ? x_new : x_old;
        const double q = c - (c - x) * zz;                 // q = c - (c - s) z
        th_out[lane] = q;
        if (publish) f.prop_out[(size_t)b * P + lane] = q;
    }
    if (publish && lane == 0) f.factors_out[b] = (P - 1.) * log(zz);
    DL_EF_STAMP(4)
}

// Extra workgroup e of the launch: state after the pending accepts of slots [8 e, 8 e + 8) -- wave v handles slots 8 e + 2 v (lanes 0-31) and + 1 (lanes 32-63): the
// slot's walker (accepted: the proposal and its log-posterior, else the old row) and one walker outside the pending half (copied) go to the OTHER state buffer, and
// to the chain record if one is due: between them the extra workgroups write every walker.
__device__ __forceinline__ void dl_ens_write_state(const DlEnsFold& f, int e, double* __restrict__ scratch) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = f.nw / 2, P = f.P, hl = lane & 31, hb = lane >> 5;
    const int s0 = DL_ENS_SLOTS_PER_WG * e + 2 * wave, s1 = s0 + 1;
    if (s0 >= half) return;
    const bool two = s1 < half;
    const int w0 = dl_ens_split_at(f.pend.split, f.pend.half * half + s0), w1 = two ? dl_ens_split_at(f.pend.split, f.pend.half * half + s1) : w0;
    const double lw0 = f.logp[w0], lw1 = f.logp[w1];
    const int s = hb ? s1 : s0, w = hb ? w1 : w0;
    const int n = dl_ens_split_at(f.pend.split, (1 - f.pend.half) * half + (hb && !two ? s0 : s));   // a walker outside the pending half: copied
    const int p = hl < P ? hl : P - 1;
    const double v_old = f.coords[(size_t)w * P + p], v_new = f.pend.prop[(size_t)(hb && !two ? s0 : s) * P + p], v_n = f.coords[(size_t)n * P + p];
    const double ln = f.logp[n];
    const DlEnsDecision2 d = dl_ens_decide2(f.pend, f.priors, P, f.n_tiles, f.offset, f.k0, f.k1, s0, lw0, two ? s1 : -1, lw1, scratch + DL_ENS_SCRATCH * wave);
    if (hb && !two) return;
    const bool acc = hb ? d.acc[1] : d.acc[0];
    const double lp = hb ? d.lp[1] : d.lp[0], lw = hb ? lw1 : lw0;
    if (hl < P) {
        const double v = acc ? v_new : v_old;
        f.coords_out[(size_t)w * P + hl] = v; f.coords_out[(size_t)n * P + hl] = v_n;
        if (f.chain != nullptr) { f.chain[(size_t)w * P + hl] = v; f.chain[(size_t)n * P + hl] = v_n; }
    }
    if (hl == 0) {
        const double lv = acc ? lp : lw;
        f.logp_out[w] = lv; f.logp_out[n] = ln;
        if (acc) f.nacc[w] += 1;
        if (f.chain_logp != nullptr) { f.chain_logp[w] = lv; f.chain_logp[n] = ln; }
    }
}

// (non-DENSE: three workgroups per CU instead of four -- 168 registers: the prologue's values and the two-wavenumber projection loop do not fit 128 without spills,
//  and a half-ensemble of a few hundred proposals does not fill four slots per CU anyway)
template <int NL, bool DENSE = false>
__global__ __launch_bounds__(DL_FS_THREADS, DENSE ? 5 : 3) void dl_fullshape_ens_kernel(const DlObsDev* __restrict__ obs, const DlEnsFold f, double* __restrict__ power, int64_t ld_power,
                                                                         int flags, int B) {
    __shared__ double dl_ens_theta[DL_ENS_MAXP];
    __shared__ double dl_ens_scratch[DL_ENS_SCRATCH * (DL_FS_THREADS / 64)];
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    dl_kernarg_prefetch<(sizeof(DlEnsFold) + 8 + 63) / 64 * 64 + 64>();    // the fold description and the pointers behind it: one round trip instead of one per line
    if ((int)blockIdx.x >= B) {                        // extra workgroups: the state after the pending accepts (nothing of this launch reads it)
        if (blockIdx.y == 0 && f.pend.half >= 0) dl_ens_write_state(f, (int)blockIdx.x - B, dl_ens_scratch);
        return;
    }
    if (threadIdx.x < 64) dl_ens_propose(f, dl_fs_point_of_wg(blockIdx.x, (flags >> 8) & 0xff), blockIdx.y == 0, dl_ens_theta, dl_ens_scratch);
    else if (threadIdx.x < 128) {
        // a second wavefront touches the lines of the observable's description the phases will read (scalar cache: shared by the waves of the CU) while wave 0 derives
        // the proposal
        dl_scalar_prefetch<0, (offsetof(DlObsDev, ct_in) + 63) / 64 * 64, offsetof(DlObsDev, coef_w) / 64 * 64, (sizeof(DlObsDev) + 63) / 64 * 64>((const void*)(obs + blockIdx.y));
    }
    __syncthreads();
    if (f.stamps != nullptr && blockIdx.y == 0 && threadIdx.x == 0) { f.stamps[(size_t)dl_fs_point_of_wg(blockIdx.x, (flags >> 8) & 0xff) * 8 + 0] = t_entry; f.stamps[(size_t)dl_fs_point_of_wg(blockIdx.x, (flags >> 8) & 0xff) * 8 + 5] = __builtin_amdgcn_s_memtime(); }
    dl_fullshape_body<true, NL, false, DENSE, true>(obs[blockIdx.y], nullptr, f.P, power, ld_power, nullptr, 0, flags, nullptr, dl_ens_theta);
    if (f.stamps != nullptr && blockIdx.y == 0 && threadIdx.x == 0) f.stamps[(size_t)dl_fs_point_of_wg(blockIdx.x, (flags >> 8) & 0xff) * 8 + 6] = __builtin_amdgcn_s_memtime();
}

// One launch for all observables of a fast (uniform knots, no counter terms) full-shape likelihood, proposals derived in the kernel; false: not applicable
bool dl_launch_fullshape_ens(const DlObsDev* obs_host, int n_obs, const DlObsDev* obs_dev, const DlEnsFold& f, int64_t B, double* power, int64_t ld_power, int xcd_block, hipStream_t stream) {
    if (obs_dev == nullptr || n_obs < 1 || n_obs > 8 || f.P > DL_ENS_MAXP || f.P > 64) return false;
    const bool nl3 = obs_host[0].n_ell <= 3;
    size_t shmem = 0;
    for (int i = 0; i < n_obs; ++i) {
        const DlObsDev& oh = obs_host[i];
        const bool generic = !oh.uniform_knots || !(oh.toeplitz || oh.fixed_spline);
        if (oh.theory >= 2 || generic || oh.n_ct > 0 || oh.n_sn > 0 || (oh.n_ell <= 3) != nl3) return false;
        shmem = std::max(shmem, dl_fs_shared_doubles_obs(oh, true) * sizeof(double));
    }
    static const int64_t dense_min = getenv("DL_FS_DENSE_MIN") ? atoll(getenv("DL_FS_DENSE_MIN")) : 4096;
    auto launch = [&](auto kernel) {
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        const int flags = (xcd_block > 0 && B % (8 * xcd_block) == 0) ? (xcd_block << 8) : 0;
        const int n_extra = f.pend.half >= 0 ? (int)((B + DL_ENS_SLOTS_PER_WG - 1) / DL_ENS_SLOTS_PER_WG) : 0;   // workgroups that write the state after the pending accepts
        DL_LAUNCH(kernel, dim3((unsigned)(B + n_extra), (unsigned)n_obs), dim3(DL_FS_THREADS), shmem, stream, obs_dev, f, power, ld_power, flags, (int)B);
    };
    if (nl3) { if (B * n_obs > dense_min) launch(dl_fullshape_ens_kernel<3, true>); else launch(dl_fullshape_ens_kernel<3>); }
    else { if (B * n_obs > dense_min) launch(dl_fullshape_ens_kernel<5, true>); else launch(dl_fullshape_ens_kernel<5>); }
    return true;
}

// ---- analytic gradient (dl_fullshape_grad.h): one workgroup per (point, observable) contracts d(theory vector) / d(physical inputs) with Y = -W~^T d~ ----------------
// Phases (barriers between them): per-mu chain + gradient weights on wave 3 beside the template at the knots on waves 0-2; the template's dm-derivative data; the two
// convolutions; the two sets of interval polynomials; the (k, mu) contraction; [the dn spline and its contraction]; the fixed-order reduction.  72 KB of LDS: two
// workgroups per CU.
template <int NL>
__global__ __launch_bounds__(DL_FS_THREADS, 4) void dl_fullshape_grad_kernel(const DlObsDev* __restrict__ obs, const double* __restrict__ theta, int n_params,
                                                                              const double* __restrict__ Y, int64_t ldy, double* __restrict__ gphys) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const DlObsDev& o = obs[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = DL_FS_THREADS;
    constexpr int KT = DL_FS_KT;
    const double* th = theta + (size_t)b * n_params;
    const bool toep = o.toeplitz && !o.fixed_spline;
    const DlFsShared s = dl_fs_shared_carve(lds, o.n_t, o.n_in, dl_fs_n_dd0(o), toep);
    double* gw = s.pt + DL_PT_SIZE_FAST;
    double* red = gw + (size_t)DL_MAX_MU * DL_GW;
    // the spline of whatever sits in s.y: convolution, barrier, interval polynomials (waves 0-2), barrier
    auto build = [&]() {
        if (tid < KT) dl_fs_phase2_fir(tid, KT, o, s);
        __syncthreads();
        if (tid < KT) {
            double dlt_pref[DL_TOEP_PREF];
#pragma unroll
            for (int it = 0; it < DL_TOEP_PREF; ++it) dlt_pref[it] = (tid + it * KT < o.n_t - 1) ? o.dlt[tid + it * KT] : 0.;
            dl_fs_phase2d_toep(tid, KT, o, s, dlt_pref);
        }
        __syncthreads();
    };
    // wave 3: the per-mu chain and the gradient weights, beside the template at the knots and its convolution on waves 0-2 (their results are first read in the
    // contraction: the barrier after the convolution is the first one this wave's work must be finished by)
    if (tid >= KT) {
        const int m = tid - KT;
        const bool mu_lane = m < o.n_mu;
        DlMuCarry c;
        dl_fs_mu_partA(o, th, mu_lane ? m : 0, c);
        dl_fs_mu_partB(c);
        if (mu_lane) { dl_fs_mu_partC(o, s, m, c, false); dl_fs_grad_weights(o, m, c, gw); }
        if (tid == nthr - 1) { dl_fs_scalars(o, th, s, c, false); dl_fs_grad_weights_pad(o, gw); }
    } else dl_fs_knots(tid, KT, o, th, s);
    __syncthreads();
    if (toep) build();
    double acc[DL_NPHYS];
#pragma unroll
    for (int p = 0; p < DL_NPHYS; ++p) acc[p] = 0.;
    const double* Yrow = Y + (size_t)b * ldy + o.col_offset;
    dl_fs_grad_phase3<NL>(tid, nthr, o, s, gw, Yrow, 0, acc);
    if (toep && o.templ == 1) {
        for (int which = 0; which < 2; ++which) {
            if (which == 0 ? o.dm.col < 0 : o.dn.col < 0) continue;
            __syncthreads();                          // (everyone is done with the previous spline)
            if (tid < KT) dl_fs_grad_knots(tid, KT, o, th, s, which);
            __syncthreads();
            build();
            dl_fs_grad_phase3<NL>(tid, nthr, o, s, gw, Yrow, 1 + which, acc);
        }
    }
    // reduction in a fixed order: butterfly inside each wavefront (shuffles: registers), then the four wavefronts' sums through LDS.  (Eight threads walking 256 LDS
    // values each, one dependent read-and-add after the other -- the host emulation's dl_fs_grad_reduce -- took 10 us.)
#pragma unroll
    for (int p = 0; p < DL_NPHYS; ++p) {
        double v = acc[p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) red[(tid >> 6) * DL_NPHYS + p] = v;
    }
    __syncthreads();
    if (tid < DL_NPHYS) {
        double v = 0.;
#pragma unroll
        for (int w = 0; w < DL_FS_THREADS / 64; ++w) v += red[w * DL_NPHYS + tid];
        gphys[((size_t)b * gridDim.y + blockIdx.y) * DL_NPHYS + tid] = v;
    }
}

// chain rule to the theta columns + gradient of the log-prior (uniform: 0, norm: -(x - loc) / scale^2); rows whose log-posterior is -inf get a zero gradient
__global__ __launch_bounds__(64) void dl_grad_finalize_kernel(const DlObsDev* __restrict__ obs, int n_obs, const double* __restrict__ theta, int n_params, const double* __restrict__ priors,
                                                               const double* __restrict__ gphys, const int32_t* __restrict__ status, int64_t B, double* __restrict__ grad) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* th = theta + (size_t)b * n_params;
    double* g = grad + (size_t)b * n_params;
    const bool ok = status == nullptr || status[b] == 0;
    for (int p = 0; p < n_params; ++p) {
        const double* pr = priors + 5 * p;
        g[p] = (ok && pr[0] == 1.) ? -(th[p] - pr[3]) / (pr[4] * pr[4]) : 0.;          // parameter.py:2007 differentiated
    }
    if (!ok) return;
    for (int i = 0; i < n_obs; ++i) dl_fs_grad_chain(obs[i], th, gphys + ((size_t)b * n_obs + i) * DL_NPHYS, g);
}

bool dl_grad_applicable(const DlObsDev* obs_host, int n_obs) {
    if (n_obs < 1) return false;
    for (int i = 0; i < n_obs; ++i) if (!dl_fs_grad_applicable(obs_host[i])) return false;
    return true;
}

void dl_launch_fullshape_grad(const DlObsDev* obs_host, int n_obs, const DlObsDev* obs_dev, const double* theta, int n_params, int64_t B, const double* Y, int64_t ldy, double* gphys,
                              const double* priors, const int32_t* status, double* grad, hipStream_t stream) {
    size_t shm = 0;
    bool nl3 = true;
    for (int i = 0; i < n_obs; ++i) { shm = std::max(shm, dl_fs_grad_shared_doubles(obs_host[i]) * sizeof(double)); nl3 = nl3 && obs_host[i].n_ell <= 3; }
    auto launch = [&](auto kernel) {
        if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        DL_LAUNCH(kernel, dim3((unsigned)B, (unsigned)n_obs), dim3(DL_FS_THREADS), shm, stream, obs_dev, theta, n_params, Y, ldy, gphys);
    };
    if (nl3) launch(dl_fullshape_grad_kernel<3>); else launch(dl_fullshape_grad_kernel<5>);
    hipLaunchKernelGGL(dl_grad_finalize_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, obs_dev, n_obs, theta, n_params, priors, gphys, status, B, grad);
}

// BAO wiggle model: one workgroup per point; constant splines read from global memory, no per-point spline build.  One kernel per wiggle model: registers are
// allocated for the worst branch of a kernel (all four models behind one run-time switch: 165 VGPRs, three waves per SIMD, the standard model 15 % slower).
// Workgroup size by batch (dl_bao_threads): ONE wave per point once the batch fills the chip (>= 4096 points: 16 resident workgroups per CU overlap each other's
// per-mu chain and no wave waits at a barrier for another: 101 -> 86 us per 8192 points of the damped-BAO xi model, 70 -> 54 us for P_ell), 128 threads from 2048
// points, 256 threads (the shortest life of a single point) below.  More threads than wavenumbers were tried: 320 / 384 threads per point, 146 us.
template <int MODEL>
__global__ __launch_bounds__(512) void dl_bao_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, double* __restrict__ power, int64_t ld_power) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const double* th = theta + (size_t)b * n_params;
    dl_bao_phaseA(tid, nthr, o, th, lds);
    __syncthreads();
    dl_bao_phaseB_m<MODEL>(tid, nthr, o, lds);
    __syncthreads();
    dl_store_with_pass(tid, nthr, o, th, lds + DL_BAO_PT, power + (size_t)b * (1 + o.n_var) * ld_power + o.col_offset);
}

static int dl_bao_threads(int n_kin, int64_t B) {
    static const int forced = getenv("DL_BAO_THREADS") ? atoi(getenv("DL_BAO_THREADS")) : 0;
    if (forced >= 64 && forced <= 512 && forced % 64 == 0) return forced;
    int t = (n_kin + 63) / 64 * 64;
    t = t > 256 ? 256 : t;
    if (B >= 4096) t = 64;
    else if (B >= 2048 && t > 128) t = 128;
    return t;
}

// emulated theory: MLP / Taylor forward pass and feature expansion, one workgroup per point
__global__ __launch_bounds__(DL_FS_THREADS) void dl_emulated_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, double* __restrict__ power, int64_t ld_power,
                                                                   double* __restrict__ feat, int64_t feat_ld) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x;
    dl_emu_point(o, theta + (size_t)b * n_params, lds, power + (size_t)b * (1 + o.n_var) * ld_power + o.col_offset, ld_power,
                 feat ? feat + (size_t)b * feat_ld + o.feat_off : nullptr);
}

void dl_launch_fullshape(const DlObsDev* obs_host, int n_obs, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power, double* tables,
                         int64_t ld_tables, hipStream_t stream, double* feat, int64_t feat_ld, int xcd_block, const DlObsDev* obs_dev) {
    static const int stop_after = getenv("DL_FS_STOP") ? atoi(getenv("DL_FS_STOP")) : 0;   // per-phase timing diagnostics
    // DL_FS_STAMPS=<file>: in-kernel timestamps of the launches with B >= 256 are appended to <file> as text (synchronises: diagnostics only)
    static const char* stamp_file = getenv("DL_FS_STAMPS");
    static unsigned long long* stamps_dev = nullptr;
    static int stamp_launches = 0;
    if (stamp_file && !stamps_dev) { (void)hipMalloc((void**)&stamps_dev, (size_t)65536 * 8 * sizeof(unsigned long long)); }
    unsigned long long* stamps = (stamp_file && B >= 256 && B <= 65536 && stamp_launches >= 30 && stamp_launches < 34) ? stamps_dev : nullptr;
    if (stamp_file && B >= 256) stamp_launches++;
    // all observables in one launch when they share a fast instantiation (same multipole count class, no counter terms, no separate tables, same LDS footprint class)
    const bool merge = !dl_options().no_merged_theory;   // (the tests compare both paths in one process: dl_options_refresh)
    if (merge && obs_dev != nullptr && n_obs > 1 && n_obs <= 8 && tables == nullptr && feat == nullptr && stop_after == 0 && !stamp_file) {
        bool same = true, eft0 = obs_host[0].n_ct > 0 || obs_host[0].n_sn > 0, nl3 = obs_host[0].n_ell <= 3;
        size_t shmem = 0;
        for (int i = 0; i < n_obs && same; ++i) {
            const DlObsDev& oh = obs_host[i];
            const bool generic = !oh.uniform_knots || !(oh.toeplitz || oh.fixed_spline);
            if (oh.theory >= 2 || generic || (oh.n_ct > 0 || oh.n_sn > 0) != eft0 || (oh.n_ell <= 3) != nl3) same = false;
            shmem = std::max(shmem, dl_fs_shared_doubles_obs(oh, true) * sizeof(double));
        }
        if (same && !eft0) {
            static const int64_t dense_min = getenv("DL_FS_DENSE_MIN") ? atoll(getenv("DL_FS_DENSE_MIN")) : 4096;
            auto launch = [&](auto kernel) {
                if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                const int flags = (xcd_block > 0 && B % (8 * xcd_block) == 0) ? (xcd_block << 8) : 0;
                DL_LAUNCH(kernel, dim3((unsigned)B, (unsigned)n_obs), dim3(DL_FS_THREADS), shmem, stream, obs_dev, theta, n_params, power, ld_power, flags);
            };
            if (nl3) { if (B * n_obs > dense_min) launch(dl_fullshape_multi_kernel<true, 3, false, true>); else launch(dl_fullshape_multi_kernel<true, 3, false>); }
            else { if (B * n_obs > dense_min) launch(dl_fullshape_multi_kernel<true, 5, false, true>); else launch(dl_fullshape_multi_kernel<true, 5, false>); }
            return;
        }
    }
    for (int i = 0; i < n_obs; ++i) {  // one launch per observable (1-2 in practice)
        if (obs_host[i].theory == 5) {   // DL_THEORY_PNG
            const size_t shm = dl_png_shared_doubles(obs_host[i].n_t, obs_host[i].n_in) * sizeof(double);
            if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_png_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            DL_LAUNCH(dl_png_kernel, dim3((unsigned)B), dim3(DL_FS_THREADS), shm, stream, obs_host[i], theta, n_params, power, ld_power);
            continue;
        }
        if (obs_host[i].theory == 4) {   // DL_THEORY_TNS: loop GEMM + assembly (dl_tns.hip)
            dl_launch_tns(obs_host[i], theta, n_params, B, power, ld_power, stream);
            continue;
        }
        if (obs_host[i].theory == 3 && feat != nullptr && !dl_options().no_emu_batch) {   // feature path: 16 points per workgroup, MLP layers by MFMA
            size_t shm = dl_eb_shared_doubles(obs_host[i]) * sizeof(double);
            if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_emulated_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            DL_LAUNCH(dl_emulated_batch_kernel, dim3((unsigned)((B + DL_EB_PTS - 1) / DL_EB_PTS)), dim3(256), shm, stream, obs_host[i], theta, n_params, B, feat, feat_ld);
            continue;
        }
        if (obs_host[i].theory == 3) {   // DL_THEORY_EMULATED
            size_t shm = dl_emu_shared_doubles_obs(obs_host[i]) * sizeof(double);
            DL_LAUNCH(dl_emulated_kernel, dim3((unsigned)B), dim3(DL_FS_THREADS), shm, stream, obs_host[i], theta, n_params, power, ld_power, feat, feat_ld);
            continue;
        }
        if (obs_host[i].theory == 2) {   // DL_THEORY_BAO_DAMPED
            size_t shm = dl_bao_shared_doubles(obs_host[i].n_in) * sizeof(double);
            auto launch_bao = [&](auto kernel) {
                if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                DL_LAUNCH(kernel, dim3((unsigned)B), dim3((unsigned)dl_bao_threads(obs_host[i].n_kin, B)), shm, stream, obs_host[i], theta, n_params, power, ld_power);
            };
            const int model = obs_host[i].bao_mode >> 4;
            if (model == 0) launch_bao(dl_bao_kernel<0>);
            else if (model & 32) launch_bao(dl_bao_kernel<3>);
            else if (model & 16) launch_bao(dl_bao_kernel<2>);
            else launch_bao(dl_bao_kernel<1>);
            continue;
        }
        const DlObsDev& oh = obs_host[i];
        const bool generic = tables || !oh.uniform_knots || !(oh.toeplitz || oh.fixed_spline);
        size_t shmem = dl_fs_shared_doubles_obs(obs_host[i], !generic) * sizeof(double);
        auto launch = [&](auto kernel) {
            if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);  // e.g. 2000-knot BAO tables
            const int flags = (xcd_block > 0 && stop_after == 0 && B % (8 * xcd_block) == 0) ? (xcd_block << 8) : stop_after;
            DL_LAUNCH(kernel, dim3((unsigned)B), dim3(DL_FS_THREADS), shmem, stream, obs_host[i], theta, n_params, power, ld_power, tables, ld_tables, flags, stamps);
            if (stamps) {
                (void)hipStreamSynchronize(stream);
                std::vector<unsigned long long> h((size_t)B * 8);
                (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                if (FILE* f = fopen(stamp_file, "a")) {
                    for (int64_t w = 0; w < B; ++w) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", h[(size_t)w * 8 + q]); fprintf(f, "\n"); }
                    fprintf(f, "#\n");
                    fclose(f);
                }
            }
        };
        static const int64_t dense_min = getenv("DL_FS_DENSE_MIN") ? atoll(getenv("DL_FS_DENSE_MIN")) : 4096;   // batches above: the 5-workgroups-per-CU variant
        bool nl3 = oh.n_ell <= 3, eft = oh.n_ct > 0 || oh.n_sn > 0;
        static const int wide_env = getenv("DL_FS_WIDE") ? atoi(getenv("DL_FS_WIDE")) : -1;      // 1 / 0: force / forbid the 512-thread form (default: batches of at most 256 workgroups)
        const bool wide = !generic && !eft && tables == nullptr && (wide_env == 1 || (wide_env != 0 && B * n_obs <= 256));
        if (wide) {
            if (shmem > 48 * 1024) { (void)hipFuncSetAttribute((const void*)dl_fullshape_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); (void)hipFuncSetAttribute((const void*)dl_fullshape_wide_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); }
            const int flags = (xcd_block > 0 && stop_after == 0 && B % (8 * xcd_block) == 0) ? (xcd_block << 8) : stop_after;
            if (nl3) DL_LAUNCH(dl_fullshape_wide_kernel<3>, dim3((unsigned)B), dim3(512), shmem, stream, obs_host[i], theta, n_params, power, ld_power, flags, stamps);
            else DL_LAUNCH(dl_fullshape_wide_kernel<5>, dim3((unsigned)B), dim3(512), shmem, stream, obs_host[i], theta, n_params, power, ld_power, flags, stamps);
            continue;
        }
        if (generic) launch(dl_fullshape_kernel<false, 5, true>);   // generic: run-time decisions
        else if (nl3 && !eft) { if (B > dense_min) launch(dl_fullshape_kernel<true, 3, false, true>); else launch(dl_fullshape_kernel<true, 3, false>); }
        else if (nl3) launch(dl_fullshape_kernel<true, 3, true>);
        else if (!eft) { if (B > dense_min) launch(dl_fullshape_kernel<true, 5, false, true>); else launch(dl_fullshape_kernel<true, 5, false>); }
        else launch(dl_fullshape_kernel<true, 5, true>);
    }
}

// ------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM:  C[M, N] = A[M, K] . Wt[N, K]^T + bias[N]
//   A  row-major, leading dimension lda (multiple of 32, padding columns zero)
//   Wt row-major, leading dimension ldw (multiple of 32, padding zero), N_pad rows (multiple of 32)
//   one workgroup = 4 waves computes a 16 (M) x 32 (N) tile; the 4 waves split K in 32-wide chunks
//   (round-robin) and are summed through LDS.  v_mfma_f64_16x16x4_f64 operand layout (guide section 3):
//   A operand lane l = A[row l&15][k l>>4], B operand lane l = B[k l>>4][col l&15],
//   C/D reg r of lane l = C[row (l>>4) + 4 r][col l&15].
//   Inside a 32-chunk lane group g = l>>4 owns k = 8 q + 2 g + {0, 1}, q = 0..3 (16 B per lane per load, the four lane
//   groups of a row contiguous: 64 B per row per load instruction), i.e. the k -> (mfma step, lane group) assignment
//   is permuted identically for A and Wt: the sum is unchanged.
// ------------------------------------------------------------------------------------------------
typedef double dl_double4 __attribute__((ext_vector_type(4)));
typedef double dl_double2 __attribute__((ext_vector_type(2)));

#define DL_GEMM_WAVES 8   // waves per workgroup = K-split factor

struct DlGemmFrag {
    dl_double2 a[4], b0[4], b1[4];   // 8 consecutive k per lane for one A row and two Wt rows
};

__device__ __forceinline__ void dl_gemm_load(DlGemmFrag& f, const double* ap, const double* b0p, const double* b1p, int kc) {
    const dl_double2* a2 = reinterpret_cast<const dl_double2*>(ap + (size_t)kc * 32);
    const dl_double2* b02 = reinterpret_cast<const dl_double2*>(b0p + (size_t)kc * 32);
    const dl_double2* b12 = reinterpret_cast<const dl_double2*>(b1p + (size_t)kc * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) { f.a[q] = a2[4 * q]; f.b0[q] = b02[4 * q]; f.b1[q] = b12[4 * q]; }
}

__device__ __forceinline__ void dl_gemm_mma(const DlGemmFrag& f, dl_double4& acc0, dl_double4& acc1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[q].x, f.b0[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[q].x, f.b1[q].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[q].y, f.b0[q].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[q].y, f.b1[q].y, acc1, 0, 0, 0);
    }
}

__global__ __launch_bounds__(64 * DL_GEMM_WAVES) void dl_window_gemm_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ Wt, int64_t ldw,
                                                                            const double* __restrict__ bias, double* __restrict__ C, int64_t ldc, int M, int N_valid,
                                                                            int K_pad, int bias_period) {
    __shared__ __attribute__((aligned(16))) double red[DL_GEMM_WAVES - 1][2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 32;
    int arow = m0 + r16;
    if (arow > M - 1) arow = M - 1;
    // lane (r16, g) reads k = 8 q + 2 g + {0, 1} of each 32-chunk: one load instruction = 16 rows x 64 contiguous bytes
    const double* ap = A + (size_t)arow * lda + g * 2;
    const double* b0p = Wt + (size_t)(n0 + r16) * ldw + g * 2;
    const double* b1p = Wt + (size_t)(n0 + 16 + r16) * ldw + g * 2;
    dl_double4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
    const int nchunks = K_pad / 32;
    // software pipeline: the loads of chunk kc + 2 W are in flight while chunk kc is multiplied (two register buffers)
    DlGemmFrag f0, f1;
    int kc = wave;
    if (kc < nchunks) dl_gemm_load(f0, ap, b0p, b1p, kc);
    for (; kc < nchunks; kc += 2 * DL_GEMM_WAVES) {
        int kn = kc + DL_GEMM_WAVES;
        if (kn < nchunks) dl_gemm_load(f1, ap, b0p, b1p, kn);
        dl_gemm_mma(f0, acc0, acc1);
        if (kn < nchunks) {
            int kn2 = kn + DL_GEMM_WAVES;
            if (kn2 < nchunks) dl_gemm_load(f0, ap, b0p, b1p, kn2);
            dl_gemm_mma(f1, acc0, acc1);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[wave - 1][0][r][lane] = acc0[r]; red[wave - 1][1][r][lane] = acc1[r]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s0 = acc0[r], s1 = acc1[r];
#pragma unroll
            for (int w = 0; w < DL_GEMM_WAVES - 1; ++w) { s0 += red[w][0][r][lane]; s1 += red[w][1][r][lane]; }
            int row = m0 + g + 4 * r;
            if (row < M) {
                bool wb = (row % bias_period) == 0;   // derivative rows (analytic marginalisation) carry no bias
                if (n0 + r16 < N_valid) C[(size_t)row * ldc + n0 + r16] = wb ? s0 + bias[n0 + r16] : s0;
                if (n0 + 16 + r16 < N_valid) C[(size_t)row * ldc + n0 + 16 + r16] = wb ? s1 + bias[n0 + 16 + r16] : s1;
            }
        }
    }
}

void dl_launch_window_gemm(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* C, int64_t ldc, int64_t M, int N_valid, int N_pad,
                           int K_pad, int bias_period, hipStream_t stream) {
    dim3 grid((unsigned)((M + 15) / 16), (unsigned)(N_pad / 32));
    DL_LAUNCH(dl_window_gemm_kernel, grid, dim3(64 * DL_GEMM_WAVES), 0, stream, A, lda, Wt, ldw, bias, C, ldc, (int)M, N_valid, K_pad, bias_period);
}

// ------------------------------------------------------------------------------------------------
// Tiled split-K fp64 MFMA GEMM (main chi2 path):  slab[s][M, N] = A[M, Ks] . Wt[N, Ks]^T over the K-slice Ks of split s.
//   workgroup = 8 waves = 64 (M) x 128 (N) output tile; waves arranged 2 (M) x 4 (N), each 32 x 32 = 2 x 2 MFMA tiles (4 accumulators).
//   The kernel is bound by the LATENCY of operand delivery, not by MFMA issue or bandwidth (measured: ~2 us per dependent global round trip
//   with a 16-wide K step, whatever the number of workgroups), so K advances in "panels" of up to 96 columns = 6 x 16: a thread issues all
//   18 16-byte loads of a panel back to back (full 128-byte row segments, 8 lanes x 16 B), the panel is staged once through LDS
//   (192 rows x 98 doubles = 147 KB: one workgroup per CU; row stride = 4 banks mod 64 keeps the 32 lanes of a ds_read_b64 group on distinct banks)
//   and multiplied with 96 MFMAs per wave; the loads of the next panel are in flight meanwhile.  One round trip is exposed per workgroup.
//   Split-K over blockIdx.z keeps ~one workgroup per CU at small M; the partial slabs are summed by the finalize kernels
//   (deterministic order, no atomics).
// ------------------------------------------------------------------------------------------------
#include "dl_gemm_tiled.h"
#include "dl_gemm_dma.h"
#include "dl_chi2_gemm.h"
#include "dl_feature_gemm.h"

// number of K splits: ~one workgroup per CU at small M, whole panels per split
int dl_gemm_tiled_splits(int64_t M, int N_pad, int K_pad, int* chunks_per_split) {
    int64_t mtiles = (M + DL_GT_M - 1) / DL_GT_M, ntiles = N_pad / DL_GT_N;
    int nchunks = K_pad / DL_GT_K;
    static const int target = getenv("DL_GEMM_WGS") ? atoi(getenv("DL_GEMM_WGS")) : 256;   // tuning knob (<= 256: the slab workspace is sized for that)
    int64_t tiles = mtiles * ntiles;
    int S;
    if (tiles <= target) {
        S = (int)std::min<int64_t>(std::max<int64_t>(target / tiles, 1), std::min(nchunks, 32));   // small M: ~one workgroup per CU
    } else {
        // large M: one workgroup per CU at a time, so the launch runs in ceil(tiles S / CUs) rounds of (chunks / S + c0) each (c0 ~ 12 chunks of fixed cost per
        // workgroup); a second split pays when it removes a half-empty round (e.g. 384 tiles on 256 CUs: 2 x 92 -> 3 x 52).  At most 2: the slab workspace.
        auto cost = [&](int s) { return (double)((tiles * s + target - 1) / target) * ((double)((nchunks + s - 1) / s) + 12.); };
        S = (nchunks >= 2 * DL_GT_PANEL && cost(2) < 0.95 * cost(1)) ? 2 : 1;
    }
    int cps = (nchunks + S - 1) / S;
    if (S > 1) cps = (cps + DL_GT_PANEL - 1) / DL_GT_PANEL * DL_GT_PANEL;   // whole panels
    *chunks_per_split = cps;
    return (nchunks + cps - 1) / cps;
}

void dl_launch_window_gemm_tiled(const double* A, int64_t lda, const double* Wt, int64_t ldw, double* slabs, int64_t slab_stride, int64_t ldc, int64_t M, int N_pad, int K_pad,
                                 int n_splits, int chunks_per_split, hipStream_t stream, int n_live) {
    static const bool use_dma = !(getenv("DL_GEMM_DMA") && atoi(getenv("DL_GEMM_DMA")) == 0);   // DL_GEMM_DMA=0: register-staged predecessor (diagnostics)
    if (use_dma && chunks_per_split % 2 == 0 && K_pad % DL_GD_KP == 0) {
        dim3 grid((unsigned)((M + DL_GD_M - 1) / DL_GD_M), (unsigned)(N_pad / DL_GD_N), (unsigned)n_splits);
        static bool optin_dma = false;
        if (!optin_dma) {
            (void)hipFuncSetAttribute((const void*)dl_window_gemm_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GD_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)dl_window_gemm_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GD_LDS_BYTES);
            optin_dma = true;
        }
        DL_LAUNCH(dl_window_gemm_dma_kernel<false>, grid, dim3(64 * DL_GD_WAVES), DL_GD_LDS_BYTES, stream, A, lda, Wt, ldw, slabs, slab_stride, ldc, (int)M,
                           chunks_per_split / 2, K_pad / DL_GD_KP, nullptr, n_live > 0 ? n_live : N_pad);
        return;
    }
    dim3 grid((unsigned)((M + DL_GT_M - 1) / DL_GT_M), (unsigned)(N_pad / DL_GT_N), (unsigned)n_splits);
    static bool optin = false;
    if (!optin) { (void)hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES); optin = true; }
    DL_LAUNCH((dl_window_gemm_tiled_kernel<true, true, true>), grid, dim3(512), DL_GT_LDS_BYTES, stream, A, lda, Wt, ldw, slabs, slab_stride, ldc, (int)M, chunks_per_split, K_pad / DL_GT_K);
}

// split-K GEMM (single split) with the partial-chi2 epilogue: part[M, dl_gemm_dma_chi2_parts(N_pad)]
int dl_gemm_dma_chi2_parts(int N_pad) { return N_pad / (16 * DL_GD_TJ); }
void dl_launch_window_gemm_dma_chi2(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* part, int64_t M, int N_pad, int K_pad, hipStream_t stream,
                                    int n_live) {
    dim3 grid((unsigned)((M + DL_GD_M - 1) / DL_GD_M), (unsigned)(N_pad / DL_GD_N), 1);
    static bool optin = false;
    if (!optin) { (void)hipFuncSetAttribute((const void*)dl_window_gemm_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GD_LDS_BYTES); optin = true; }
    DL_LAUNCH(dl_window_gemm_dma_kernel<true>, grid, dim3(64 * DL_GD_WAVES), DL_GD_LDS_BYTES, stream, A, lda, Wt, ldw, part, (int64_t)0, (int64_t)0, (int)M, K_pad / DL_GD_KP,
                       K_pad / DL_GD_KP, bias, n_live > 0 ? n_live : N_pad);
}

// ------------------------------------------------------------------------------------------------
// observable transform (power_spectrum.py:402-404): flat -> (3 (flat / data)^(1/3) - 2) data, in place
// ------------------------------------------------------------------------------------------------
__global__ void dl_transform_kernel(double* __restrict__ flat, int64_t ld, const double* __restrict__ data, const int32_t* __restrict__ transform, int n, int64_t B) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    int64_t b = idx / n;
    int j = (int)(idx - b * n);
    if (transform[j] == 1) {
        double d = data[j], t = flat[b * ld + j];
        flat[b * ld + j] = (3. * pow(t / d, 1. / 3.) - 2.) * d;
    }
}

void dl_launch_transform(double* flat, int64_t ld, const double* data, const int32_t* transform, int n, int64_t B, hipStream_t stream) {
    int64_t total = B * n;
    DL_LAUNCH(dl_transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, flat, ld, data, transform, n, B);
}

// ------------------------------------------------------------------------------------------------
// finalize: one wavefront per point.  loglike = -1/2 sum_j dtilde_j^2 (likelihoods/base.py:13-17, 660 with the
// precision folded in as its Cholesky factor), logprior (parameter.py:1889-1897, 1994-2007), status.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dl_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void dl_finalize_kernel(const double* __restrict__ dtilde, int64_t ld, int n, int n_slabs, int64_t slab_stride,
                                                          const double* __restrict__ bias, const double* __restrict__ theta, int n_params,
                                                          const double* __restrict__ priors, int64_t B, double* __restrict__ loglike,
                                                          double* __restrict__ logprior, int32_t* __restrict__ status, int post_mode) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const double* row = dtilde + (size_t)b * ld;
    double sum = 0.;
    // split-K slabs are summed in a fixed order (deterministic); two columns x eight slabs = 16 independent loads in flight per lane
    for (int j = lane; j < n; j += 128) {
        const bool two = (j + 64 < n);
        double v0 = bias ? bias[j] : 0., v1 = (bias && two) ? bias[j + 64] : 0.;
        int sl = 0;
        for (; sl + 8 <= n_slabs; sl += 8) {
            double t[8], u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { t[q] = row[(size_t)(sl + q) * slab_stride + j]; u[q] = two ? row[(size_t)(sl + q) * slab_stride + j + 64] : 0.; }
            v0 += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            v1 += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
        }
        double t[8], u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bool on = sl + q < n_slabs;
            t[q] = on ? row[(size_t)(sl + q) * slab_stride + j] : 0.;
            u[q] = (on && two) ? row[(size_t)(sl + q) * slab_stride + j + 64] : 0.;
        }
        v0 += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        v1 += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
        sum = fma(v0, v0, sum);
        sum = fma(v1, v1, sum);
    }
    sum = dl_wave_sum(sum);
    // priors: lanes stride over parameters
    double lp = 0.;
    int nan_in = 0;
    const double inf = __builtin_huge_val();
    for (int p = lane; p < n_params; p += 64) {
        double x = theta[(size_t)b * n_params + p];
        const double* pr = priors + 5 * p;
        if (x != x) nan_in = 1;
        lp += dl_prior_logpdf(pr, x);
    }
    lp = dl_wave_sum(lp);
    nan_in = __any(nan_in);
    if (lane == 0) {
        double ll = -0.5 * sum;
        int st = DL_ST_OK;
        if (nan_in) st = DL_ST_NAN_INPUT;
        else if (lp == -inf) st = DL_ST_OUT_OF_PRIOR;
        else if (!(ll == ll) || ll == inf || ll == -inf) st = DL_ST_NONFINITE;
        if (loglike) loglike[b] = post_mode ? (st == DL_ST_OK ? ll + lp : -inf) : ll;   // post_mode: log-posterior with the samplers' conventions (samplers/base.py:185-191)
        if (logprior) logprior[b] = lp;
        if (status) status[b] = st;
    }
}

void dl_launch_finalize(const double* dtilde, int64_t ld, int n, int n_slabs, int64_t slab_stride, const double* bias, const double* theta, int n_params,
                        const double* priors, int64_t B, double* loglike, double* logprior, int32_t* status, int post_mode, hipStream_t stream) {
    DL_LAUNCH(dl_finalize_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, dtilde, ld, n, n_slabs, slab_stride, bias, theta, n_params, priors, B, loglike,
                       logprior, status, post_mode);
}

// ------------------------------------------------------------------------------------------------
// feature GEMM of the emulated theories (dl_feature_gemm.h): residual rows [B * R, N_pad] from the point records
// ------------------------------------------------------------------------------------------------
void dl_launch_feature_gemm(const double* feat, int64_t feat_ld, int64_t feat_off, int nb_pad, int R, const double* gfrag, double* out, int64_t ldo, int N_pad, int64_t B,
                            int accumulate, hipStream_t stream) {
    const int rec_len = nb_pad + R * DL_FG_MONO_LD;
    const size_t shm = (size_t)DL_FG_PTS * dl_fg_lds_stride(rec_len) * sizeof(double);
    if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_feature_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    DL_LAUNCH(dl_feature_gemm_kernel, dim3((unsigned)((B + DL_FG_PTS - 1) / DL_FG_PTS), (unsigned)(N_pad / 128)), dim3(512), shm, stream, feat, feat_ld, feat_off, nb_pad, R,
                       gfrag, out, ldo, B, accumulate);
}

// fused emulator forward + feature GEMM (dl_emu_batch.h): theta -> residual rows of one observable in one launch
void dl_launch_emulated_feature(const DlObsDev& obs, const double* theta, int n_params, int64_t B, const double* gfrag, double* out, int64_t ldo, int N_pad, int accumulate,
                                hipStream_t stream) {
    const size_t shm = dl_ef_shared_doubles(obs) * sizeof(double);
    if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_emulated_feature_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    DL_LAUNCH(dl_emulated_feature_kernel, dim3((unsigned)((B + DL_EB_PTS - 1) / DL_EB_PTS), (unsigned)(N_pad / 128)), dim3(512), shm, stream, obs, theta, n_params, B, gfrag,
                       out, ldo, accumulate);
}

// stacked table engine (dl_emu_stacked.h): networks of every group + feature GEMM, theta -> residual rows of one observable in one launch
bool dl_emulated_stacked_ok(const DlObsDev& obs) { return dl_stk_feature_ok(obs); }
// fin != nullptr (one observable, N_pad = 128, every solved parameter on a device row or constant, X fits the LDS): the marginalised finalize runs in the kernel's tail
// (outputs of DlGramFinalize; fin->done = true) and no row is written; otherwise the residual rows go to `out`
bool dl_launch_stk_chains(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* basis_ws, hipStream_t stream) {
    if (basis_ws == nullptr || dl_options().no_stk_split || (dl_options().stk_overlap != 0 && dl_stko_ok(obs)) || !dl_stks_ok(obs)) return false;
    const int n_pt_tiles = (int)((B + DL_STK_PTS - 1) / DL_STK_PTS);
    const int64_t chains = (int64_t)obs.stk.n_trunks * n_pt_tiles, ldk = (int64_t)obs.stk.n_trunks * obs.eng[0].widths[obs.eng[0].n_layers];
    if (chains > 0) {
        const size_t shm_a = (size_t)4 * DL_STK_PTS * DL_STKS_LD * sizeof(double);
        auto launch = [&](auto kernel) { DL_LAUNCH(kernel, dim3((unsigned)((chains + 3) / 4)), dim3(256), shm_a, stream, theta, n_params, B, obs, basis_ws, ldk, n_pt_tiles); };
        if (obs.eng[0].act == 0) launch(dl_stk_chain_kernel<0>); else if (obs.eng[0].act == 1) launch(dl_stk_chain_kernel<1>); else launch(dl_stk_chain_kernel<2>);
    }
    return true;
}

void dl_launch_emulated_stacked(const DlObsDev& obs, const double* theta, int n_params, int64_t B, const double* gfrag, double* out, int64_t ldo, int N_pad, int accumulate,
                                int steps_per_block, hipStream_t stream, DlGramFinalize* fin, const double* bias, const DlMargDev* mg, int n_valid, double* basis_ws, bool chains_done) {
    // the overlapped form (dl_emu_stacked_ov.h) where the shape allows: networks of the next batch under the feature GEMM of the current group
    const bool overlap = dl_options().stk_overlap != 0 && dl_stko_ok(obs);   // (off by default: measured slower than the plain form, docs/EXPERIMENTS.md round 6)
    // the two-launch form (dl_emu_stacked_split.h): every network a wave-private chain, then the feature GEMMs with the basis records through memory
    const bool split = !overlap && basis_ws != nullptr && !dl_options().no_stk_split && dl_stks_ok(obs);
    size_t shm = (overlap || split) ? (dl_stko_fixed_doubles() + dl_stko_work_doubles(obs)) * sizeof(double) : dl_stk_shared_doubles(obs) * sizeof(double);
    const int R = 1 + obs.n_var;
    const unsigned grid = (unsigned)((B + DL_STK_PTS - 1) / DL_STK_PTS);
    DlStkTail tl;
    std::memset(&tl, 0, sizeof(tl));
    bool tail_fits = false;
    if (mg != nullptr && mg->n_s >= 0 && 1 + mg->n_s <= 8) {
        if (overlap || split) {   // X [16][xr][DL_FG_XLD] over the work area, which grows to hold it when the LDS allows
            const size_t need = (dl_stko_fixed_doubles() + (size_t)DL_STK_PTS * (1 + mg->n_s) * DL_FG_XLD) * sizeof(double);
            if (need + DL_STK_STATIC_LDS <= 160 * 1024) { tail_fits = true; if (need > shm) shm = need; }
        } else tail_fits = dl_stk_tail_fits(obs, 1 + mg->n_s);
    }
    if (fin != nullptr && mg != nullptr && N_pad == 128 && n_valid <= 128 && !accumulate && tail_fits && !dl_options().no_fused_solve) {
        bool ok = true;
        tl.xr = 1 + mg->n_s;
        tl.row_of[0] = 0; tl.cst[0] = bias;
        for (int s = 0; s < mg->n_s && ok; ++s) {
            const int slot = mg->var_slot[s];
            if (slot >= 0) {
                if (1 + slot >= R || 1 + slot >= DL_STK_ROWS) ok = false;
                else { tl.row_of[1 + slot] = 1 + s; tl.cst[1 + slot] = mg->tconst + (size_t)s * 128; }
            } else { tl.const_row[tl.n_const] = 1 + s; tl.const_ptr[tl.n_const] = mg->tconst + (size_t)s * 128; tl.n_const++; }
        }
        for (int r = 1; r < R && ok; ++r) if (tl.cst[r] == nullptr) ok = false;   // a device row that feeds no solved parameter
        if (ok) {
            tl.enabled = 1; tl.post_mode = fin->post_mode & 0xff; tl.priors = fin->priors; tl.loglike = fin->loglike; tl.logprior = fin->logprior; tl.status = fin->status;
            tl.solved = fin->solved; tl.hessian = fin->hessian; tl.mg = *mg;
            fin->done = true;
        }
    }
    static const char* stamp_file = getenv("DL_STK_STAMPS");   // diagnostics: s_memtime at the phase boundaries of launches 30..33 appended to the file (synchronises)
    static unsigned long long* stamps_dev = nullptr;
    static int stamp_launches = 0;
    const int slots = (overlap || split) ? 128 : 32;                      // per workgroup (the overlapped kernel stamps waves 0 and 4, 64 slots each)
    if (stamp_file && !stamps_dev) (void)hipMalloc((void**)&stamps_dev, (size_t)8192 * 128 * sizeof(unsigned long long));
    unsigned long long* stamps = (stamp_file && grid <= 8192 && stamp_launches >= 30 && stamp_launches < 34) ? stamps_dev : nullptr;
    if (stamp_file) stamp_launches++;
    if (stamps) (void)hipMemsetAsync(stamps, 0, (size_t)grid * slots * sizeof(unsigned long long), stream);
    auto launch = [&](auto kernel) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        DL_LAUNCH(kernel, dim3(grid, (unsigned)(N_pad / 128)), dim3(512), shm, stream, theta, n_params, B, gfrag, obs, out, ldo, accumulate, steps_per_block, stamps, tl);
    };
    auto launch_ov = [&](auto kernel) {
        const int sel = dl_options().stk_overlap, mode = (sel & 4) ? 4 : ((sel & 2) ? 0 : 1);       // kernel's mode bits: 1 raised priority of the network waves, 4 split halves
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        DL_LAUNCH(kernel, dim3(grid, (unsigned)(N_pad / 128)), dim3(512), shm, stream, theta, n_params, B, gfrag, obs, out, ldo, accumulate, steps_per_block, stamps, tl, mode);
    };
    const bool wide = dl_stk_tld(obs) > 66;   // a layer wider than 64 units: eight output tiles per layer
    if (split) {
        const int Hs = obs.eng[0].widths[obs.eng[0].n_layers];
        const int64_t ldk = (int64_t)obs.stk.n_trunks * Hs;
        if (!chains_done) (void)dl_launch_stk_chains(obs, theta, n_params, B, basis_ws, stream);
        auto launch_b = [&](auto kernel) {
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            DL_LAUNCH(kernel, dim3(grid, (unsigned)(N_pad / 128)), dim3(512), shm, stream, theta, n_params, B, gfrag, obs, out, ldo, accumulate, steps_per_block, stamps, tl, (const double*)basis_ws, ldk);
        };
        if (R <= 1) launch_b(dl_emulated_stacked_gemm_kernel<1>); else if (R <= 4) launch_b(dl_emulated_stacked_gemm_kernel<4>); else if (R <= 6) launch_b(dl_emulated_stacked_gemm_kernel<6>); else launch_b(dl_emulated_stacked_gemm_kernel<8>);
    } else if (overlap) { if (R <= 1) launch_ov(dl_emulated_stacked_ov_kernel<1, DL_STKO_TMAX>); else if (R <= 4) launch_ov(dl_emulated_stacked_ov_kernel<4, DL_STKO_TMAX>); else if (R <= 6) launch_ov(dl_emulated_stacked_ov_kernel<6, DL_STKO_TMAX>); else launch_ov(dl_emulated_stacked_ov_kernel<8, DL_STKO_TMAX>); }
    else if (wide) { if (R <= 1) launch(dl_emulated_stacked_kernel<8, 1>); else if (R <= 4) launch(dl_emulated_stacked_kernel<8, 4>); else if (R <= 6) launch(dl_emulated_stacked_kernel<8, 6>); else launch(dl_emulated_stacked_kernel<8, 8>); }
    else { if (R <= 1) launch(dl_emulated_stacked_kernel<4, 1>); else if (R <= 4) launch(dl_emulated_stacked_kernel<4, 4>); else if (R <= 6) launch(dl_emulated_stacked_kernel<4, 6>); else launch(dl_emulated_stacked_kernel<4, 8>); }
    if (stamps) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * slots);
        (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_file, "a")) {
            for (unsigned w = 0; w < grid; ++w) { for (int q = 0; q < slots; ++q) fprintf(f, "%llu ", h[(size_t)w * slots + q]); fprintf(f, "\n"); }
            fprintf(f, "#\n");
            fclose(f);
        }
    }
}

bool dl_launch_emulated_feature_gram(const DlObsDev& obs, const double* theta, int n_params, int64_t B, const double* gfrag, const double* bias, const DlMargDev& mg, int n_valid,
                                     double* gram, hipStream_t stream, DlGramFinalize* fin) {
    const int R = 1 + obs.n_var;
    if (obs.eng[0].type == 2) return false;   // (stacked table engine: rows through dl_launch_emulated_stacked, then the general finalize kernels)
    if (R > 6 || mg.n_s < 0 || mg.n_s > 15 || n_valid > 128) return false;
    // no solved parameters: only with the finalize in the kernel's tail (there is no separate finalize on a 1 x 1 Gram matrix)
    if (mg.n_s == 0 && (fin == nullptr || dl_options().no_fused_solve || dl_options().no_gram_plain)) return false;
    DlEfGramArgs ga;
    std::memset(&ga, 0, sizeof(ga));
    ga.xr = 1 + mg.n_s; ga.gram = gram;
    ga.row_of[0] = 0; ga.cst[0] = bias;
    for (int s = 0; s < mg.n_s; ++s) {
        const int slot = mg.var_slot[s];
        if (slot >= 0) {
            if (1 + slot >= R) return false;
            ga.row_of[1 + slot] = 1 + s; ga.cst[1 + slot] = mg.tconst + (size_t)s * 128;   // (N_pad = 128: tconst rows are 128 doubles apart)
        } else { ga.const_row[ga.n_const] = 1 + s; ga.const_ptr[ga.n_const] = mg.tconst + (size_t)s * 128; ga.n_const++; }
    }
    for (int r = 1; r < R; ++r) if (ga.cst[r] == nullptr) return false;   // a device row that feeds no solved parameter
    for (int r = 0; r < 6; ++r) ga.nz[r][0] = ga.nz[r][1] = -1;
    for (int c = 4; c < DL_N_VPARS; ++c) {   // the monomials each derivative row touches
        const int slot = obs.vp_slot[c];
        if (slot >= 0 && 1 + slot < 6) dl_velocileptors_row_support(obs, c, ga.nz[1 + slot]);
    }
    // every derivative row on monomials 12-18 (the solved alpha* / sn* of the velocileptors order): monomials 0-11 feed row 0 only, through registers (DL_NO_SCALED_ROW0=1: three
    // full epilogues per wave, the form before; read at every launch: the tests compare)
    // (and eight k-step pairs in the main loops -- 64 hidden units + the constant basis function that starts the accumulators: that form is completely unrolled)
    ga.scaled = DL_FG_NM == 19 && obs.mono_mode != 0 && obs.eng[0].type == 0 && obs.n_basis == 65 && obs.nb_pad == 72 && !dl_options().no_scaled_row0;
    for (int r = 1; r < 6; ++r)
        for (int z = 0; z < 2; ++z) if (ga.nz[r][z] >= 0 && ga.nz[r][z] < 12) ga.scaled = 0;
    ga.no_early = dl_options().ef_no_early_theta;
    const size_t shm = dl_ef_gram_shared_doubles(obs, ga.xr) * sizeof(double);
    if (shm > 146 * 1024) return false;   // (the kernel also holds 10 KB of static LDS: parameter rows, prior table and prior terms of the fused finalize)
    // (per device: a second device of the process needs the raised limit too -- the attribute is set at every launch, as the other launchers do; it is a host-side table write)
    (void)hipFuncSetAttribute((const void*)dl_emulated_feature_gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    static const char* stamp_file = getenv("DL_EF_STAMPS");   // diagnostics: s_memtime at the phase boundaries of launches 30..33 appended to the file (synchronises)
    static unsigned long long* stamps_dev = nullptr;
    static int stamp_launches = 0;
    const unsigned grid = (unsigned)((B + DL_EB_PTS - 1) / DL_EB_PTS);
    if (stamp_file && !stamps_dev) (void)hipMalloc((void**)&stamps_dev, (size_t)8192 * 16 * sizeof(unsigned long long));
    ga.stamps = (stamp_file && grid <= 8192 && stamp_launches >= 30 && stamp_launches < 34) ? stamps_dev : nullptr;
    if (stamp_file) stamp_launches++;
    if (ga.stamps) (void)hipMemsetAsync(ga.stamps, 0, (size_t)grid * 16 * sizeof(unsigned long long), stream);
    DlEfSolve sv;
    std::memset(&sv, 0, sizeof(sv));
    if (fin != nullptr && ga.xr <= 8 && !dl_options().no_fused_solve) {   // (DL_NO_FUSED_SOLVE=1: Gram matrix to memory + dl_finalize_marg_gram_kernel)
        sv.enabled = 1; sv.post_mode = fin->post_mode & 0xff; sv.priors = fin->priors; sv.loglike = fin->loglike; sv.logprior = fin->logprior; sv.status = fin->status;
        sv.solved = fin->solved; sv.hessian = fin->hessian; sv.mg = mg;
        fin->done = true;
    }
    DL_LAUNCH(dl_emulated_feature_gram_kernel, dim3(grid, 1), dim3(512), shm, stream, theta, n_params, B, gfrag, obs, ga, sv);
    if (ga.stamps) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * 16);
        (void)hipMemcpy(h.data(), ga.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_file, "a")) {
            for (unsigned w = 0; w < grid; ++w) { for (int q = 0; q < 16; ++q) fprintf(f, "%llu ", h[(size_t)w * 16 + q]); fprintf(f, "\n"); }
            fprintf(f, "#\n");
            fclose(f);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// chi2 GEMM path (plain likelihood): partial chi2 per (point, 16-column block) from dl_chi2_gemm_kernel, then one THREAD per point
// sums them in a fixed order and adds the priors (same status logic as dl_finalize_kernel).
// ------------------------------------------------------------------------------------------------
// rows per workgroup of the chi2 GEMM for a batch of M points: 16 when 32-row blocks would occupy at most half of the 256 CUs (DL_CG_MT=32 / 16 overrides: diagnostics)
int dl_chi2_gemm_row_tile(int64_t M, int N_pad) {
    const int forced = dl_options().cg_mt;   // (the tests compare both tiles in one process: dl_options_refresh)
    if (forced == 16 || forced == 32) return forced;
    return ((M + DL_CG_M - 1) / DL_CG_M) * (N_pad / DL_CG_N) <= 128 ? 16 : DL_CG_M;
}

void dl_launch_chi2_gemm(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* part, int64_t M, int N_pad, int K_pad, int32_t* counters,
                         const double* theta, int n_params, const double* priors, double* loglike, double* logprior, int32_t* status, int post_mode, hipStream_t stream,
                         const uint8_t* panel_ranges, int k_live, double* resid, int64_t ldr, const double* wfrag) {
    const int n_tiles = N_pad / DL_CG_N;
    const int mt = counters == nullptr ? dl_chi2_gemm_row_tile(M, N_pad) : DL_CG_M;   // (the experimental fused finalize counts 32-row blocks)
    const int64_t mblocks = (M + mt - 1) / mt;
    const unsigned grid = (unsigned)(8 * n_tiles * ((mblocks + 7) / 8));
    static bool optin = false;
    if (!optin) {
        (void)hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<true, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<true, true, DL_CG_M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<true, true, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES);
        optin = true;
    }
    DlChi2Fin fin;
    fin.counters = counters; fin.theta = theta; fin.priors = priors; fin.loglike = loglike; fin.logprior = logprior; fin.status = status;
    fin.n_params = n_params; fin.post_mode = post_mode; fin.ready = nullptr;
    static const char* stamp_file = getenv("DL_CG_STAMPS");   // diagnostics: in-kernel timestamps of launches 30..33 appended to the file (synchronises)
    static unsigned long long* stamps_dev = nullptr;
    static int stamp_launches = 0;
    if (stamp_file && !stamps_dev) (void)hipMalloc((void**)&stamps_dev, (size_t)65536 * 8 * sizeof(unsigned long long));
    fin.stamps = (stamp_file && grid <= 65536 && M >= 256 && stamp_launches >= 30 && stamp_launches < 34) ? stamps_dev : nullptr;
    if (stamp_file && M >= 256) stamp_launches++;
    DlChi2Panels panels;
    std::memset(&panels, 0, sizeof(panels));
    if (panel_ranges != nullptr && n_tiles <= DL_CG_MAX_TILES)
        for (int t = 0; t < n_tiles; ++t) panels.range[t] = (uint32_t)panel_ranges[2 * t] | ((uint32_t)panel_ranges[2 * t + 1] << 8);
    double* const no_resid = nullptr;
    int max_live = K_pad / DL_CG_KP;      // live panels of the widest column block
    if (panel_ranges != nullptr && n_tiles <= DL_CG_MAX_TILES) {
        max_live = 0;
        for (int t = 0; t < n_tiles; ++t) max_live = std::max(max_live, panel_ranges[2 * t + 1] != 0 ? (int)panel_ranges[2 * t + 1] - (int)panel_ranges[2 * t] : K_pad / DL_CG_KP);
    }
    if (wfrag != nullptr && dl_options().chi2_bfrag && max_live <= DL_CG_PMAX) {
        // the B operand in fragment order straight into registers: LDS holds the rows of A only (dl_chi2_gemm_tile_bf)
        const int kl = k_live > 0 ? k_live : K_pad;
        // (three panel buffers of mt rows; at least the reduction area of the tile's end: [waves][2][4][64] doubles)
        const size_t shm = std::max((size_t)DL_CG_NBUF * mt * DL_CG_LD * 8, (size_t)DL_CG_WAVES * 2 * 4 * 64 * 8);
        auto go = [&](auto kernel, double* rs, int64_t lr) {
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            DL_LAUNCH(kernel, dim3(grid), dim3(64 * DL_CG_WAVES), shm, stream, A, lda, wfrag, bias, part, (int)M, K_pad, n_tiles, fin, panels, kl, rs, lr);
        };
        if (resid != nullptr) { if (mt == 16) go(dl_chi2_gemm_bf_kernel<true, true, 16, true>, resid, ldr); else go(dl_chi2_gemm_bf_kernel<true, true, DL_CG_M, true>, resid, ldr); }
        else { if (mt == 16) go(dl_chi2_gemm_bf_kernel<true, true, 16, false>, no_resid, (int64_t)0); else go(dl_chi2_gemm_bf_kernel<true, true, DL_CG_M, false>, no_resid, (int64_t)0); }
    } else if (resid != nullptr) {
        if (mt == 16) DL_LAUNCH((dl_chi2_gemm_kernel<true, true, 16, true>), dim3(grid), dim3(64 * DL_CG_WAVES), DL_CG_NBUF * (16 + DL_CG_N) * DL_CG_LD * 8, stream, A, lda, Wt, ldw, bias, part, (int)M, K_pad, n_tiles, fin, panels, k_live > 0 ? k_live : K_pad, resid, ldr);
        else DL_LAUNCH((dl_chi2_gemm_kernel<true, true, DL_CG_M, true>), dim3(grid), dim3(64 * DL_CG_WAVES), DL_CG_LDS_BYTES, stream, A, lda, Wt, ldw, bias, part, (int)M, K_pad, n_tiles, fin, panels, k_live > 0 ? k_live : K_pad, resid, ldr);
    } else if (mt == 16) DL_LAUNCH((dl_chi2_gemm_kernel<true, true, 16>), dim3(grid), dim3(64 * DL_CG_WAVES), DL_CG_NBUF * (16 + DL_CG_N) * DL_CG_LD * 8, stream, A, lda, Wt, ldw, bias, part, (int)M, K_pad, n_tiles, fin, panels, k_live > 0 ? k_live : K_pad, no_resid, (int64_t)0);
    else DL_LAUNCH((dl_chi2_gemm_kernel<true, true>), dim3(grid), dim3(64 * DL_CG_WAVES), DL_CG_LDS_BYTES, stream, A, lda, Wt, ldw, bias, part, (int)M, K_pad, n_tiles, fin, panels, k_live > 0 ? k_live : K_pad, no_resid, (int64_t)0);
    if (fin.stamps) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * 8);
        (void)hipMemcpy(h.data(), fin.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_file, "a")) {
            for (unsigned w = 0; w < grid; ++w) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", h[(size_t)w * 8 + q]); fprintf(f, "\n"); }
            fprintf(f, "#\n");
            fclose(f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ONE LAUNCH PER STEP (BASELINE configs[1]: one Kaiser-type observable in its fast instantiation, plain likelihood, N_pad = 128): theory, chi2 GEMM and finalize of a batch
// of up to 1024 points by one grid of 1024-thread workgroups, one per CU.  The three launches of the step each pay a ramp, a cold L2 and a drain (about 7 of 25 us);
// here the producers of a row block hand it to its consumers inside the launch:
//   workgroup L = xcd + 8 (8 q + c) (workgroups are dealt round-robin to the 8 XCDs) first evaluates the FOUR points 32 mb + 4 c .. + 3 of row block mb = xcd + 8 q
//   (four 256-thread sub-groups running dl_fullshape_body side by side: what four workgroups per CU do in dl_fullshape_kernel), publishes them (the rows are written by
//   agent-scope write-through stores: performed once vmcnt has drained; then ONE memory-side atomic increment of ready[mb]), then becomes the chi2-GEMM workgroup of tile
//   (row block mb, column block c): it waits until the eight producers of mb have arrived (they are the eight workgroups with the same (xcd, q): the same XCD, whose L2
//   then holds the rows), runs dl_chi2_gemm_tile, and the last column block of a row block to arrive sums the partials and writes the outputs (the fused finalize of
//   dl_chi2_gemm.h).  Correctness does not rest on the XCD mapping (every hand-over is an agent-scope access; a launch starts with clean caches), only the traffic does.
//   No deadlock: a workgroup waits only for workgroups of its own aligned group of 64, all of which are dispatched before or with it (in-order dispatch, 256 CUs);
//   the wait is bounded all the same (it gives up after ~0.1 s and poisons its outputs with NaN rather than hang the device).  `ready` counts up: launch number `epoch`
//   of a context waits for 8 epoch.
// ------------------------------------------------------------------------------------------------
template <int NL>
__global__ __launch_bounds__(1024) void dl_step_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, double* __restrict__ power, int64_t ld_power,
                                                       const double* __restrict__ Wt, int64_t ldw, const double* __restrict__ bias, double* __restrict__ part, int M, int K_pad,
                                                       DlChi2Fin fin, DlChi2Panels panels, int k_live, int32_t* __restrict__ ready, int32_t target, int lds_per_point) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int q = j >> 3, c = j & 7, mb = xcd + 8 * q;
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    unsigned long long* st = fin.stamps != nullptr ? fin.stamps + (size_t)blockIdx.x * 8 : nullptr;   // DL_STEP_STAMPS: 0 entry, 1 theory done, 2 published, 3 rows ready, 4 GEMM + finalize done
    if (st != nullptr && threadIdx.x == 0) { st[0] = __builtin_amdgcn_s_memtime(); st[6] = __builtin_amdgcn_s_memrealtime(); }
    dl_fullshape_body<true, NL, false, false, false, true>(o, theta, n_params, power, ld_power, nullptr, 0, 0, nullptr, nullptr, lds + (size_t)sub * lds_per_point, tid, 32 * mb + 4 * c + sub);
    if (st != nullptr && threadIdx.x == 0) st[1] = __builtin_amdgcn_s_memtime();
    // the rows of this workgroup's four points are performed at agent scope once every wave's stores have drained; then they are published
    __asm__ volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ready + mb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st != nullptr && threadIdx.x == 0) st[2] = __builtin_amdgcn_s_memtime();
    DlChi2Fin f2 = fin;
    f2.stamps = nullptr;
    dl_chi2_gemm_tile<true, true, DL_CG_M, false>(power, ld_power, Wt, ldw, bias, part, M, K_pad, 8, f2, panels, k_live, nullptr, 0, lds, mb, c, [&]() {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(ready + mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __asm__ volatile("" : : : "memory");
        if (st != nullptr && threadIdx.x == 0) st[3] = __builtin_amdgcn_s_memtime();
    });
    if (st != nullptr && threadIdx.x == 0) { st[4] = __builtin_amdgcn_s_memtime(); st[7] = __builtin_amdgcn_s_memrealtime(); }
}

// eligibility of the one-launch step (dl_api.hip asks before every call): returns the dynamic LDS size, or 0
size_t dl_step_lds_bytes(const DlObsDev& oh, int64_t B, int N_pad) {
    const bool generic = !oh.uniform_knots || !(oh.toeplitz || oh.fixed_spline);
    if (oh.theory >= 2 || generic || oh.n_ct > 0 || oh.n_sn > 0 || oh.n_ell > 3 || oh.n_pass != 0 || oh.n_var != 0 || N_pad != 128) return 0;
    if (B <= 0 || B > 1024 || B % 256 != 0) return 0;     // whole groups of 8 row blocks: 64 workgroups
    const size_t per_point = (dl_fs_shared_doubles_obs(oh, true) + 1) / 2 * 2;
    const size_t bytes = std::max<size_t>(4 * per_point * sizeof(double), DL_CG_LDS_BYTES);
    return bytes <= 160 * 1024 ? bytes : 0;
}

void dl_launch_step(const DlObsDev& oh, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power, const double* Wt, int64_t ldw, const double* bias,
                    double* part, int K_pad, int k_live, int32_t* counters, int32_t* ready, int32_t target, const double* priors, double* loglike, double* logprior, int32_t* status,
                    int post_mode, hipStream_t stream, const uint8_t* panel_ranges) {
    DlChi2Panels panels;
    std::memset(&panels, 0, sizeof(panels));
    if (panel_ranges != nullptr) for (int t = 0; t < 8; ++t) panels.range[t] = (uint32_t)panel_ranges[2 * t] | ((uint32_t)panel_ranges[2 * t + 1] << 8);
    const size_t per_point = (dl_fs_shared_doubles_obs(oh, true) + 1) / 2 * 2;
    const size_t shm = dl_step_lds_bytes(oh, B, 128);
    static bool optin = false;
    if (!optin) { (void)hipFuncSetAttribute((const void*)dl_step_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); optin = true; }
    DlChi2Fin fin;
    fin.counters = counters; fin.theta = theta; fin.priors = priors; fin.loglike = loglike; fin.logprior = logprior; fin.status = status;
    fin.n_params = n_params; fin.post_mode = post_mode; fin.ready = ready;
    const unsigned grid = (unsigned)(B / 4);
    static const char* stamp_file = getenv("DL_STEP_STAMPS");   // diagnostics: in-kernel timestamps of launches 30..33 appended to the file (synchronises)
    static unsigned long long* stamps_dev = nullptr;
    static int stamp_launches = 0;
    if (stamp_file && !stamps_dev) (void)hipMalloc((void**)&stamps_dev, (size_t)256 * 8 * sizeof(unsigned long long));
    fin.stamps = (stamp_file && stamp_launches >= 30 && stamp_launches < 34) ? stamps_dev : nullptr;
    if (stamp_file) stamp_launches++;
    if (fin.stamps) (void)hipMemsetAsync(fin.stamps, 0, (size_t)256 * 8 * sizeof(unsigned long long), stream);
    DL_LAUNCH(dl_step_kernel<3>, dim3(grid), dim3(1024), shm, stream, oh, theta, n_params, power, ld_power, Wt, ldw, bias, part, (int)B, K_pad, fin, panels, k_live > 0 ? k_live : K_pad,
              ready, target, (int)per_point);
    if (fin.stamps) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * 8);
        (void)hipMemcpy(h.data(), fin.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_file, "a")) {
            for (unsigned w = 0; w < grid; ++w) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", h[(size_t)w * 8 + q]); fprintf(f, "\n"); }
            fprintf(f, "#\n");
            fclose(f);
        }
    }
}

// one thread per point; the prior table is staged in LDS (one cooperative load whose round trip overlaps those of the thread's own partial sums and theta row:
// read through uniform scalar loads it was three dependent round trips per parameter)
__global__ __launch_bounds__(64) void dl_finalize_part_kernel(const double* __restrict__ part, int n_tiles, const double* __restrict__ theta, int n_params,
                                                               const double* __restrict__ priors, int64_t B, double* __restrict__ loglike,
                                                               double* __restrict__ logprior, int32_t* __restrict__ status, int post_mode) {
    extern __shared__ __attribute__((aligned(16))) double dl_fp_priors[];   // [n_params, 5]
    dl_kernarg_prefetch<96>();
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t bb = b < B ? b : B - 1;
    // every global load of the thread is requested before the first wait: prior table entries, partial sums, parameter values
    double pv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) pv[q] = priors[(int)threadIdx.x + q * 64 < 5 * n_params ? (int)threadIdx.x + q * 64 : 0];
    double x0[8];
    dl_load_theta8(theta + (size_t)bb * n_params, n_params, 0, x0);
    const double chi2 = dl_chi2_of_parts(part + (size_t)bb * n_tiles, n_tiles);
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if ((int)threadIdx.x + q * 64 < 5 * n_params) dl_fp_priors[threadIdx.x + q * 64] = pv[q];
    for (int e = threadIdx.x + 128; e < 5 * n_params; e += 64) dl_fp_priors[e] = priors[e];   // (more than 25 parameters)
    __syncthreads();
    if (b >= B) return;
    const double inf = __builtin_huge_val();
    double ll, lp;
    int st;
    dl_finalize_from_chi2(chi2, x0, theta + (size_t)b * n_params, n_params, dl_fp_priors, ll, lp, st);
    if (loglike) loglike[b] = post_mode ? (st == DL_ST_OK ? ll + lp : -inf) : ll;
    if (logprior) logprior[b] = lp;
    if (status) status[b] = st;
}

void dl_launch_finalize_part(const double* part, int n_tiles, const double* theta, int n_params, const double* priors, int64_t B, double* loglike, double* logprior,
                             int32_t* status, int post_mode, hipStream_t stream) {
    const size_t shm = (size_t)5 * n_params * sizeof(double);
    if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_finalize_part_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    DL_LAUNCH(dl_finalize_part_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), shm, stream, part, n_tiles, theta, n_params, priors, B, loglike, logprior, status, post_mode);
}

// ---- lane-parallel dense algebra for the <= 15 x 15 systems of the marginalised finalize: lane i owns row i in registers, rows are exchanged with
// v_readlane (uniform source lane).  The first version did this serially on lane 0 with the matrices in scratch memory: 90 us per 4096 points.
__device__ __forceinline__ double dl_readlane(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// the same for FOUR systems per wavefront (lanes 16 q .. 16 q + 15 hold system q): the value of lane j of the caller's own group of 16 (ds_bpermute)
// (DPP row_newbcast: lane j of every row of 16 lanes to the whole row, at register speed -- j is a compile-time constant after unrolling; ds_bpermute, the
//  generic shuffle, costs an LDS round trip per exchange on the dependent chain of the factorisation)
__device__ __forceinline__ double dl_grouplane(double v, int j) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (j) {
        case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 0, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 0, 0xf, 0xf, false); break;
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 1, 0xf, 0xf, false); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 2, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 2, 0xf, 0xf, false); break;
        case 3: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 3, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 3, 0xf, 0xf, false); break;
        case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 4, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 4, 0xf, 0xf, false); break;
        case 5: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 5, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 5, 0xf, 0xf, false); break;
        case 6: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 6, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 6, 0xf, 0xf, false); break;
        case 7: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 7, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 7, 0xf, 0xf, false); break;
        case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 8, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 8, 0xf, 0xf, false); break;
        case 9: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 9, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 9, 0xf, 0xf, false); break;
        case 10: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 10, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 10, 0xf, 0xf, false); break;
        case 11: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 11, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 11, 0xf, 0xf, false); break;
        case 12: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 12, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 12, 0xf, 0xf, false); break;
        case 13: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 13, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 13, 0xf, 0xf, false); break;
        case 14: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 14, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 14, 0xf, 0xf, false); break;
        case 15: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 15, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 15, 0xf, 0xf, false); break;
        default: break;
    }
    return __hiloint2double(hi, lo);
}

// Cholesky A = C C^T of the SPD matrix whose row `lane` is a[0 .. 15] (only the leading m x m block matters; rows >= m must be unit vectors).
// On return a[] holds row `lane` of C (lower part) and invc[j] = 1 / C[j][j] (uniform); returns log det A; ok = false if a pivot is not positive.
// The pivots form one dependent chain: per step only a reciprocal square root sits on it (no division, no logarithm -- the log-determinant is taken
// afterwards, one pivot per lane in parallel).
// PACK: four systems per wavefront, `lane` = lane within the group of 16 (0 .. 15), rows exchanged with dl_grouplane; the returns are then per group.
template <bool PACK = false>
__device__ __forceinline__ double dl_lane_cholesky(double (&a)[16], double (&invc)[16], int m, int lane, bool& ok) {
    double dmine = 1.;   // pivot of this lane's own row
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        invc[j] = 1.;
        if (j < m) {
            double djj = PACK ? dl_grouplane(a[j], j) : dl_readlane(a[j], j);
            if (!(djj > 0.)) { ok = false; djj = 1.; }
            const double inv = 1. / sqrt(djj);
            invc[j] = inv;
            if (lane == j) dmine = djj;
            const double cij = (lane > j) ? a[j] * inv : (lane == j ? djj * inv : 0.);
            a[j] = cij;
#pragma unroll
            for (int k = j + 1; k < 16; ++k) a[k] -= cij * (PACK ? dl_grouplane(cij, k) : dl_readlane(cij, k));   // A[i][k] -= C[i][j] C[k][j]
        }
    }
    const double lg = log(dmine);   // log det A = sum_j log d_jj
    double logdet = 0.;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < m) logdet += PACK ? dl_grouplane(lg, j) : dl_readlane(lg, j);
    return logdet;
}

// ------------------------------------------------------------------------------------------------
// finalize with analytic marginalisation / best fit of n_s linear parameters (likelihoods/base.py:129-200, 314-413),
// one wavefront per point.  In whitened variables (dt = L^T Delta, Tt_s = L^T dDelta/dx_s):
//   H_L = -Tt Tt^T, g_L = -Tt dt, H = H_L - diag(prec), g = g_L - (x0 - loc) prec, dx = -H^-1 g,
//   loglike = -1/2 |dt|^2 + 1/2 dx H_L dx + g_L dx - 1/2 logdet(-H[marg, marg]),  logprior += sum -1/2 (x0 + dx - loc)^2 prec.
// Tt_s = tconst[s] (+ row 1 + var_slot[s] of the point when the derivative depends on the point).
// ------------------------------------------------------------------------------------------------
// LANES: lane-parallel algebra (n_s <= 15, no scratch memory); otherwise the serial fallback (n_s = 16) is compiled
template <bool LANES, bool STAGED>
__global__ __launch_bounds__(256, 4) void dl_finalize_marg_kernel(const double* __restrict__ dtilde, int64_t ld, int n, int rows_per_point, int n_slabs, int64_t slab_stride,
                                                               const double* __restrict__ bias, DlMargDev mg,
                                                               const double* __restrict__ theta, int n_params, const double* __restrict__ priors, int64_t B,
                                                               double* __restrict__ loglike, double* __restrict__ logprior, int32_t* __restrict__ status,
                                                               double* __restrict__ solved, double* __restrict__ hessian, int post_mode,
                                                               unsigned long long* __restrict__ stamps, const double* __restrict__ gram) {
    // gram != nullptr: G [B, 16, 16] was formed by the feature GEMM's epilogue (dl_feature_gemm.h): no residual rows to read, no Gram product here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#define DL_FM_STAMP(slot) if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();
    DL_FM_STAMP(0)
    if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (post_mode & 0x100) {
        // residual rows written by a 16-point workgroup of the feature GEMM (workgroup t on XCD t % 8): read them on the XCD that wrote them
        // (this workgroup w = xcd + 8 r takes quarter r % 4 of tile xcd + 8 (r / 4); B a multiple of 128)
        const int64_t w = blockIdx.x, xcd = w & 7, r = w >> 3;
        b = 16 * (xcd + 8 * (r >> 2)) + 4 * (r & 3) + wave;
    }
    post_mode &= 0xff;
    bool active = b < B;
    if (!active) b = B - 1;   // spare waves of the last workgroup recompute the last point (the barrier below is common) and store nothing
    const int ns = mg.n_s;
    const double* row0 = dtilde + (size_t)b * rows_per_point * ld;
    double chi2 = 0.;
    double HL[DL_MAX_SOLVED * (DL_MAX_SOLVED + 1) / 2];   // lower triangle of -H_L = Tt Tt^T (lane-uniform)
    double gL[DL_MAX_SOLVED];                              // Tt dt = -g_L
    // priors of the sampled parameters: the loads are issued first so that their round trip overlaps the staging of the Gram operands
    double lp = 0.;
    int nan_in = 0;
    const double inf = __builtin_huge_val();
    for (int p = lane; p < n_params; p += 64) {
        double x = theta[(size_t)b * n_params + p];
        const double* pr = priors + 5 * p;
        if (x != x) nan_in = 1;
        lp += dl_prior_logpdf(pr, x);
    }
    __shared__ double gram_lds[4][STAGED ? 2 : 16 * 16];   // (staged variant: G and the Cholesky rows reuse the wave's staging area)
    __shared__ double pk_lp[4];       // hand-over of the waves' points to the wavefront that solves all four (LANES)
    __shared__ long long pk_b[4];
    __shared__ int pk_flags[4];
    double* Gw = gram_lds[wave];
    if (LANES) {
        // Gram matrix of X = [dt; Tt_1 .. Tt_ns] (1 + ns <= 16 rows, n columns) with v_mfma_f64_16x16x4_f64: the A operand of lane l is X[l & 15][4 k + (l >> 4)]
        // and the B operand X^T[4 k + (l >> 4)][l & 15] -- the same register.  chi2 = G[0][0], Tt dt = G[0][1 + s], Tt Tt^T = G[1 + s][1 + t].
        const int xr = lane & 15, g = lane >> 4;
        const double* cptr = nullptr;   // point-independent part of row xr (bias / tconst)
        const double* vptr = nullptr;   // point-dependent part (rows of this point in the residual buffer)
        if (xr == 0) { cptr = bias; vptr = row0; }
        else if (xr <= ns) {
            cptr = mg.tconst + (size_t)(xr - 1) * ld;
            int vs = -1;
#pragma unroll
            for (int s = 0; s < DL_MAX_SOLVED; ++s) if (s == xr - 1) vs = mg.var_slot[s];
            if (vs >= 0) vptr = row0 + (size_t)(1 + vs) * ld;
        }
        dl_double4 acc = {0., 0., 0., 0.};
        const int n_ks = (n + 3) / 4;
        if (gram != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Gw[lane + 64 * q] = gram[(size_t)b * 256 + lane + 64 * q];
        } else if (STAGED) {
            // Rows of X staged in LDS by coalesced 16-byte loads, ALL in flight at once (one memory round trip instead of one per batch of k-steps:
            // the direct operand loads below touch 16 different rows per instruction and were 12.7 us of a 20 us workgroup life), constant and point-dependent
            // parts added on the way in; the MFMA operands then come from LDS.  Per wave: (1 + ns) rows of `stride` doubles.
            extern __shared__ __attribute__((aligned(16))) double dl_fm_dyn[];
            const int rows = 1 + ns, n4 = 4 * n_ks, stride = n4 + 4;
            const int region = rows * stride > 512 ? rows * stride : 512;   // doubles per wave: the staged rows, then G [16][16] | Cholesky rows [16][16] in the same place
            double* X = dl_fm_dyn + (size_t)wave * region;
            Gw = X;
            for (int c0 = 2 * lane; c0 < n4; c0 += 128) {
#pragma unroll 4
                for (int r = 0; r < rows; ++r) {
                    const double* cp = (r == 0) ? bias : mg.tconst + (size_t)(r - 1) * ld;
                    const int vs = (r == 0) ? 0 : mg.var_slot[r - 1];
                    const double* vp = (r == 0) ? row0 : (vs >= 0 ? row0 + (size_t)(1 + vs) * ld : nullptr);
                    dl_double2 v = {0., 0.};
                    if (cp) v = *reinterpret_cast<const dl_double2*>(cp + c0);
                    if (vp) for (int sl = 0; sl < n_slabs; ++sl) v += *reinterpret_cast<const dl_double2*>(vp + (size_t)sl * slab_stride + c0);
                    if (c0 >= n) v.x = 0.;
                    if (c0 + 1 >= n) v.y = 0.;
                    *reinterpret_cast<dl_double2*>(X + (size_t)r * stride + c0) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool live = xr <= ns;
            const double* xrow = X + (size_t)(live ? xr : 0) * stride + g;
            for (int k0 = 0; k0 < n_ks; k0 += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = (live && k0 + u < n_ks) ? xrow[4 * (k0 + u)] : 0.;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], x[u], acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();   // every operand read of this wave precedes the overwrite of the staging area by G
        } else
        for (int k0 = 0; k0 < n_ks; k0 += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {   // eight k-steps of independent loads in flight (the kernel is bound by these round trips)
                const int col = 4 * (k0 + u) + g;
                double v = 0.;
                if (col < n) {
                    if (cptr) v = cptr[col];
                    if (vptr) for (int sl = 0; sl < n_slabs; ++sl) v += vptr[(size_t)sl * slab_stride + col];
                }
                x[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], x[u], acc, 0, 0, 0);
        }
        DL_FM_STAMP(1)
        // C layout: register r of lane l = G[(l >> 4) + 4 r][l & 15]
        if (gram == nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Gw[(g + 4 * r) * 16 + xr] = acc[r];
        }
        // The solve below runs the FOUR points of the workgroup in ONE wavefront, 16 lanes each (it needs 16 lanes per point: with a wavefront per point three
        // quarters of every instruction were idle lanes, and the kernel was bound by the issue of those instructions): what the other waves know goes through LDS
        lp = dl_wave_sum(lp);
        nan_in = __any(nan_in);
        if (lane == 0) { pk_lp[wave] = lp; pk_b[wave] = (long long)b; pk_flags[wave] = (nan_in ? 1 : 0) | (active ? 2 : 0); }
        __syncthreads();
        DL_FM_STAMP(2)
        if (wave != 0) return;
    } else if (!LANES) {
    // per-lane slices of dt and of every Tt_s (n <= 64 * DL_MARG_NJ)
    double dj[DL_MARG_NJ];
#pragma unroll
    for (int q = 0; q < DL_MARG_NJ; ++q) {
        int j = lane + 64 * q;
        double v0 = 0.;
        if (j < n) {
            v0 = bias ? bias[j] : 0.;
            for (int sl = 0; sl < n_slabs; ++sl) v0 += row0[(size_t)sl * slab_stride + j];
        }
        dj[q] = v0;
        chi2 = fma(dj[q], dj[q], chi2);
    }
    chi2 = dl_wave_sum(chi2);
    for (int s = 0; s < ns; ++s) {
        double ts[DL_MARG_NJ];
#pragma unroll
        for (int q = 0; q < DL_MARG_NJ; ++q) {
            int j = lane + 64 * q;
            double v = 0.;
            if (j < n) {
                v = mg.tconst[(size_t)s * ld + j];
                if (mg.var_slot[s] >= 0)
                    for (int sl = 0; sl < n_slabs; ++sl) v += row0[(size_t)sl * slab_stride + (size_t)(1 + mg.var_slot[s]) * ld + j];
            }
            ts[q] = v;
        }
        double acc = 0.;
#pragma unroll
        for (int q = 0; q < DL_MARG_NJ; ++q) acc = fma(ts[q], dj[q], acc);
        gL[s] = dl_wave_sum(acc);
        for (int t = 0; t <= s; ++t) {
            double a2 = 0.;
#pragma unroll
            for (int q = 0; q < DL_MARG_NJ; ++q) {
                int j = lane + 64 * q;
                double v = 0.;
                if (j < n) {
                    v = mg.tconst[(size_t)t * ld + j];
                    if (mg.var_slot[t] >= 0)
                        for (int sl = 0; sl < n_slabs; ++sl) v += row0[(size_t)sl * slab_stride + (size_t)(1 + mg.var_slot[t]) * ld + j];
                }
                a2 = fma(ts[q], v, a2);
            }
            HL[s * (s + 1) / 2 + t] = dl_wave_sum(a2);
        }
    }
    }
    if (!LANES) { lp = dl_wave_sum(lp); nan_in = __any(nan_in); }
    if (LANES) {
        // ---- lane-parallel solve, four points per wavefront: lanes 16 q + i <-> point q of the workgroup, solved parameter i ----
        const int q = lane >> 4, li = lane & 15;
        __shared__ double chol_lds[4][STAGED ? 2 : 16 * 16];
        const double* G;
        double* Cm;
        if (STAGED && gram == nullptr) {
            extern __shared__ __attribute__((aligned(16))) double dl_fm_dyn[];
            const int rows = 1 + ns, n4 = 4 * ((n + 3) / 4), stride = n4 + 4;
            const int region = rows * stride > 512 ? rows * stride : 512;
            G = dl_fm_dyn + (size_t)q * region; Cm = dl_fm_dyn + (size_t)q * region + 256;
        } else { G = gram_lds[q]; Cm = chol_lds[q]; }
        lp = pk_lp[q]; b = (int64_t)pk_b[q];
        const bool nan_q = (pk_flags[q] & 1) != 0;
        active = (pk_flags[q] & 2) != 0;
        double prec_i = 0., x0_i = 0., loc_i = 0.;
        int marg_i = 0;
#pragma unroll
        for (int s = 0; s < DL_MAX_SOLVED; ++s) if (s == li) { prec_i = mg.prec[s]; x0_i = mg.x0[s]; loc_i = mg.loc[s]; marg_i = mg.is_marg[s]; }
        const bool mine = li < ns;
        // A = -H = Tt Tt^T + diag(prec) (SPD); rhs = g = -(Tt dt) - (x0 - loc) prec; dx = A^-1 g
        double a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = (mine && k < ns) ? G[(1 + li) * 16 + 1 + k] + (k == li ? prec_i : 0.) : (k == li ? 1. : 0.);
        double gi = mine ? -G[1 + li] - (x0_i - loc_i) * prec_i : 0.;
        bool ok = true;
        DL_FM_STAMP(3)
        double invc[16];
        const double logdet_all = dl_lane_cholesky<true>(a, invc, ns, li, ok);
        DL_FM_STAMP(4)
        // forward substitution C y = g
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < ns) {
                const double yj = dl_grouplane(gi, j) * invc[j];
                if (li > j) gi -= a[j] * yj; else if (li == j) gi = yj;
            }
        }
        // backward substitution C^T dx = y needs column entries C[j][i]: rows go through LDS (same wave: LDS operations of a wave execute in order)
#pragma unroll
        for (int k = 0; k < 16; ++k) Cm[li * 16 + k] = a[k];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            if (j < ns) {
                const double xj = dl_grouplane(gi, j) * invc[j];
                if (li < j) gi -= Cm[j * 16 + li] * xj; else if (li == j) gi = xj;
            }
        }
        const double dxi = mine ? gi : 0.;
        // 1/2 dx H_L dx + g_L dx  (likelihoods/base.py:385-386), H_L = -Tt Tt^T, g_L = -Tt dt
        double rowdot = 0.;
#pragma unroll
        for (int t = 0; t < 16; ++t) if (t < ns) rowdot += (mine ? G[(1 + li) * 16 + 1 + t] : 0.) * dl_grouplane(dxi, t);
        const double xs = x0_i + dxi;
        const double quad_i = dxi * rowdot, lin_i = mine ? G[1 + li] * dxi : 0.;
        const double lps_i = mine ? -0.5 * (xs - loc_i) * (xs - loc_i) * prec_i : 0.;   // 363-364 with parameter.py:2007 (0 for flat priors: prec = 0)
        double quad = 0., lin = 0., lps = 0.;
#pragma unroll
        for (int t = 0; t < 16; ++t) if (t < ns) { quad += dl_grouplane(quad_i, t); lin += dl_grouplane(lin_i, t); lps += dl_grouplane(lps_i, t); }   // fixed order
        if (mine && active && solved) solved[(size_t)b * ns + li] = xs;
        if (mine && active && hessian) {   // likelihood Hessian H_L = -Tt Tt^T w.r.t. the solved parameters (derived output, likelihoods/base.py:388-390)
#pragma unroll
            for (int k = 0; k < 16; ++k) if (k < ns) hessian[((size_t)b * ns + li) * ns + k] = -G[(1 + li) * 16 + 1 + k];
        }
        double ll = -0.5 * G[0] - 0.5 * quad - lin;
        // -1/2 logdet(-H[marg, marg]) (394-404); all-marg: reuse the Cholesky above, else factor the compacted sub-block
        if (mg.n_marg == ns) ll -= 0.5 * logdet_all;
        else if (mg.n_marg > 0) {
            int pos_i = 0;
#pragma unroll
            for (int s = 0; s < DL_MAX_SOLVED; ++s) if (s < li && s < ns && mg.is_marg[s]) pos_i++;
            __builtin_amdgcn_wave_barrier();
            if (mine && marg_i) {
                int pos_k = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < ns && mg.is_marg[k]) { Cm[pos_i * 16 + pos_k] = G[(1 + li) * 16 + 1 + k] + (k == li ? prec_i : 0.); pos_k++; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int nm = mg.n_marg;
            double sub[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) sub[k] = (li < nm && k < nm) ? Cm[li * 16 + k] : (k == li ? 1. : 0.);
            double invs[16];
            ll -= 0.5 * dl_lane_cholesky<true>(sub, invs, nm, li, ok);
        }
        if (li == 0 && active) {
            const double lptot = lp + lps;
            int st = DL_ST_OK;
            if (nan_q) st = DL_ST_NAN_INPUT;
            else if (lp == -inf) st = DL_ST_OUT_OF_PRIOR;
            else if (!ok || !(ll == ll) || ll == inf || ll == -inf) st = DL_ST_NONFINITE;
            if (loglike) loglike[b] = post_mode ? (st == DL_ST_OK ? ll + lptot : -inf) : ll;
            if (logprior) logprior[b] = lptot;
            if (status) status[b] = st;
        }
        DL_FM_STAMP(5)
        if (stamps != nullptr && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    if (!LANES && lane == 0 && active) {
        // A = -H = Tt Tt^T + diag(prec) (SPD); rhs = g = -(Tt dt) - (x0 - loc) prec; dx = A^-1 g
        double A[DL_MAX_SOLVED][DL_MAX_SOLVED], g[DL_MAX_SOLVED], dx[DL_MAX_SOLVED];
        for (int s = 0; s < ns; ++s) {
            for (int t = 0; t <= s; ++t) A[s][t] = A[t][s] = HL[s * (s + 1) / 2 + t];
            A[s][s] += mg.prec[s];
            g[s] = -gL[s] - (mg.x0[s] - mg.loc[s]) * mg.prec[s];
        }
        bool ok = true;
        double logdet_all = 0.;
        // Cholesky A = C C^T (in place, lower)
        for (int j = 0; j < ns; ++j) {
            double d = A[j][j];
            for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
            if (!(d > 0.)) { ok = false; d = 1.; }
            d = sqrt(d);
            A[j][j] = d;
            logdet_all += 2. * log(d);
            for (int i = j + 1; i < ns; ++i) {
                double sum = A[i][j];
                for (int k = 0; k < j; ++k) sum -= A[i][k] * A[j][k];
                A[i][j] = sum / d;
            }
        }
        for (int i = 0; i < ns; ++i) {   // forward, backward substitution
            double sum = g[i];
            for (int k = 0; k < i; ++k) sum -= A[i][k] * dx[k];
            dx[i] = sum / A[i][i];
        }
        for (int i = ns - 1; i >= 0; --i) {
            double sum = dx[i];
            for (int k = i + 1; k < ns; ++k) sum -= A[k][i] * dx[k];
            dx[i] = sum / A[i][i];
        }
        // 1/2 dx H_L dx + g_L dx  (likelihoods/base.py:385-386), H_L = -HL, g_L = -gL
        double quad = 0., lin = 0., lps = 0.;
        for (int s = 0; s < ns; ++s) {
            double rowsum = 0.;
            for (int t = 0; t < ns; ++t) rowsum += HL[(s >= t) ? s * (s + 1) / 2 + t : t * (t + 1) / 2 + s] * dx[t];
            quad += dx[s] * rowsum;
            lin += gL[s] * dx[s];
            double xs = mg.x0[s] + dx[s];
            lps += -0.5 * (xs - mg.loc[s]) * (xs - mg.loc[s]) * mg.prec[s];   // 363-364 with parameter.py:2007 (0 for flat priors: prec = 0)
            if (solved) solved[(size_t)b * ns + s] = xs;
            if (hessian) for (int t = 0; t < ns; ++t) hessian[((size_t)b * ns + s) * ns + t] = -HL[(s >= t) ? s * (s + 1) / 2 + t : t * (t + 1) / 2 + s];
        }
        double ll = -0.5 * chi2 - 0.5 * quad - lin;
        // -1/2 logdet(-H[marg, marg]) (394-404); all-marg: reuse the Cholesky above, else factor the sub-block
        if (mg.n_marg == ns) ll -= 0.5 * logdet_all;
        else if (mg.n_marg > 0) {
            double S[DL_MAX_SOLVED][DL_MAX_SOLVED];
            int idx[DL_MAX_SOLVED], nm = 0;
            for (int s = 0; s < ns; ++s) if (mg.is_marg[s]) idx[nm++] = s;
            for (int a = 0; a < nm; ++a)
                for (int c = 0; c <= a; ++c) {
                    int s = idx[a], t = idx[c];
                    S[a][c] = HL[(s >= t) ? s * (s + 1) / 2 + t : t * (t + 1) / 2 + s] + (a == c ? mg.prec[s] : 0.);
                }
            double ld2 = 0.;
            for (int j = 0; j < nm; ++j) {
                double d = S[j][j];
                for (int k = 0; k < j; ++k) d -= S[j][k] * S[j][k];
                if (!(d > 0.)) { ok = false; d = 1.; }
                d = sqrt(d);
                S[j][j] = d;
                ld2 += 2. * log(d);
                for (int i = j + 1; i < nm; ++i) {
                    double sum = S[i][j];
                    for (int k = 0; k < j; ++k) sum -= S[i][k] * S[j][k];
                    S[i][j] = sum / d;
                }
            }
            ll -= 0.5 * ld2;
        }
        double lptot = lp + lps;
        int st = DL_ST_OK;
        if (nan_in) st = DL_ST_NAN_INPUT;
        else if (lp == -inf) st = DL_ST_OUT_OF_PRIOR;
        else if (!ok || !(ll == ll) || ll == inf || ll == -inf) st = DL_ST_NONFINITE;
        if (loglike) loglike[b] = post_mode ? (st == DL_ST_OK ? ll + lptot : -inf) : ll;
        if (logprior) logprior[b] = lptot;
        if (status) status[b] = st;
    }
}

// ---- the same finalize when the Gram matrix G [B, 16, 16] of X = [dt; Tt_1 .. Tt_ns] is already there (feature GEMM of the emulated path): ONE LANE PER POINT ----
// The solve of a point is a few hundred flops on an (ns x ns) system.  The lane-parallel kernel above gives it 16 lanes and pays a cross-lane exchange per
// elimination step on one dependent chain (10.6 us per 4096 points of config 3, 1024 workgroups for 150 kFLOP); here a lane keeps the lower triangle of its point
// in registers and runs Cholesky, the two substitutions and the quadratic forms serially (dl_marg_solve.h); 64 points per wavefront, the priors of the same points
// on a second wavefront beside it.
template <int NS>
__global__ __launch_bounds__(128) void dl_finalize_marg_gram_kernel(const double* __restrict__ gram, DlMargDev mg, const double* __restrict__ theta, int n_params,
                                                                      const double* __restrict__ priors, int64_t B, double* __restrict__ loglike, double* __restrict__ logprior,
                                                                      int32_t* __restrict__ status, double* __restrict__ solved, double* __restrict__ hessian, int post_mode) {
    const int lane = threadIdx.x & 63;
    const bool prior_wave = threadIdx.x >= 64;   // wave 1: the priors of the same 64 points, beside the solve (a quarter of the instructions of a point)
    __shared__ double lp_lds[64];
    __shared__ int nan_lds[64];
    int64_t b = (int64_t)blockIdx.x * 64 + lane;
    if (post_mode & 0x100) {
        // G of points 16 t .. 16 t + 15 was written by workgroup t of the feature GEMM, on XCD t % 8: workgroup w = xcd + 8 r reads tiles xcd + 8 (4 r + j), j = lane / 16
        const int64_t w = blockIdx.x, xcd = w & 7, r = w >> 3;
        b = 16 * (xcd + 8 * (4 * r + (lane >> 4))) + (lane & 15);
    }
    post_mode &= 0xff;
    const bool active = b < B;
    if (!active) b = B - 1;
    if (prior_wave) {
        double lp;
        int nan_in;
        dl_marg_priors_lane(theta + (size_t)b * n_params, n_params, priors, lp, nan_in);
        lp_lds[lane] = lp; nan_lds[lane] = nan_in;
        __syncthreads();
        return;
    }
    const double* G = gram + (size_t)b * 256;
    double lps;
    bool ok;
    const double ll = dl_marg_solve_lane<NS>([&](int i, int j) { return G[i * 16 + j]; }, mg, (active && solved) ? solved + (size_t)b * NS : nullptr,
                                             (active && hessian) ? hessian + (size_t)b * NS * NS : nullptr, lps, ok);
    __syncthreads();   // the priors of the other wave
    if (active) dl_marg_store_lane(ll, lps, ok, lp_lds[lane], nan_lds[lane] != 0, post_mode, b, loglike, logprior, status);
}

void dl_launch_finalize_marg(const double* dtilde, int64_t ld, int n, int rows_per_point, int n_slabs, int64_t slab_stride, const double* bias, const DlMargDev& mg,
                             const double* theta, int n_params, const double* priors, int64_t B, double* loglike, double* logprior, int32_t* status, double* solved,
                             double* hessian, int post_mode, hipStream_t stream, bool xcd_tile16, const double* gram) {
    static const char* stamp_file = getenv("DL_FM_STAMPS");   // diagnostics, see dl_launch_fullshape
    static unsigned long long* stamps_dev = nullptr;
    static int stamp_launches = 0;
    const unsigned grid = (unsigned)((B + 3) / 4);
    if (stamp_file && !stamps_dev) (void)hipMalloc((void**)&stamps_dev, (size_t)65536 * 8 * sizeof(unsigned long long));
    unsigned long long* stamps = (stamp_file && grid <= 65536 && B >= 256 && stamp_launches >= 10 && stamp_launches < 12) ? stamps_dev : nullptr;
    if (stamp_file && B >= 256) stamp_launches++;
    const int xcd_local = dl_options().xcd_local;
    if (xcd_local && xcd_tile16 && B % 128 == 0) post_mode |= 0x100;
    static const bool allow_staged = !getenv("DL_FM_NO_STAGE");   // DL_FM_NO_STAGE=1: operands of the Gram product straight from global memory (comparison)
    const size_t region = std::max<size_t>((size_t)(1 + mg.n_s) * (((n + 3) & ~3) + 4), 512);
    const size_t shm = 4 * region * sizeof(double);   // staged rows of the four waves (reused for G and the Cholesky rows)
    const bool lane_solve = !dl_options().fm_no_lane_solve;   // DL_FM_NO_LANE_SOLVE=1: the 16-lanes-per-point kernel also with a ready Gram matrix (read at every launch: the tests compare both in one process)
    if (gram != nullptr && lane_solve && mg.n_s >= 1 && mg.n_s <= 8) {
        const unsigned grid64 = (unsigned)((B + 63) / 64);
        const int mode = (xcd_local && xcd_tile16 && B % 512 == 0) ? ((post_mode & 0xff) | 0x100) : (post_mode & 0xff);
        auto launch = [&](auto kernel) { DL_LAUNCH(kernel, dim3(grid64), dim3(128), 0, stream, gram, mg, theta, n_params, priors, B, loglike, logprior, status, solved, hessian, mode); };
        switch (mg.n_s) {
            case 1: launch(dl_finalize_marg_gram_kernel<1>); break;
            case 2: launch(dl_finalize_marg_gram_kernel<2>); break;
            case 3: launch(dl_finalize_marg_gram_kernel<3>); break;
            case 4: launch(dl_finalize_marg_gram_kernel<4>); break;
            case 5: launch(dl_finalize_marg_gram_kernel<5>); break;
            case 6: launch(dl_finalize_marg_gram_kernel<6>); break;
            case 7: launch(dl_finalize_marg_gram_kernel<7>); break;
            default: launch(dl_finalize_marg_gram_kernel<8>); break;
        }
        return;
    }
    if (gram != nullptr) {   // (needs n_s < 16: checked by the caller)
        DL_LAUNCH((dl_finalize_marg_kernel<true, false>), dim3(grid), dim3(256), 0, stream, dtilde, ld, n, rows_per_point, n_slabs, slab_stride, bias, mg, theta,
                           n_params, priors, B, loglike, logprior, status, solved, hessian, post_mode, stamps, gram);
    } else if (mg.n_s < 16 && allow_staged && shm <= 96 * 1024) {
        if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)dl_finalize_marg_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        DL_LAUNCH((dl_finalize_marg_kernel<true, true>), dim3(grid), dim3(256), shm, stream, dtilde, ld, n, rows_per_point, n_slabs, slab_stride, bias, mg, theta,
                           n_params, priors, B, loglike, logprior, status, solved, hessian, post_mode, stamps, nullptr);
    } else if (mg.n_s < 16)
        DL_LAUNCH((dl_finalize_marg_kernel<true, false>), dim3(grid), dim3(256), 0, stream, dtilde, ld, n, rows_per_point, n_slabs, slab_stride, bias, mg, theta,
                           n_params, priors, B, loglike, logprior, status, solved, hessian, post_mode, stamps, nullptr);
    else
        DL_LAUNCH((dl_finalize_marg_kernel<false, false>), dim3(grid), dim3(256), 0, stream, dtilde, ld, n, rows_per_point, n_slabs, slab_stride, bias, mg, theta,
                           n_params, priors, B, loglike, logprior, status, solved, hessian, post_mode, stamps, nullptr);
    if (stamps) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * 8);
        (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_file, "a")) {
            for (unsigned w = 0; w < grid; ++w) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", h[(size_t)w * 8 + q]); fprintf(f, "\n"); }
            fprintf(f, "#\n");
            fclose(f);
        }
    }
}
