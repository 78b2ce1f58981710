// dl_kernels.h -- launchers of the gfx950 kernels (definitions in dl_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "dl_fullshape.h"

// Kernel-interval measurement (dl_profile_*): when dl_eval_* sets these events around a launcher, the kernel is launched with hipExtLaunchKernelGGL, which
// attaches them to the dispatch packet itself -- start / stop are the packet's own timestamps (what rocprofv3 --kernel-trace reads), no barrier packets
// are inserted between the kernels (an hipEventRecord between two launches costs 3.5-4.6 us of stream time).
struct DlProfEvents { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local DlProfEvents dl_prof_events;
#define DL_LAUNCH(kernel, grid, block, shm, stream, ...)                                                                                       \
    do {                                                                                                                                       \
        if (dl_prof_events.start) hipExtLaunchKernelGGL(kernel, grid, block, shm, stream, dl_prof_events.start, dl_prof_events.stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);                                                                \
    } while (0)

// Diagnostic switches of the library (environment variables DL_*: alternative kernels kept for comparison, forced variants -- include/desilike_amd.h lists them): read ONCE,
// at first use, into this table; nothing on the per-call path calls getenv.  dl_options_refresh() (C ABI; the tests flip switches inside one process) re-reads the environment.
struct DlOptions {
    bool no_merged_theory, no_emu_batch, no_fused_solve, no_gram_plain, no_scaled_row0, ef_no_early_theta, fm_no_lane_solve;
    bool chi2_bfrag;                    // DL_CHI2_BFRAG=1: the chi2 GEMM with its B operand in registers (dl_chi2_gemm_tile_bf: round-6 experiment, measured slower; docs/EXPERIMENTS.md)
    bool no_stk_split;                  // DL_NO_STK_SPLIT: the stacked engine in ONE launch (dl_emulated_stacked_kernel: networks and feature GEMMs group by group) instead of dl_stk_chain_kernel + dl_emulated_stacked_gemm_kernel
    int stk_overlap;                    // DL_STK_OVERLAP: the stacked engine on dl_emulated_stacked_ov_kernel where the shape allows (round-6 experiment, slower: docs/EXPERIMENTS.md):
                                        // 1 networks of the next batch under the feature GEMM (+2: without raised priority), 4 the two halves of the workgroup on half of the networks each
    bool ens_global, ens_force_comm, ens_no_defer, ens_no_fold, ens_stamps, ens_fold_stamps;
    bool no_emu_fused, no_gram_epilogue, step_kernel, chi2_fused, no_chi2_big;   // DL_NO_EMU_FUSED, DL_NO_GRAM_EPILOGUE, DL_STEP_KERNEL, DL_CHI2_FUSED, DL_NO_CHI2_BIG (dl_api.hip)
    int xcd_local;    // DL_XCD_LOCAL: 0 off, 1 chi2 GEMM path (default), 2 also the theory kernel's point order
    long long chi2_max_rows;   // DL_CHI2_GEMM_MAX (default 2048): above, the split-K / LDS-DMA GEMM paths
    int cg_mt;        // DL_CG_MT: forced row tile of the chi2 GEMM (0: chosen per batch)
    int host_mode;    // DL_HOST_MODE: see dl_eval_batch_host (-1: default)
};
const DlOptions& dl_options();

#define DL_ST_OK 0
#define DL_ST_OUT_OF_PRIOR 1
#define DL_ST_NONFINITE 2
#define DL_ST_NAN_INPUT 3

// obs_host: HOST array of observables whose pointers already point into device memory (passed by value to the kernel)
void dl_launch_fullshape(const DlObsDev* obs_host, int n_obs, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power, double* tables,
                         int64_t ld_tables, hipStream_t stream, double* feat = nullptr, int64_t feat_ld = 0, int xcd_block = 0,   // xcd_block: rows per row block of the consuming GEMM (0: points in launch order)
                         const DlObsDev* obs_dev = nullptr);   // obs_dev: the same observables as a DEVICE array (enables one launch for all of them)
// bias is added to rows r with r % bias_period == 0 only (bias_period = 1: every row)
void dl_launch_window_gemm(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* C, int64_t ldc, int64_t M, int N_valid, int N_pad,
                           int K_pad, int bias_period, hipStream_t stream);

#define DL_MARG_NJ 8   // finalize-marg keeps n / 64 <= 8 residual entries per lane: n_data <= 512
struct DlMargDev {
    int32_t n_s, n_marg;
    int32_t is_marg[DL_MAX_SOLVED];
    int32_t var_slot[DL_MAX_SOLVED];     // row (1 + slot) of the point holds the point-dependent part of Tt_s, or -1
    double x0[DL_MAX_SOLVED], loc[DL_MAX_SOLVED], prec[DL_MAX_SOLVED];
    const double* tconst;                // [n_s, N_pad] point-independent part of Tt_s = L^T W dpower/dx_s
};
void dl_launch_finalize_marg(const double* dtilde, int64_t ld, int n, int rows_per_point, int n_slabs, int64_t slab_stride, const double* bias, const DlMargDev& mg,
                             const double* theta, int n_params, const double* priors, int64_t B, double* loglike, double* logprior, int32_t* status, double* solved,
                             double* hessian /* [B, n_s, n_s] likelihood Hessian w.r.t. the solved parameters, may be null */, int post_mode, hipStream_t stream, bool xcd_tile16 = false,   // xcd_tile16: the rows were written by 16-point workgroups in launch order (feature GEMM)
                             const double* gram = nullptr);   // gram [B, 16, 16]: Gram matrix of [dt; Tt_1 .. Tt_ns] already formed by the feature GEMM's epilogue (dtilde is not read)
void dl_launch_transform(double* flat, int64_t ld, const double* data, const int32_t* transform, int n, int64_t B, hipStream_t stream);
// tiled split-K variant: writes n_splits partial slabs (no bias); N_pad multiple of 128, K_pad multiple of 16
int dl_gemm_tiled_splits(int64_t M, int N_pad, int K_pad, int* chunks_per_split);
void dl_launch_window_gemm_tiled(const double* A, int64_t lda, const double* Wt, int64_t ldw, double* slabs, int64_t slab_stride, int64_t ldc, int64_t M, int N_pad, int K_pad,
                                 int n_splits, int chunks_per_split, hipStream_t stream, int n_live = 0);
// residual of a row = bias (may be null) + sum of the n_slabs partial slabs (slab_stride doubles apart).
// post_mode (all finalize launchers): write the log-posterior (loglike + logprior, -inf unless status OK: samplers/base.py:185-191) to `loglike`
void dl_launch_finalize(const double* dtilde, int64_t ld, int n, int n_slabs, int64_t slab_stride, const double* bias, const double* theta, int n_params,
                        const double* priors, int64_t B, double* loglike, double* logprior, int32_t* status, int post_mode, hipStream_t stream);
// chi2 GEMM path: part[M, N_pad / 16] = partial chi2 per 16-column block (bias added inside), summed with the priors by dl_launch_finalize_part
// counters != nullptr ([ceil(M / 32) rounded up to 8] zeroed int32): the finalize (priors, status, outputs) is fused into the GEMM's last-arriving workgroups
int dl_chi2_gemm_row_tile(int64_t M, int N_pad);   // 32 or 16: rows per workgroup the chi2 GEMM will use for a batch of M points (the theory kernel deals the points to the XCDs accordingly)
void dl_launch_chi2_gemm(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* part, int64_t M, int N_pad, int K_pad, int32_t* counters,
                         const double* theta, int n_params, const double* priors, double* loglike, double* logprior, int32_t* status, int post_mode, hipStream_t stream,
                         const uint8_t* panel_ranges = nullptr, int k_live = 0, double* resid = nullptr, int64_t ldr = 0, const double* wfrag = nullptr);   // wfrag: Wt in MFMA fragment order [N_pad / 16][K_pad / 4][64] (dl_chi2_gemm_tile_bf, taken under DL_CHI2_BFRAG=1 only: the B operand in registers -- measured slower than both operands through LDS); resid [M, ldr]: the residual rows themselves, also written (may be null); panel_ranges [N_pad / 16][2]: 128-wide K panels [lo, hi) with non-zero Wt entries per column block (host array), or null
void dl_launch_finalize_part(const double* part, int n_tiles, const double* theta, int n_params, const double* priors, int64_t B, double* loglike, double* logprior,
                             int32_t* status, int post_mode, hipStream_t stream);
#ifndef DL_FG_NM
#define DL_FG_NM 19           // bias monomials of the velocileptors table combination
#define DL_FG_MONO_LD 20      // monomial rows are padded to 20 doubles in the point records
#endif
// emulated theories, feature path: residual rows out[B * R, ldo] (+)= monomial rows x (G . basis) from the point records written by the theory kernel (dl_feature_gemm.h);
// gfrag = the whitened folded operator of the observable in fragment order [N_pad / 16][nb_pad / 8][19][64][2]
void dl_launch_feature_gemm(const double* feat, int64_t feat_ld, int64_t feat_off, int nb_pad, int R, const double* gfrag, double* out, int64_t ldo, int N_pad, int64_t B,
                            int accumulate, hipStream_t stream);
// large plain-likelihood batches: LDS-DMA split-K GEMM (one split) with the partial-chi2 epilogue (dl_gemm_dma.h), finished by dl_launch_finalize_part
int dl_gemm_dma_chi2_parts(int N_pad);
void dl_launch_window_gemm_dma_chi2(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* part, int64_t M, int N_pad, int K_pad, hipStream_t stream, int n_live = 0);
// emulated theories, fused: emulator forward pass (MFMA) and feature GEMM of one observable in one launch (dl_emu_batch.h)
void dl_launch_emulated_feature(const DlObsDev& obs, const double* theta, int n_params, int64_t B, const double* gfrag, double* out, int64_t ldo, int N_pad, int accumulate,
                                hipStream_t stream);
// the whole step in ONE launch (dl_step_kernel: theory of four points per workgroup, hand-over of row blocks inside the launch, chi2 GEMM, fused finalize): BASELINE configs[1]-type
// contexts at <= 1024 points; dl_step_lds_bytes returns 0 when the configuration is not eligible.  `ready`: [>= B / 32] arrival counters that count up (target = 8 x launch number)
size_t dl_step_lds_bytes(const DlObsDev& obs, int64_t B, int N_pad);
void dl_launch_step(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power, const double* Wt, int64_t ldw, const double* bias,
                    double* part, int K_pad, int k_live, int32_t* counters, int32_t* ready, int32_t target, const double* priors, double* loglike, double* logprior, int32_t* status,
                    int post_mode, hipStream_t stream, const uint8_t* panel_ranges);
// stacked table engine (obs.eng[0].type == 2: the jaxeffort layout of emulators/conversion.py:44-98; dl_emu_stacked.h): every network of every group by MFMA and the feature
// GEMM in one launch; gfrag: [N_pad / 16][steps_per_block][64][2] (group by group, k / 8 by k / 8, monomial by monomial)
bool dl_emulated_stacked_ok(const DlObsDev& obs);
struct DlGramFinalize;
void dl_launch_emulated_stacked(const DlObsDev& obs, const double* theta, int n_params, int64_t B, const double* gfrag, double* out, int64_t ldo, int N_pad, int accumulate,
                                int steps_per_block, hipStream_t stream, DlGramFinalize* fin = nullptr, const double* bias = nullptr, const DlMargDev* mg = nullptr, int n_valid = 0,
                                double* basis_ws = nullptr, bool chains_done = false);   // basis_ws [B, n_networks x H]: workspace of the two-launch form (dl_emu_stacked_split.h); null: one launch
// the first launch of the two-launch form on its own (so that it carries the events of the theory phase): true if the shape takes that form and the chains were launched --
// dl_launch_emulated_stacked(..., basis_ws, true) then launches the feature GEMMs only
bool dl_launch_stk_chains(const DlObsDev& obs, const double* theta, int n_params, int64_t B, double* basis_ws, hipStream_t stream);
// (fin, bias, mg: the marginalised finalize in the kernel's tail -- one observable, N_pad = 128 --, see dl_kernels.hip; fin->done tells whether it was taken)
// ... with the Gram-matrix epilogue: gram [B, 16, 16] = Gram matrix of [residual + bias; derivative rows + tconst] per point instead of the rows themselves (one observable,
// N_pad = 128).  Returns false (nothing launched) when the rows of 16 points do not fit the LDS next to the forward pass.
// fin != nullptr and n_s <= 7: the marginalised finalize runs in the tail of the same kernel (outputs of DlGramFinalize; *fin->done = true), nothing is written to `gram`.
struct DlGramFinalize {
    const double* priors;
    double *loglike, *logprior;   // [B] or null
    int32_t* status;              // [B] or null
    double *solved, *hessian;     // [B, n_s], [B, n_s, n_s] or null
    int post_mode;
    bool done;
};
bool dl_launch_emulated_feature_gram(const DlObsDev& obs, const double* theta, int n_params, int64_t B, const double* gfrag, const double* bias, const DlMargDev& mg, int n_valid,
                                     double* gram, hipStream_t stream, DlGramFinalize* fin = nullptr);
// Fisher algebra (dl_fisher.hip): stencil rows of theta, then per centre the Gram matrix of [residual; derivative rows]
void dl_launch_fisher_stencil(const double* centers, const double* steps, int P, int64_t B, double* theta, hipStream_t stream);
int dl_fisher_waves(int n, int P, size_t* shm_bytes, int* chunk);   // centres per workgroup and columns staged at a time (0: no chunk fits the LDS)
void dl_launch_fisher(const double* rows, int64_t ld, int n, int n_slabs, int64_t slab_stride, const double* bias, const double* steps, int P, int64_t B, double* hessian,
                      double* gradient, double* offset, hipStream_t stream);
// Internal (dl_api.hip -> dl_ensemble.hip): theory + chi2 GEMM of B <= 2048 points of a plain likelihood, WITHOUT the finalize launch: *part = partial chi2
// [B, *n_tiles] (the context's workspace: valid until its next call), *priors = the prior table [P, 5].  Returns 0, 1 (error) or 2 (this context / batch does not
// take the chi2 GEMM path: use dl_eval_logposterior).
struct dl_ctx;
int dl_internal_eval_partials(dl_ctx* ctx, const double* theta_dev, int64_t B, const double** part, int* n_tiles, const double** priors, hipStream_t stream);
// Folded ensemble update (dl_ens_fold.h): theory launch that derives the B proposals of the half-step described by `fold` itself + chi2 GEMM that writes the accepts
// of the pending half-step; part_out [B, n_tiles] receives the proposals' partial chi2.  dl_internal_fold_info: 0 if the context takes this path (then *n_tiles and
// *priors are set), 2 if not.
struct DlEnsFold;
int dl_internal_fold_info(dl_ctx* ctx, int64_t B, int* n_tiles, const double** priors);
int dl_internal_eval_fold(dl_ctx* ctx, const DlEnsFold& fold, int64_t B, double* part_out, hipStream_t stream);
bool dl_launch_fullshape_ens(const DlObsDev* obs_host, int n_obs, const DlObsDev* obs_dev, const DlEnsFold& f, int64_t B, double* power, int64_t ld_power, int xcd_block, hipStream_t stream);
// analytic gradient (dl_fullshape_grad.h): per (point, observable) sums over the physical inputs -> gphys [B, n_obs, 8], then chain rule + prior gradient -> grad [B, P];
// Y [B, ldy] = -d~ W~ (the columns of every observable at its col_offset)
bool dl_grad_applicable(const DlObsDev* obs_host, int n_obs);
void dl_launch_fullshape_grad(const DlObsDev* obs_host, int n_obs, const DlObsDev* obs_dev, const double* theta, int n_params, int64_t B, const double* Y, int64_t ldy, double* gphys,
                              const double* priors, const int32_t* status, double* grad, hipStream_t stream);
// last-error string of the C ABI (thread-local, read by dl_last_error(NULL)); set by translation units other than dl_api.hip
void dl_set_last_error(const char* msg);
