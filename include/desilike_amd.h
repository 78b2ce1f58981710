/*
 * desilike_amd.h -- C ABI of the MI355X (gfx950) implementation of desilike's
 * theory -> observable -> Gaussian-likelihood hot path.
 *
 * The reference (cosmodesi/desilike) is pure Python: there is no FFI to mirror symbol by symbol.
 * What this library replaces is the chain of ``calculate()`` bodies that
 * ``BasePipeline.calculate`` (desilike/base.py:510-572) runs for one parameter point, evaluated
 * here for a whole batch of points per call (what ``vmap`` desilike/base.py:232-383 and
 * ``BasePosteriorSampler.logposterior`` desilike/samplers/base.py:144-200 loop over):
 *
 *   dl_eval_batch   <->  APEffect.calculate                      theories/galaxy_clustering/base.py:325-353
 *                        ShapeFitPowerSpectrumTemplate.calculate theories/galaxy_clustering/power_template.py:747-761
 *                        KaiserPowerSpectrumMultipoles.calculate theories/galaxy_clustering/full_shape.py:488-500
 *                        KaiserTracerPowerSpectrumMultipoles.calculate            full_shape.py:545-550 (+628-634 EFT-like)
 *                        WindowedPowerSpectrumMultipoles.calculate   observables/galaxy_clustering/window.py:459-473
 *                        TracerPowerSpectrumMultipolesObservable.calculate        power_spectrum.py:400-404
 *                        ObservablesGaussianLikelihood.calculate     likelihoods/base.py:13-17, 658-664
 *                        BaseLikelihood.get / ParameterCollection.prior           likelihoods/base.py:242-245, parameter.py:1889-1897
 *                        BaseLikelihood._solve (analytic marginalisation)         likelihoods/base.py:314-413
 *   dl_eval_theory  <->  the ``power`` / ``pktable`` state of the theory calculators (``__getstate__`` full_shape.py:502-510)
 *
 * Conventions
 *  - plain pointers and sizes only; float64 throughout (the reference forces x64: desilike/jax.py:18).
 *  - ``*_dev`` pointers are device (HBM) addresses on the context's GPU; the caller allocates and
 *    owns every in/out buffer, the library never frees caller memory.
 *  - return code 0 = success; non-zero = error, message via dl_last_error(ctx) (or dl_last_error(NULL)
 *    for errors before a context exists).  Per-point numerical failures are NOT errors: they are
 *    reported in ``status`` (the Python host maps them to -inf exactly like samplers/base.py:185-191).
 *  - a dl_ctx is used by one host thread at a time.  Its calls are asynchronous on the HIP stream passed in (NULL = default stream) and are
 *    serialised ACROSS streams by the library: a call issued on another stream than the previous call on the same context (the *_host entry points
 *    use a private stream) first waits, on the device, for that previous call -- the workspaces of a context are shared by all its calls.
 *    Different contexts (devices) are independent; no global mutable state except the last-error string.
 *  - the library FAILS (non-zero) when no GPU is present: there is no CPU fallback.
 */
#ifndef DESILIKE_AMD_H
#define DESILIKE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dl_config dl_config;   /* host-side key -> array store describing one likelihood */
typedef struct dl_ctx dl_ctx;         /* opaque; owns all persistent device constants and workspaces */

/* per-point status written by dl_eval_batch */
#define DL_STATUS_OK            0
#define DL_STATUS_OUT_OF_PRIOR  1   /* logprior = -inf (parameter.py:1994-2007) */
#define DL_STATUS_NONFINITE     2   /* loglikelihood is NaN/inf */
#define DL_STATUS_NAN_INPUT     3   /* a theta entry is NaN (samplers/base.py:57-61) */

/* enumerations used as values of integer config keys */
#define DL_TEMPLATE_FIXED     0   /* Fixed / Standard / BAO templates: fiducial table, only f = f_fid * df varies (power_template.py:198-202, 592-596, 372-376) */
#define DL_TEMPLATE_SHAPEFIT  1   /* power_template.py:747-761 */
#define DL_TEMPLATE_TURNOVER  2   /* power_template.py:1324-1333: P(k) = P_TO^(1 - m x^2) below the turn-over, P_TO^(1 - n x^2) above, x = log10 k / log10 k_TO - 1; keys
                                   * "kto_fid", "pkto_fid" (fiducial turn-over), inputs "in.m", "in.n", "in.qto", "in.dpto"; Kaiser-type theories */
#define DL_TEMPLATE_BANDS     3   /* power_template.py:893-961: P_tt = P_tt_fid (1 + sum_i (dptt_i - 1) tent_i(k)), P_dd = P_tt / (f_fid df)^2; keys "band_templates" [n_band, n_t] (tent
                                   * functions at the template knots), "in.band" [n_band, 2] (theta column or -1, value); "pk_dd_fid" = P_tt_fid / f_fid^2; Kaiser-type theories */
#define DL_THEORY_KAISER      0   /* full_shape.py:488-500, 545-550 */
#define DL_THEORY_EFT_KAISER  1   /* + counter / stochastic terms full_shape.py:628-634 */
#define DL_THEORY_BAO_DAMPED  2   /* damped BAO wiggles, 'standard' model bao.py:117-140 (+ broadband terms as pass-through columns, bao.py:495-534, 881-905) */
#define DL_THEORY_EMULATED    3   /* emulated P_ell tables (Taylor / MLP engines, emulators/base.py) + velocileptors bias-table combination full_shape.py:1150-1190 */
#define DL_APMODE_QPARQPER    0   /* theories/galaxy_clustering/base.py:341-350 */
#define DL_APMODE_QISO        1
#define DL_APMODE_QAP         2
#define DL_APMODE_QISOQAP     3
#define DL_PRIOR_UNIFORM      0
#define DL_PRIOR_NORM         1
#define DL_THEORY_TNS         4   /* TNS one-loop tables + tracer bias combination, full_shape.py:688-971 (csrc/dl_tns.h) */
#define DL_THEORY_PNG         5   /* Kaiser + scale-dependent bias from local primordial non-Gaussianity, primordial_non_gaussianity.py:75-112 */
#define DL_TRANSFORM_NONE     0
#define DL_TRANSFORM_CUBIC    1   /* power_spectrum.py:402-404 */

/* ---- configuration store ---------------------------------------------------------------------
 * Keys (all arrays are copied; "obs<i>." prefix = i-th observable of ObservablesGaussianLikelihood):
 *   n_params        i32[1]   number of sampled parameters P (= columns of theta)
 *   priors          f64[P*5] rows (kind, lo, hi, loc, scale)                   parameter.py:1994-2017
 *                            kind 0 uniform, 1 norm (fast paths 2003-2007); scipy.stats location-scale families evaluated as rv.logpdf(x) - rv.logpdf(loc)
 *                            (2012-2016): 2 expon, 3 laplace, 4 cauchy, 5 logistic, 6 halfnorm, 7 halfcauchy, 8 gumbel_r, 9 gumbel_l (csrc/dl_prior.h)
 *   n_obs           i32[1]
 *   precision       f64[n*n] or f64[n]  precision matrix (or its diagonal), n = total data size
 *                                        (Hartlap / Percival factors already applied by the host: likelihoods/base.py:623-656)
 *   precision_factor f64[n*n]  optional: any F with precision = F F^T (may be rank-deficient), used instead of the Cholesky factor of ``precision``
 *                                        (posterior marginalised once over linear parameters with constant derivative rows, likelihoods/base.py:257-312)
 *   obs<i>.theory, .template, .apmode, .transform   i32[1]
 *   obs<i>.damping_fid  i32[1]  1: Gaussian damping at the fiducial (k, mu) (SimpleTracerPowerSpectrumMultipoles full_shape.py:410-411) instead of the AP-distorted ones (492-493)
 *   obs<i>.eta, .f_fid, .a, .kp, .nd                f64[1]
 *   obs<i>.ells_in  i32[n_ell]      theory multipoles
 *   obs<i>.kin      f64[n_kin]      theory wavenumbers (same for every multipole)
 *   obs<i>.mu, obs<i>.wmu_ell       f64[n_mu], f64[n_ell*n_mu]   GL nodes and (2l+1) L_l(mu) w   (tgc/base.py:201-204)
 *   obs<i>.k_t, obs<i>.pk_dd_fid    f64[n_t]   template knots and fiducial linear power
 *   obs<i>.in.<name>  f64[2] = (theta column or -1, constant value) for
 *                     qpar qper qiso qap df dm dn sigmapar sigmaper b1X b1Y sn0
 *   obs<i>.ct_matrix f64[n_ell*n_kin*n_ct], obs<i>.in.ct f64[n_ct*2*2] (X and Y tracer inputs per term)
 *   obs<i>.sn_matrix f64[n_ell*n_kin*n_sn], obs<i>.in.sn f64[n_sn*2]
 *   obs<i>.wmatrix  f64[n_out*n_in] row-major (n_in = n_ell*n_kin); absent = identity      window.py:459-468
 *   obs<i>.kmask    i32[n_out]      row selection applied after the matrix / identity
 *   obs<i>.offset   f64[n_out_before_mask], obs<i>.shotnoise_in f64[n_ell], obs<i>.shotnoise_out f64[n_out]
 *   obs<i>.flatdata f64[n_out]
 *   obs<i>.in.pass  f64[n_pass*2]  parameters the observable is linear in through a constant matrix: appended to the theory vector as pass-through
 *                     columns n_in .. n_in+n_pass-1; obs<i>.wmatrix then has n_in+n_pass columns (BAO broadband terms; folded Hankel / emulator operators)
 *   BAO model (theory = 2): obs<i>.pknow_dd_fid f64[n_t] (no-wiggle table), obs<i>.in.dbeta, .in.sigmas, .in.sigmapar, .in.sigmaper f64[2],
 *                     obs<i>.bao_mode i32[1] (bits 0-3: 0 '' / recsym, 1 reciso; bits 4-7: wiggle model, 0 'standard', else 8 | 1 'fix-damping' | 2 'move-all' | 4 'fog-damping', bao.py:137-150), obs<i>.smoothing_radius f64[1]
 *                     flexible wiggles (bao.py:269-391): obs<i>.ml_matrix f64[n_ml*n_kin] kernels K_i(k), obs<i>.ml_ell i32[n_ml] multipole index of each term,
 *                     obs<i>.in.ml f64[n_ml*2], obs<i>.legendre f64[n_ell*n_mu] L_ell(mu); obs<i>.resummed f64[4] damping scales of the resummed wiggles (bao.py:186-199), obs<i>.in.dres
 *   TNS one-loop theory (theory = 4; full_shape.py:688-971): obs<i>.tns_k11 f64[n_k11] table wavenumbers (linspace(0.7 kin[0], 1.3 kin[-1], int(1.6 n_kin + 0.5)), full_shape.py:875),
 *                     obs<i>.tns_mu, obs<i>.tns_wmu f64[10] cosines and weights of the loop integrals (utils.weights_mu(10, 'leggauss'), full_shape.py:757), obs<i>.tns_fog i32[1]
 *                     (0 lorentzian, 1 gaussian, full_shape.py:870-873); obs<i>.in.sigmav, .in.b2, .in.bs, .in.b3 f64[2] (b1 through in.b1X); obs<i>.k_t = the template's
 *                     500 wavenumbers geomspace(1e-3, 2) (full_shape.py:855) are also the integration grid; EFT-like terms (ct_matrix, sn_matrix) as for theory 1
 *   PNG theory (theory = 5; primordial_non_gaussianity.py:75-112): obs<i>.png_alpha f64[n_t] alpha(k) = sqrt(P_phi / P_dd) of the fiducial template at the knots (the device rescales
 *                     it with the template factor per point), obs<i>.png_mode i32[1] (0 'bphi': bfnl = bphi fnl_loc; 1 'b-p': bfnl = 2 * 1.686 (b1 - p) fnl_loc), optional obs<i>.png_knorm f64[1]
 *                     (normalisation wavenumber of the transfer-function method); obs<i>.in.fnl_loc, .in.pX, .in.pY, .in.bphiX, .in.bphiY, .in.sigmas, .in.sigmasY f64[2] (b1 through in.b1X / in.b1Y);
 *                     tracer-velocity variant (primordial_non_gaussianity.py:196-330): obs<i>.png_velocity i32[1] = 1, obs<i>.png_velfac f64[1] = 100 / (1 + z), obs<i>.in.bv, .in.sigmau f64[2];
 *                     odd multipoles in ells_in, mu / wmu_ell on the nodes mu >= 0 with the weights of the mirror nodes added (<= 48 nodes)
 *   emulated theory (theory = 3): obs<i>.in.x f64[n_x*2] emulator inputs, obs<i>.in.vp f64[11*2] velocileptors 'pars' (b1 b2 bs b3 alpha0 alpha2 alpha4 alpha6
 *                     sn0 sn2 sn4), obs<i>.mono_mode i32[1] (0 none, 1 LPT physical basis, 2 REPT physical, 3 LPT direct, 4 REPT direct), obs<i>.vconst f64[3] (snd fsat sigv),
 *                     obs<i>.emu<e>.{type i32[1] (-1 constant, 0 MLP, 1 Taylor, 2 stacked MLPs: below), xlimits, widths, act, weights, ylimits, center, powers, coef, const}
 *                     for e = 0 (table basis), 1 (sigma8), 2 (fsigma8); obs<i>.marg.vp i32[11], obs<i>.marg.pass i32[n_pass]
 *   stacked table engine (obs<i>.emu0.type = 2): the layout the reference ships (emulators/conversion.py:44-98: engines '11' / 'loop' / 'ct' / 'st', each one network per
 *                     (z, ell) with kernels stacked [n_z, n_ell, in, out], outputs rescaled by the input logA; redshift selection / blend of full_shape.py:1416-1443):
 *                     obs<i>.emu0.widths i32[L + 1] (n_x, hidden widths <= 128: the same for every network; the final linear layers, y-scalers, the assembly of pktable, the
 *                     redshift blend, the k-interpolation and the window are folded into obs<i>.wmatrix), .act i32[1], .xlimits f64[n_x*2] (one min-max scaler),
 *                     .weights f64[n_networks * per_network] (per network, layer by layer: kernel [in, out] row-major then bias [out]),
 *                     .groups i32[n_groups*4]: (first network, one past the last, first bias monomial, one past the last) of each group of networks (a monomial belongs to
 *                     one group; a group may have no network: constant tables), .scale f64[n_groups*(n_x + 1)]: amplitude of a group's tables,
 *                     log amp = scale[g][n_x] + sum_j scale[g][j] x_j (conversion.py:88-92: v * exp(logA) * 1e-10, squared for 'loop');
 *                     obs<i>.wmatrix columns, group by group: [basis function h = (network, hidden unit) ..., then the constant 1][monomial of the group]
 *   marg.kind       i32[n_s]    analytically solved linear parameters (likelihoods/base.py:314-413): 1 = marginalised ('.marg'), 0 = best fit ('.best')
 *   marg.prior      f64[n_s*2]  (loc, 1 / scale^2) of their Gaussian priors (0 precision = flat prior)
 *   marg.x0         f64[n_s]    values at which the theory is evaluated (the parameters' default values, likelihoods/base.py:355)
 *   obs<i>.marg.sn0 i32[1], obs<i>.marg.sn i32[n_sn], obs<i>.marg.ct i32[n_ct*2]: index of the solved parameter each linear input feeds, or -1
 */
dl_config* dl_config_new(void);
int  dl_config_set_f64(dl_config* cfg, const char* key, const double* data, int64_t n);
int  dl_config_set_i32(dl_config* cfg, const char* key, const int32_t* data, int64_t n);
void dl_config_free(dl_config* cfg);

/* ---- context ---------------------------------------------------------------------------------*/
/* Uploads every constant once (window matrix folded with the Cholesky factor of the precision,
 * spline elimination coefficients, quadrature weights, ...).  device = HIP device ordinal. */
int  dl_create(dl_ctx** out, int device, const dl_config* cfg);
void dl_destroy(dl_ctx* ctx);
const char* dl_last_error(const dl_ctx* ctx);

/* integer properties: "n_params", "n_data", "n_obs", "n_in_total", "n_in_obs<i>", "n_out_obs<i>", "n_solved", "device" */
int64_t dl_info(const dl_ctx* ctx, const char* key);

/* ---- evaluation ------------------------------------------------------------------------------*/
/* theta_dev [B, P] row-major.  Outputs (any may be NULL): loglike_dev[B], logprior_dev[B],
 * flattheory_dev[B, n_data] (theory at the default values of the solved parameters), status_dev[B],
 * solved_dev[B, n_solved] (solution of the analytically solved parameters).  Asynchronous on ``hip_stream``. */
int  dl_eval_batch(dl_ctx* ctx, const double* theta_dev, int64_t B,
                   double* loglike_dev, double* logprior_dev, double* flattheory_dev,
                   int32_t* status_dev, double* solved_dev, void* hip_stream);

/* dl_eval_batch plus the derived outputs the reference attaches to the log-likelihood of a marginalised fit (likelihoods/base.py:388-390, consumed by
 * Chain.sample_solved, samples/chain.py:229-263): hessian_dev[B, n_solved, n_solved] = second derivatives of the log-likelihood w.r.t. the analytically
 * solved parameters, -T^T P T (the prior's part is the constant -diag(1 / scale^2) of marg.prior).  Any output may be NULL. */
int  dl_eval_batch_derived(dl_ctx* ctx, const double* theta_dev, int64_t B, double* loglike_dev, double* logprior_dev, int32_t* status_dev,
                           double* solved_dev, double* hessian_dev, void* hip_stream);

/* What a sampler consumes (BasePosteriorSampler.logposterior, desilike/samplers/base.py:144-200): logposterior_dev[B] = loglikelihood + logprior, and -inf
 * for points outside the prior, with NaN inputs or a non-finite likelihood (samplers/base.py:185-191); status_dev[B] optional.  One launch sequence, no
 * separate addition.  Asynchronous on ``hip_stream``. */
int  dl_eval_logposterior(dl_ctx* ctx, const double* theta_dev, int64_t B, double* logposterior_dev, int32_t* status_dev, void* hip_stream);

/* Fisher algebra of the Gaussian likelihood (desilike/fisher.py:731-750; LikelihoodFisher consumes the result, fisher.py:216-257).  For each of the B centres
 * (centers_dev [B, P]) the central-difference stencil centre -+ (lower_p, upper_p) e_p (steps_dev [B, P, 2], all positive; what the reference's Differentiation
 * derives from Parameter.delta, differentiation.py:306-352) is evaluated as one batch; per centre
 *     offset = -D.precision.D (no 1/2, fisher.py:746),  gradient [P] = -dD.precision.D,  hessian [P, P] = -dD.precision.dD^T,   dD_p = (D(c + u_p e_p) - D(c - l_p e_p)) / (l_p + u_p)
 * D = flattheory - flatdata.  Outputs (any may be NULL): hessian_dev [B, P, P], gradient_dev [B, P], offset_dev [B].  The context must have no analytically
 * solved parameter (vary them: the reference's Fisher does the same, fisher.py:688-695); P <= 31.  Asynchronous on ``hip_stream``. */
int  dl_eval_fisher(dl_ctx* ctx, const double* centers_dev, const double* steps_dev, int64_t B, double* hessian_dev, double* gradient_dev, double* offset_dev, void* hip_stream);
/* log-posterior [B] and its ANALYTIC gradient [B, P] (what the reference's gradient-based samplers take from jax.value_and_grad: desilike/samplers/hmc.py:194,
 * samplers/nuts.py:205): d logL / d theta = Y . d(theory vector) / d theta with Y = -W~^T d~, the derivative contracted on the fly by the theory's gradient kernel
 * (qpar / qper through every AP mode, df, dm, dn, b1 of either tracer, sn0; the spline of the template is linear in its data, so d / d dm is the spline of
 * d template / d dm), + the gradient of uniform / Gaussian priors.  Rows whose log-posterior is -inf get a zero gradient.  status_dev may be NULL.
 * Returns 0; 1 on error; 2 -- nothing launched -- when the context is outside the scope (Kaiser tracers without counter terms on uniform template knots, no damping,
 * no observable transform, no solved parameters, uniform / norm priors): differentiate numerically then (dl_eval_logposterior on a stencil). */
int  dl_eval_logposterior_grad(dl_ctx* ctx, const double* theta_dev, int64_t B, double* logposterior_dev, double* grad_dev, int32_t* status_dev, void* hip_stream);

/* Theory state of observable ``iobs`` for parity / plots / emulation:
 * power_dev [B, n_ell, n_kin] and (optional) tables_dev [B, 3, n_ell, n_kin] = pk_dd, pk_dt, pk_tt. */
int  dl_eval_theory(dl_ctx* ctx, const double* theta_dev, int64_t B, int32_t iobs,
                    double* power_dev, double* tables_dev, void* hip_stream);

/* TNS one-loop theory (obs<i>.theory = 4; reference: tns_kernels / tns_pt / TNSPowerSpectrumMultipoles / TNSTracerPowerSpectrumMultipoles, full_shape.py:688-971):
 * the 29 loop tables of the points on the table wavenumbers tns_k11, BEFORE the AP distortion / damping / projection -- tables_dev [B, 29, n_k11] in the order of
 * full_shape.py:882-883 (pk11, pk_dd, pk_b2d, pk_bs2d, pk_sig3sq, pk_b22, pk_b2s2, pk_bs22, pk_dt, pk_b2t, pk_bs2t, pk_tt, A0..A4, B0..B11).  For parity and plots:
 * dl_eval_batch / dl_eval_logposterior / dl_eval_theory run the same kernels and continue to the multipoles. */
int  dl_eval_tns_tables(dl_ctx* ctx, const double* theta_dev, int64_t B, int32_t iobs, double* tables_dev, void* hip_stream);

/* Host-pointer conveniences (copy in, evaluate on the default stream, copy out, synchronise):
 * used by the scalar ``likelihood(**params)`` call surface (desilike/base.py:1194-1196). */
int  dl_eval_batch_host(dl_ctx* ctx, const double* theta, int64_t B,
                        double* loglike, double* logprior, double* flattheory, int32_t* status, double* solved);
int  dl_eval_theory_host(dl_ctx* ctx, const double* theta, int64_t B, int32_t iobs,
                         double* power, double* tables);
/* dl_eval_logposterior with host pointers: what BasePosteriorSampler.logposterior (desilike/samplers/base.py:144-200) returns for ``values [B, ndim]``, one call
 * (pinned staging, one copy in, one copy out, one synchronisation of a private stream); status may be NULL. */
int  dl_eval_logposterior_host(dl_ctx* ctx, const double* theta, int64_t B, double* logposterior, int32_t* status);

/* ---- measurement -----------------------------------------------------------------------------*/
/* enable > 0: on one dl_eval_* call out of ``enable`` the kernels are launched with start / stop HIP events attached to their dispatch packets
 * (hipExtLaunchKernelGGL: the packets' own timestamps, as rocprofv3 --kernel-trace reads them; no event records between the kernels).
 * dl_profile_read synchronises and returns per-kernel milliseconds, the median over the (up to 256) sampled calls:
 * ms[0] theory kernel, ms[1] window / chi2 GEMM, ms[2] chi2 / prior finalize, ms[3] start of the first kernel to the end of the last,
 * ms[4] (if n >= 5) 0 (kept for compatibility), ms[5] (if n >= 6) number of sampled calls, ms[6..8] (if n >= 9) samples per kernel.
 * enable | (1 << 16) | (k << 17): single-kernel mode -- a kernel launched with events is followed by a ~3 us gap in the stream, so a sampled call attaches
 * them to ONE kernel only (k = 0 theory, 1 GEMM, 2 finalize); the other entries of ms[] are then 0.  (Without the gap that follows an instrumented
 * predecessor, the interval also contains the overlap with the predecessor's drain: ~0.2 us for the theory kernel.) */
int  dl_profile_enable(dl_ctx* ctx, int enable);
int  dl_profile_read(dl_ctx* ctx, double* ms, int32_t n);

/* ---- diagnostic switches ----------------------------------------------------------------------
 * Environment variables DL_* select alternative kernels kept for comparison, force variants or switch diagnostics on; NONE of them is needed in production.  They are read
 * ONCE per process (most at the first call that meets them, the ones of the table below at the first call of dl_options / dl_create) -- nothing on the per-call path calls
 * getenv.  dl_options_refresh() re-reads the table from the environment (the tests flip switches inside one process).
 *   kernel selection (results identical to the default path: tests/test_gpu_switches.py runs the parity checks under every one of them):
 *     DL_NO_MERGED_THEORY, DL_NO_EMU_BATCH, DL_NO_EMU_FUSED, DL_NO_GRAM_EPILOGUE, DL_NO_FUSED_SOLVE, DL_NO_GRAM_PLAIN, DL_NO_SCALED_ROW0, DL_EF_NO_EARLY_THETA, DL_FM_NO_LANE_SOLVE,
 *     DL_FM_NO_STAGE, DL_NO_FEATURE_PATH, DL_NO_TOEPLITZ, DL_NO_PANEL_SKIP, DL_NO_ROW_ALIGN, DL_NO_CHI2_BIG, DL_CHI2_FUSED, DL_STEP_KERNEL, DL_CHI2_GEMM_MAX, DL_CG_MT, DL_XCD_LOCAL,
 *     DL_FS_DENSE_MIN, DL_GEMM_DMA, DL_GEMM_WGS, DL_BAO_THREADS, DL_FFTLOG_GENERIC, DL_TNS_W, DL_TNS_WAVEK, DL_ENS_GLOBAL, DL_ENS_NO_DEFER, DL_ENS_NO_FOLD, DL_ENS_FORCE_COMM,
 *     DL_MH_NO_DEFER, DL_HOST_MODE (0 - 4: how the *_host entry points wait, see dl_eval_batch_host)
 *   diagnostics (in-kernel time stamps written to the named file, per-phase early exits; they synchronise: never set in production):
 *     DL_FS_STAMPS, DL_FS_STOP, DL_CG_STAMPS, DL_EF_STAMPS, DL_FM_STAMPS, DL_STK_STAMPS, DL_STEP_STAMPS, DL_ENS_STAMPS, DL_ENS_FOLD_STAMPS
 *   environment of the collectives: DL_RCCL_PATH, DL_COMM_TIMEOUT (desilike_amd/parallel.py, bench.py) */
void dl_options_refresh(void);

/* ---- FFTLog Hankel transform (row a11) --------------------------------------------------------
 * Batched device version of the reference's third-party ``cosmoprimo.PowerToCorrelation(k, ell, q=0, lowring=True)`` (call sites
 * theories/galaxy_clustering/base.py:76-77, 135; used by get_corr 127-136): for every (point, multipole)
 *     out = post_ell * reverse(irfft(rfft(zero-pad(fun * pre)) * u_ell))   restricted to the n un-padded points.
 * The grid constants are computed by the host (desilike_amd/fftlog.py: pre[n] = k^{3/2}; u[n_ell, npad/2 + 1, 2] = Mellin coefficients (re, im) with the
 * low-ringing offset; post[n_ell, n] = (-1)^{ell/2} (2 pi)^{-3/2} s^{-3/2}); npad = power of two in [16, 8192], zero padding (npad - n) / 2 in front.
 * dl_fftlog_apply: fun_dev, out_dev [B, n_ell, n] device arrays, asynchronous on ``hip_stream``.  Errors: non-zero, message via dl_last_error(NULL). */
typedef struct dl_fftlog dl_fftlog;
int  dl_fftlog_create(dl_fftlog** out, int device, int32_t n, int32_t npad, int32_t n_ell, const double* pre, const double* u, const double* post);
int  dl_fftlog_apply(dl_fftlog* plan, const double* fun_dev, int64_t B, double* out_dev, void* hip_stream);
void dl_fftlog_destroy(dl_fftlog* plan);

/* ---- exchange across the GPUs of a node (SURVEY 8b ``dl_allgather_logl``; 8e) -------------------------------------------------
 * One process per GPU.  The reference parallelises over parameter points only: ``vmap(..., backend='mpi')`` scatters the points and gathers the results
 * (desilike/base.py:310-335), samplers broadcast the log-posteriors to all ranks (desilike/samplers/base.py:196-200).  Here the one exchange is an
 * all-gather of per-point results over RCCL (xGMI), enqueued on the caller's HIP stream: no host synchronisation.  RCCL is bound at run time (dlopen):
 * ``rccl_library_path`` NULL = the copy already loaded in the process (e.g. PyTorch's), else the default search path.
 * Bootstrap: rank 0 calls dl_comm_unique_id and ships the 128 bytes to the other ranks by any host channel (the Python host uses a TCP store at
 * MASTER_ADDR:MASTER_PORT); every rank then calls dl_comm_create (collective).  Errors: non-zero, message via dl_last_error(NULL). */
typedef struct dl_comm dl_comm;
#define DL_COMM_ID_BYTES 128
int  dl_comm_unique_id(char* id /* [DL_COMM_ID_BYTES] */, const char* rccl_library_path);
int  dl_comm_create(dl_comm** out, int device, int rank, int world, const char* id /* [DL_COMM_ID_BYTES] */, const char* rccl_library_path);
void dl_comm_destroy(dl_comm* comm);
/* integer properties: "rank", "world", "device", "rccl_version" (comm may be NULL for the last) */
int64_t dl_comm_info(const dl_comm* comm, const char* key);
/* recv_dev[world * count] <- concatenation over ranks of send_dev[count]; in place when send_dev == recv_dev + rank * count.  Asynchronous on ``hip_stream``. */
int  dl_comm_allgather_f64(dl_comm* comm, const double* send_dev, double* recv_dev, int64_t count, void* hip_stream);
/* buf_dev[count] of rank ``root`` -> every rank (walker positions must be identical on all ranks: desilike/samplers/base.py:45-54) */
int  dl_comm_broadcast_f64(dl_comm* comm, double* buf_dev, int64_t count, int root, void* hip_stream);

/* ---- device-resident ensemble sampler (BASELINE configs[4]) ------------------------------------------------------------------
 * The affine-invariant stretch move that ``EmceeSampler`` drives on the host in the reference (desilike/samplers/emcee.py:69-111: emcee.EnsembleSampler(...,
 * vectorize=True), default StretchMove; log-posterior conventions of desilike/samplers/base.py:144-200), with walker positions, log-posteriors and the
 * counter-based random number generator (Philox4x32-10 keyed by ``seed``) resident on the GPU: per half-step ONE small kernel (accept the previous half,
 * draw the split, stretch proposals), dl_eval_logposterior on this rank's share of the proposals, and -- with a communicator -- one in-place
 * dl_comm_allgather_f64; nothing synchronises with the host.  Every rank holds the full ensemble (identical bits: the draws are pure functions of
 * (seed, iteration, half-step, slot)); rank r evaluates proposals [r * c, (r + 1) * c) of a half-step, c = ceil(nwalkers / 2 / world).
 * ``offset``: constant added to every log-posterior (dl_eval_logposterior of a posterior context marginalised once over linear parameters). */
typedef struct dl_ensemble dl_ensemble;
int  dl_ensemble_create(dl_ensemble** out, dl_ctx* ctx, int32_t nwalkers, double a, uint64_t seed, double offset, dl_comm* comm /* may be NULL */);
void dl_ensemble_destroy(dl_ensemble* ens);
/* host arrays coords[nwalkers, P], logposterior[nwalkers] (NULL: evaluated at the next dl_ensemble_run); synchronises ``hip_stream`` */
int  dl_ensemble_set_state(dl_ensemble* ens, const double* coords, const double* logposterior, void* hip_stream);
/* ``niterations`` ensemble updates, enqueued on ``hip_stream`` (asynchronous).  chain_dev[niterations / thin_by, nwalkers, P] and
 * chain_logp_dev[niterations / thin_by, nwalkers] (device, caller-owned, either may be NULL) receive the ensemble after every thin_by-th update. */
int  dl_ensemble_run(dl_ensemble* ens, int64_t niterations, int32_t thin_by, double* chain_dev, double* chain_logp_dev, void* hip_stream);
/* host arrays (any may be NULL): current positions, log-posteriors, number of accepted proposals per walker; synchronises ``hip_stream`` */
int  dl_ensemble_get_state(dl_ensemble* ens, double* coords, double* logposterior, int64_t* naccepted, void* hip_stream);
/* resume: the iteration counter of the random number generator (the draw of update i is a pure function of (seed, i): a sampler that restores positions, seed and
 * counter continues the very chain it saved) and, if not NULL, the accepted-proposal counts per walker [nwalkers] (host) */
int  dl_ensemble_set_counter(dl_ensemble* ens, int64_t iteration, const int64_t* naccepted, void* hip_stream);
/* integer properties: "nwalkers", "n_params", "iteration", "rank", "world", "rows_per_rank" */
int64_t dl_ensemble_info(const dl_ensemble* ens, const char* key);

/* ---- device-resident blocked Metropolis-Hastings sampler ----------------------------------------------------------------------------------
 * The reference's own MCMC (desilike/samplers/mcmc.py: MHSampler 25-127, BlockProposer 199-328 -- the CosmoMC / cobaya fast-slow blocked proposal: cycle through
 * the columns of a random rotation of every block, radial scale from a mixture, jump = Cholesky factor of the proposal covariance x direction), with ``nchains``
 * chains x ``vectorize`` speculative proposals per try (mcmc.py:86-105: the first proposal that passes the Metropolis test is taken, the rejected ones before it
 * add to the weight of the current state) evaluated as ONE batch of nchains x vectorize rows of dl_eval_logposterior.  Positions, log-posteriors, weights and the
 * random draws (csrc/dl_mh.h: Philox4x32-10, pure functions of (seed, chain id, proposer call)) are resident on the GPU; a try is one small kernel + the evaluation,
 * nothing synchronises with the host.  Not built: dragging (mcmc.py:52-84 -- it exists to spare evaluations of slow parameters; here every parameter costs the same launch).
 *   chain_ids[nchains]   global index of every chain (NULL: 0 .. nchains - 1): chains are reproducible whatever the rank that runs them
 *   order[P]             sorted (block) position i -> parameter index of the context (NULL: identity); blocks[nblocks] sizes in sorted order, slowest first;
 *                        oversample[nblocks] (NULL: 1) as BlockProposer's oversample_factors
 *   offset               constant added to every log-posterior (posterior contexts marginalised once over linear parameters) */
typedef struct dl_mh dl_mh;
int  dl_mh_create(dl_mh** out, dl_ctx* ctx, int32_t nchains, int32_t vectorize, const int32_t* chain_ids, const int32_t* order, const int32_t* blocks,
                  const int32_t* oversample, int32_t nblocks, double proposal_scale, uint64_t seed, double offset, int64_t max_tries);
void dl_mh_destroy(dl_mh* mh);
/* lower-triangular Cholesky factor [P, P] (host, row-major) of the proposal covariance of the parameters in SORTED order (BlockProposer.set_covariance, mcmc.py:298-328) */
int  dl_mh_set_covariance(dl_mh* mh, const double* cholesky, void* hip_stream);
/* host arrays: coords[nchains, P] (context order), logposterior[nchains] (NULL: evaluated at the next dl_mh_run; must be finite), weight[nchains] of the current
 * states (NULL: 1), naccepted[nchains] (NULL: 0), and the try counter of the random draws (resume: a chain continues from (position, counters) alone) */
int  dl_mh_set_state(dl_mh* mh, const double* coords, const double* logposterior, const int64_t* weight, const int64_t* naccepted, int64_t tries, void* hip_stream);
/* ``ntries`` tries of every chain, enqueued on ``hip_stream`` (asynchronous).  A state is recorded when the chain leaves it (with its final weight; the starting
 * state is skipped and every thin_by-th accepted state kept, mcmc.py:97-99): out_coords_dev[nchains, ntries, P], out_logp_dev[nchains, ntries],
 * out_weight_dev[nchains, ntries] hold the out_count_dev[nchains] records of this call (device, caller-owned). */
int  dl_mh_run(dl_mh* mh, int64_t ntries, int32_t thin_by, double* out_coords_dev, double* out_logp_dev, int64_t* out_weight_dev, int32_t* out_count_dev, void* hip_stream);
/* the same with HOST record arrays of the same shapes (callers without device memory of their own: FFI bindings); the sampler keeps the device buffers; synchronises */
int  dl_mh_run_host(dl_mh* mh, int64_t ntries, int32_t thin_by, double* out_coords, double* out_logp, int64_t* out_weight, int32_t* out_count, void* hip_stream);
/* host arrays (any may be NULL): current positions, log-posteriors, weights, accepted moves, consecutive tries without an accepted proposal; synchronises */
int  dl_mh_get_state(dl_mh* mh, double* coords, double* logposterior, int64_t* weight, int64_t* naccepted, int32_t* fails, void* hip_stream);
/* integer properties: "nchains", "vectorize", "n_params", "tries", "cycle" (entries of the parameter cycler), "max_tries" */
int64_t dl_mh_info(const dl_mh* mh, const char* key);

/* ---- MLP emulator training ---------------------------------------------------------------------------------------
 * The reference trains its MLP emulators through the third-party engine ``cosmoprimo.emulators.tools.MLPEmulatorEngine`` (desilike/emulators/__init__.py:510-533;
 * network structure: emulators/conversion.py:20-96).  Here: fp64 mini-batch Adam on the mean squared error of the (already scaled) outputs, entirely on the device
 * (fp64 MFMA GEMMs for forward / backward, fixed summation orders: deterministic).  Parameters are one flat vector: per layer the kernel [n_in, n_out] row-major,
 * then the bias [n_out].  activation: 0 silu, 1 relu, 2 tanh between the layers; the last layer is linear.  x_dev [n_samples, n_in], y_dev [n_samples, n_out] device
 * arrays.  Batches are consecutive chunks of ``batch`` samples (the caller shuffles the set); dl_mlp_train enqueues ``n_steps`` Adam steps on ``hip_stream`` and, if
 * loss_host != NULL, copies the per-step batch losses [n_steps] back (then synchronises).  Errors: non-zero, message via dl_last_error(NULL). */
typedef struct dl_mlp dl_mlp;
int  dl_mlp_create(dl_mlp** out, int device, int32_t n_layers, const int32_t* widths /* [n_layers + 1] */, int32_t activation, const double* weights /* host, flat */);
void dl_mlp_destroy(dl_mlp* net);
int64_t dl_mlp_info(const dl_mlp* net, const char* key);   /* "n_weights", "n_layers", "n_in", "n_out", "step" */
int  dl_mlp_train(dl_mlp* net, const double* x_dev, const double* y_dev, int64_t n_samples, int64_t batch, int64_t n_steps, double lr, double beta1, double beta2, double eps,
                  double* loss_host, void* hip_stream);
int  dl_mlp_loss_and_grad(dl_mlp* net, const double* x_dev, const double* y_dev, int64_t rows, double* loss_host /* [1] */, double* grad_host /* [n_weights] */, void* hip_stream);
int  dl_mlp_forward(dl_mlp* net, const double* x_dev, int64_t rows, double* y_dev /* [rows, n_out] */, void* hip_stream);
int  dl_mlp_get_weights(dl_mlp* net, double* weights /* host, flat */, void* hip_stream);

/* ---- Gaussian covariance of P_ell / xi_ell observables ------------------------------------------------------------------
 * Reference: desilike/observables/galaxy_clustering/covariance.py:355-456 (ObservablesCovarianceMatrix._run).  Every element is
 *     C[row, col] = front / den * sum_q (sigma(k_q) w_q) w2_q + const,   sigma(k) = prefactor * sum_{la, lb} (P1_la(k) P2_lb(k) - zero lag) G[la][lb],
 * P the theory multipoles (+ shot noise on the monopole) interpolated linearly between the theory's wavenumbers like np.interp.  The plan (host arrays, copied) lists
 *   cell_i [n_cells, 8]  row, col, theory of the first / second observable, index into gtab, zero-lag flag (xi x xi: the product of the shot noises is removed from the
 *                        monopole x monopole term), first point, number of points;
 *   cell_d [n_cells, 4]  prefactor, front, den, const;
 *   pt_i   [n_points, 2] interval j of k_q in the wavenumbers of theory 1 / theory 2 (0 <= j <= n_k - 2);
 *   pt_d   [n_points, 6] (k_q - k_j, k_{j+1} - k_j) for theory 1, the same for theory 2, w, w2;
 *   gtab   [n_gtab, 5, 5] integral of L_la L_lb L_l1 L_l2 over mu, indexed by the positions of la / lb in the theories' multipoles;
 *   sym    [n_sym, 2]    pairs (row, col) of the diagonal blocks replaced by their mean with the transposed element (covariance.py:349-351);
 * theory t: n_ell[t] <= 5 multipoles on n_k[t] wavenumbers, ell0[t] = position of the monopole (-1: none), shotnoise[t] added to it.
 * dl_cov_apply: power_dev[t] = device array [B, n_ell[t], n_k[t]] (e.g. from dl_eval_theory), cov_dev [B, n, n] device output (zero where no cell writes); asynchronous
 * on ``hip_stream``; B <= 65535. */
typedef struct dl_cov dl_cov;
int  dl_cov_create(dl_cov** out, int device, int32_t n, int32_t n_theories, const int32_t* n_ell, const int32_t* n_k, const int32_t* ell0, const double* shotnoise,
                   int64_t n_cells, const int32_t* cell_i, const double* cell_d, int64_t n_points, const int32_t* pt_i, const double* pt_d, int32_t n_gtab, const double* gtab,
                   int64_t n_sym, const int32_t* sym);
int  dl_cov_apply(dl_cov* plan, const double* const* power_dev, int64_t B, double* cov_dev, void* hip_stream);
void dl_cov_destroy(dl_cov* plan);

#ifdef __cplusplus
}
#endif
#endif /* DESILIKE_AMD_H */
