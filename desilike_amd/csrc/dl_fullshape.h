// dl_fullshape.h -- per-point arithmetic of the full-shape theory kernel.
//
// One workgroup evaluates one (parameter point, observable): ShapeFit template rescaling at the N_t knots,
// not-a-knot cubic-spline moments by a segmented Thomas sweep, conversion to per-interval polynomials,
// AP remap + spline evaluation at (k, mu'), Gauss-Legendre projection onto multipoles, tracer bias combination.
// The body is written as barrier-separated *phases*, each a function of (tid, nthreads) acting on
// workgroup-shared arrays, so that the HIP kernel (dl_kernels.hip) and the CPU emulation used by the
// `not gpu` tests (tests/csrc/emulate.cpp) run literally the same code.
//
// Design notes (measured on MI355X with tools/mfma_f64_probe.hip: a dependent fp64 FMA costs 40 cycles, one wave
// issues an fp64 FMA every 8.5 cycles, a SIMD every 4.4):
//  * the tridiagonal sweeps are split into 64 segments; the state entering a segment is a 30-40 term DOT PRODUCT
//    with precomputed weights (independent FMAs) instead of a warm-up recurrence (dependent FMAs): the sweep
//    multipliers decay like (2 - sqrt 3)^d, so truncating at < 1e-19 is exact in fp64;
//  * the (k, mu) loop evaluates S(u) = ((d3 u + d2) u + d1) u + d0 with u the fractional knot index, the four
//    coefficients of an interval being one 32-byte LDS read; the multipole weights, the Jacobian and (when the
//    separate tables are not requested) the bias factors (b1X + f mu'^2)(b1Y + f mu'^2) are folded per mu node.
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
// pins the order of an unrolled loop: the four accumulators must be complete, and no memory access may move across (without it the scheduler
// turns the convolution below into four serial chains over the whole window, with the window held in 120 VGPRs)
#define DL_PIN4(a, b, c, d) __asm__ volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")
// streaming store: the row goes out to memory while the kernel is still running instead of sitting dirty in L2 until the end-of-kernel write-back
#define DL_STREAM_STORE(ptr, value) __hip_atomic_store((ptr), (value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define DL_HD inline
#define DL_PIN4(a, b, c, d)
#define DL_STREAM_STORE(ptr, value) (*(ptr) = (value))
#endif

#define DL_MAX_ELL 5
#define DL_MAX_MU 32
#define DL_MAX_EFT 8
#define DL_FS_THREADS 256
#define DL_FS_KT 192          // threads of the fast kernel that build the spline (waves 0-2); wave 3 runs the per-mu chain beside them
#define DL_MAX_SEG 64
#define DL_SEG_PARTS 4      // threads cooperating on the warm-up dot product of one segment (DL_MAX_SEG * DL_SEG_PARTS = DL_FS_THREADS)
#define DL_MAX_X 16        // emulator inputs
#define DL_MAX_LAYERS 8    // dense layers of an emulator MLP
#define DL_MAX_WIDTH 256   // hidden units per layer (one thread each)
#define DL_N_VPARS 11      // velocileptors 'pars': b1 b2 bs b3 alpha0 alpha2 alpha4 alpha6 sn0 sn2 sn4 (full_shape.py:1184)
#define DL_N_MONO 19       // bias monomials (full_shape.py:1185)
#define DL_STK_REC 8       // doubles per group record of the stacked table engine (DlObsDev::Stack)
#define DL_STK_MAX_GROUPS 8   // device groups (a group of more than DL_STK_MAX_MONO monomials is split; the networks are shared)
#define DL_STK_MAX_MONO 5   // monomials per device group (accumulator tiles of the feature GEMM: with the three operand buffers, 20 registers per monomial)
#define DL_MAX_ML 40       // multiplicative wiggle terms of the flexible BAO model (bao.py:310-322: up to 12 nodes per multipole)
#define DL_PNG_MAX_MU 48   // mu nodes of the PNG kernel (its tracer-velocity variant integrates 81 trapezoid nodes on [-1, 1]: 41 after folding)
#define DL_MAX_BAND 16     // bands of the velocity-divergence template
#define DL_MAX_PASS 32     // pass-through columns: linear (broadband) parameters appended to the theory vector
#define DL_MAX_SOLVED 16   // analytically solved (marginalised / best-fit) linear parameters
#define DL_FIR_D 28        // half-width of the convolution that inverts the uniform-knot spline system: |mu|^28 = 1e-16, mu = sqrt(3) - 2
#define DL_FIR_PAD 32      // zeros either side of the knot values in LDS (>= DL_FIR_D + 4)
// taps t_e, e = 0 .. DL_FIR_D, of  w_j = sum_e t_|e| y_{j+e}:  w solves w_{j-1} + 4 w_j + w_{j+1} = y_{j-1} - 2 y_j + y_{j+1} on the infinite grid:
// t_0 = 2 c (mu - 1), t_e = c (1 - mu)^2 mu^(e-1), c = 1 / (2 sqrt 3)
#define DL_FIR_TAPS { -0.7320508075688774, 0.4641016151377547, -0.12435565298214114, 0.033320996790809666, -0.008928334181097484, 0.002392339933580261, -0.0006410255532235571, 0.0001717622793139658, -4.602356403230609e-05, 1.2331976815258487e-05, -3.3043432287278414e-06, 8.85396099652874e-07, -2.3724116988365352e-07, 6.356857988173978e-08, -1.7033149643305495e-08, 4.564018691482174e-09, -1.2229251226231984e-09, 3.2768179901061786e-10, -8.780207341927256e-11, 2.3526494666472232e-11, -6.303905246616353e-12, 1.6891263199931699e-12, -4.5260003335632413e-13, 1.212738134321263e-13, -3.24952203721809e-14, 8.707068056597241e-15, -2.333051854208057e-15, 6.251393602349824e-16, -1.675055867318723e-16 }
#define DL_FIR_MU (-0.2679491924311228)       // sqrt(3) - 2
#define DL_FIR_LN_ABS_MU (-1.3169578969248164)
#define DL_SEG_QMAX 12     // dot-product terms per thread: warm-up length <= DL_SEG_PARTS * DL_SEG_QMAX = 48

struct DlInput {
    int32_t col;    // >= 0: column of theta; < 0: use `value`
    int32_t pad;
    double value;
};

// On the device the load is unconditional (column 0 for a fixed input, then a select): behind a branch every parameter of a point was a load-and-wait of its own
// (seven to ten serialized round trips at the head of the per-point chain); unconditional loads are issued back to back and waited for once.
#ifdef __HIP_DEVICE_COMPILE__
DL_HD double dl_get(const DlInput& in, const double* th) { const double v = th[in.col >= 0 ? in.col : 0]; return in.col >= 0 ? v : in.value; }
#else
DL_HD double dl_get(const DlInput& in, const double* th) { return in.col >= 0 ? th[in.col] : in.value; }
#endif

struct DlObsDev {
    int32_t theory, templ, apmode, transform;
    int32_t n_ell, n_kin, n_mu, n_t;
    int32_t n_in, ell0, n_ct, n_sn;
    int32_t seg_len, seg_warm, n_seg, fixed_spline;
    int32_t uniform_knots, toeplitz; // knots uniform in log10 k (to < 1e-6 of the spacing): interval index = floor of the scaled abscissa;
                                     // toeplitz: uniform to rounding (< 1e-11 of the spacing): spline moments by the convolution dl_fs_phase2_fir
    int64_t col_offset;  // first column of this observable in a row of the (concatenated) power buffer
    // analytic marginalisation: each point owns 1 + n_var consecutive rows of the power buffer; row 1 + v holds
    // d(power) / d(solved parameter of variable slot v) (counter terms: the derivative depends on the point through P_dd,l=0)
    int32_t n_var, damping_fid;            // damping_fid: damping evaluated at the FIDUCIAL (k, mu) (SimpleTracerPowerSpectrumMultipoles, full_shape.py:410-414)
    int32_t marg_ct_slot[DL_MAX_EFT][2];   // variable slot fed by counter term c through tracer X / Y, or -1
    double eta, f_fid, a, nd, x0, inv_hx;
    double end0a, end0b, end1a, end1b;  // not-a-knot end relations: M[0] = end0a M[1] + end0b M[2]; M[n-1] = end1a M[n-2] + end1b M[n-3]
    DlInput qpar, qper, qiso, qap, df, dm, dn, sigpar, sigper, b1X, b1Y, sn0;
    DlInput to_m, to_n, qto, dpto;         // turn-over template (template kind 2): slopes below / above the turn-over, its shift and amplitude (power_template.py:1324-1333)
    double lkto_fid, lpkto_fid;            // log10 of the fiducial turn-over wavenumber, ln of the fiducial power there
    // band template (template kind 3, power_template.py:893-961): P_tt = P_tt_fid (1 + sum_i (dptt_i - 1) tent_i(k)), P_dd = P_tt / f^2
    int32_t n_band, pad_band;
    DlInput band_in[DL_MAX_BAND];
    const double* band_tab;                // [n_band][n_t] tent functions at the knots
    // tracer-velocity variant of the PNG theory (primordial_non_gaussianity.py:196-330): P = jac fog (b + f mu'^2) (bv f mu' velfac / k') P(k'), fog = sinc(sigmau k') / (1 + sigmas^2 k'^2 mu'^2 / 2)
    int32_t png_vel, pad_vel;
    DlInput bv, sigmau;
    double png_velfac;                     // 100 / (1 + z)
    DlInput ct_in[DL_MAX_EFT][2];
    DlInput sn_in[DL_MAX_EFT];
    // pass-through columns n_in .. n_in + n_pass - 1 of the theory vector: parameters the observable is linear in through a constant
    // matrix folded into the window (BAO broadband terms bao.py:495-534, 881-905)
    int32_t n_pass, bao_mode;              // bao_mode bits 0-3: 0 = '' / 'recsym', 1 = 'reciso' (bao.py:131); bits 4-8: wiggle model, 0 = 'standard' (bao.py:123-136), else
                                           // 8 | (1: 'fix-damping') | (2: 'move-all') | (4: 'fog-damping') (bao.py:137-150), or 16 | (2: 'move-all') | (4: 'fog-damping'):
                                           // resummed wiggles (bao.py:165-266), or 32 | (2: 'move-all'): flexible wiggles (bao.py:269-391)
    const double* pass_tab;                // [n_pass][2]: theta column (or -1), constant value -- in the arena (see ml_tab)
    DlInput dbeta, sigmas;                 // BAO wiggle model (bao.py:117)
    DlInput dres;                          // resummed wiggles: growth rescaling d (bao.py:201)
    double res_sig[4];                     // resummed wiggles: sigma_dd^2, sigma_nl^2, sigma_x^2, shotnoise * sigma_sn^2 (bao.py:186-199)
    // flexible wiggles (bao.py:269-391): terms ml_i K_i(k) L_{ell_i}(mu) multiplying the wiggles; K [n_ml, n_kin] sits behind ct_matrix, the Legendre
    // polynomials L_ell(mu) [n_ell, n_mu] behind sn_matrix (both unused by the BAO kernels otherwise)
    int32_t n_ml, pad_ml;
    const double* ml_tab;                  // [n_ml][3]: theta column (or -1), constant value, index (in ells_in) of the term's multipole -- in the arena, not in
                                           // the kernarg segment (the struct travels by value with every launch of every theory kernel)
    double smoothing_radius;
    // emulated theory (kind 3): features phi[(h, m)] = basis_h(theta) * mono_m(theta); the last emulator layer, the bias-table sum
    // (full_shape.py:1182-1186), the k-interpolation and the window are ONE matrix folded on the host (desilike_amd/emulators.py)
    int32_t n_x, n_basis, n_mono, mono_mode;   // mono_mode: 0 none, 1 LPT physical basis, 2 REPT physical, 3 LPT direct, 4 REPT direct
    int32_t nb_pad, feat_pad;                  // feature path (dl_feature_gemm.h): n_basis rounded up to 8; (unused)
    int64_t feat_off;                          // column of this observable's record [basis nb_pad | mono (1 + n_var) x 20] in a row of the feature buffer
    DlInput x_in[DL_MAX_X];
    DlInput vp_in[DL_N_VPARS];                 // b1(p) b2(p) bs(p) b3(p) alpha0(p) alpha2(p) alpha4(p) alpha6 sn0(p) sn2(p) sn4(p)
    int32_t vp_slot[DL_N_VPARS];               // variable (derivative-row) slot of an analytically solved alpha* / sn*, or -1
    int32_t pad2;
    double snd, fsat, sigv;                    // full_shape.py:1154-1157
    struct Engine {                            // [0] table basis, [1] sigma8, [2] fsigma8
        int32_t type, n_layers, act, n_terms;  // type: -1 absent (constant `cst`), 0 MLP, 1 Taylor; act: 0 silu, 1 relu, 2 tanh
        int32_t widths[DL_MAX_LAYERS + 1];     // MLP: n_x, hidden..., output
        int32_t pad;
        double ylo, yscale, cst;               // scalar engines: y = v * yscale + ylo (inverse min-max scaler, conversion.py:79)
        const double *xlo, *xinv;              // MLP x-scaler (v - lo) / (hi - lo) (conversion.py:75-77)
        const double *weights;                 // MLP: per layer kernel [in, out] then bias [out], packed
        const double *center, *powers, *coef;  // Taylor: center [n_x], powers [n_terms, n_x], coef [n_terms] (scalar engines)
    } eng[3];
    // stacked table engine (eng[0].type == 2; the layout the reference ships, emulators/conversion.py:44-98): SEVERAL networks with the same hidden layers
    // (eng[0].widths = n_x, hidden...; eng[0].weights + t * trunk_doubles: network t), in groups: group g = networks [tb, te) whose folded final layers feed the
    // bias monomials [m0, m1) (the engines '11' / 'loop' / 'ct' / 'st' x (z, ell) stacks), times an amplitude that is log-linear in the inputs (conversion.py:88-92)
    struct Stack {
        int32_t n_groups, n_trunks, trunk_doubles, max_k;   // max_k: largest number of basis functions of a group, (te - tb) * H + 1
        int32_t frag_doubles, max_net;                      // doubles of one network in `wfrag`; largest number of networks of a group (te - tb)
        const double* wfrag;   // the networks' weights in MFMA fragment order (dl_emu_stacked.h): per network, per layer [output tile][k-step][lane = col + 16 g] = K[4 step + g][16 tile + col]
                               //   (zero beyond the layer), then the biases [output tile][16]
        const double* table;   // [n_groups][DL_STK_REC] as doubles: tb, te, m0, m1, col (first column of the group's block in the theory vector), nm (monomials per basis
                               //   function in that block: the column stride), mo (m0 - first monomial of the block), kq (first operand step of the group in a column block)
        const double* scale;   // [n_groups][n_x + 1]: log amplitude = scale[g][n_x] + sum_j scale[g][j] x_j
    } stk;
    const double *coef_w, *coef_n;         // [n_t, 4] interval polynomials of the wiggle P_dd - P_now and of P_now (fixed BAO template)
    const double *pknow_k;                 // [n_kin] P_now at the fiducial k (bao.py:137)
    const double *kin, *lkin, *mu, *wmu;          // [n_kin], log10(kin) [n_kin], [n_mu], [n_ell * n_mu]
    const double *x_t, *pk_fid, *sf_th, *sf_lg;   // log10(k_t), fiducial P, tanh(a ln(k/kp)), ln(k/kp): all [n_t]
    const double *ih, *dlt;                        // 1 / (x_t[j+1] - x_t[j]); x_t[j] - (x0 + j / inv_hx): [n_t]
    const double *sp_A, *sp_nC, *sp_inv;           // Thomas sweep multipliers / pivots of the reduced system [n_t - 2]
    const double *sp_gf, *sp_gb;                   // [DL_SEG_QMAX, DL_FS_THREADS] warm-up weights of the forward / backward sweeps: term d = 4 q + part of
                                                   // segment seg sits at [q][4 seg + part] (zero beyond the warm-up length / the ends of the system)
    const double *coef_fixed;                      // [n_t, 4] interval polynomials of the fiducial table (fixed templates)
    const double *ct_matrix, *sn_matrix;           // [n_ell, n_kin, n_ct], [n_ell, n_kin, n_sn]
    // TNS one-loop theory (kind 4, dl_tns.h): FoG dispersion and the non-linear bias parameters (full_shape.py:865, 957-971); tns_plan: HOST handle of the
    // geometry tables and the per-evaluation workspace (dl_tns.hip), never dereferenced on the device
    DlInput sigmav, b2, bs, b3;
    void* tns_plan;
    // scale-dependent bias from local primordial non-Gaussianity (kind 5; primordial_non_gaussianity.py:75-112): bX = b1X + bfnlX alpha(k'), with bfnl = bphi fnl_loc
    // (png_mode 0, 'bphi') or 2 * 1.686 (b1 - p) fnl_loc (1, 'b-p'); Lorentzian damping per tracer (sigmas: X, sigmasY); alpha at the knots = png_alpha (fiducial
    // template) * sqrt(norm / template factor), norm = template factor at the normalisation wavenumber (png_th0, png_lg0; 0, 0: none -- method 'prim')
    DlInput fnl, pX, pY, bphiX, bphiY, sigmasY;
    int32_t png_mode, pad_png;
    double png_th0, png_lg0;
    const double* png_alpha;               // [n_t]
};

DL_HD int dl_fs_n_dd0(const DlObsDev& o) { return (o.n_var > 0 && o.n_ct > 0) ? o.n_kin : 0; }   // P_dd,l=0 kept for the derivative rows

// Layout of the small per-point scratch `pt` (doubles) in workgroup-shared memory
enum {
    DL_PT_QPAR = 0, DL_PT_QPER, DL_PT_JAC, DL_PT_F, DL_PT_B1X, DL_PT_B1Y, DL_PT_SN0ND, DL_PT_DAMP,
    DL_PT_LQ = 8,                                   // log10(F_m / qper)
    DL_PT_FAC = DL_PT_LQ + DL_MAX_MU,               // F_m
    DL_PT_SD = DL_PT_FAC + DL_MAX_MU,               // sigmapar^2 mu'^2 + sigmaper^2 (1 - mu'^2)
    DL_PT_CT = DL_PT_SD + DL_MAX_MU,                // 0.5 (ctX + ctY)
    DL_PT_SN = DL_PT_CT + DL_MAX_EFT,               // sn / nd
    DL_PT_OM = DL_PT_SN + DL_MAX_EFT,               // [n_mu][8]: jac w_l(mu) (b1X + f mu'^2)(b1Y + f mu'^2) for l < 5, [5] = jac w_{l=0}: fused weights
    DL_PT_LQH = DL_PT_OM + 8 * DL_MAX_MU,           // log10(F_m / qper) inv_hx: the shift in units of the knot spacing (uniform knots)
    DL_PT_PART = DL_PT_LQH + DL_MAX_MU,             // [DL_FS_THREADS] partial dot products of the segmented sweeps / mu^k table of the convolution path
    DL_PT_SIZE_FAST = DL_PT_PART + 2 * DL_FIR_PAD + 8,   // the fast kernels stop here: 65 table entries, no separate-table weights (31 KB per workgroup at the
                                                    // benchmark shape: five workgroups per CU)
    DL_PT_W3 = DL_PT_PART + DL_FS_THREADS,          // [n_mu][5][3]: jac w_l(mu) (1, f mu'^2, (f mu'^2)^2): separate-table weights (general kernel only)
    DL_PT_SIZE = DL_PT_W3 + 15 * DL_MAX_MU
};

struct DlFsShared {
    double* y;     // [n_t] template power at the knots
    double* M;     // [n_t] pivot-scaled right-hand side, then spline moments (second derivatives) M[1 .. n_t-2]
    double* z;     // [n_t] forward-sweep result
    double* coef;  // per-interval polynomial in the fractional knot index, TWO PLANES: (c0, c1) of interval j at coef[2 j], (c2, c3) at coef[2 n_t + 2 j] -- lanes that
                   // evaluate neighbouring intervals then read consecutive 16-byte slots (ds_read_b128 conflict-free; [n_t][4] rows put them 32 bytes apart: 2-way)
    double* out;   // [n_in] output multipoles, staged for one coalesced store (ALIASES y, M, z: dead once coef is built)
    double* pt;    // [DL_PT_SIZE]
};

// out [n_in] + dd0 [n_kin <= n_in] alias y, M, z.  Sized to keep 4 workgroups per CU at the benchmark shape (4 x 38.9 KB <= 160 KB).
DL_HD size_t dl_fs_work_doubles(int n_t, int n_in) { size_t w = 3 * (size_t)n_t; if (2 * (size_t)n_in > w) w = 2 * (size_t)n_in; return (w + 1) & ~(size_t)1; }
DL_HD size_t dl_fs_work_doubles(int n_t, int n_in, int n_dd0) { size_t w = 3 * (size_t)n_t; if ((size_t)n_in + n_dd0 > w) w = (size_t)n_in + n_dd0; return (w + 1) & ~(size_t)1; }
DL_HD size_t dl_fs_shared_doubles(int n_t, int n_in) { return 4 * (size_t)n_t + dl_fs_work_doubles(n_t, n_in) + DL_PT_SIZE; }
DL_HD size_t dl_fs_shared_doubles_obs(const DlObsDev& o, bool fast = false) { return 4 * (size_t)o.n_t + dl_fs_work_doubles(o.n_t, o.n_in, dl_fs_n_dd0(o)) + (fast ? DL_PT_SIZE_FAST : DL_PT_SIZE); }

// PNG kernel (dl_kernels.hip): the generic layout (its coefficient region [4 n_t] holds the knot values and second derivatives of the two splines: alpha, template) |
// mu records [DL_PNG_MAX_MU][8] | scalars [16]
DL_HD size_t dl_png_shared_doubles(int n_t, int n_in) { return dl_fs_shared_doubles(n_t, n_in) + 8 * DL_PNG_MAX_MU + 16; }

// toep: layout of the convolution path (dl_fs_phase2_fir): y sits DL_FIR_PAD zeros inside the work region, M (the moments) right after the padded y
DL_HD DlFsShared dl_fs_shared_carve(double* base, int n_t, int n_in, int n_dd0 = -1, bool toep = false) {
    DlFsShared s;
    s.coef = base;                       // first: keeps the 32-byte coefficient groups 16-byte aligned
    double* work = base + 4 * (size_t)n_t;
    if (toep) { s.y = work + DL_FIR_PAD; s.M = work + (size_t)n_t + 2 * DL_FIR_PAD; s.z = s.M; }
    else { s.y = work; s.M = work + (size_t)n_t; s.z = work + 2 * (size_t)n_t; }
    s.out = work;
    s.pt = work + (n_dd0 < 0 ? dl_fs_work_doubles(n_t, n_in) : dl_fs_work_doubles(n_t, n_in, n_dd0));
    return s;
}

DL_HD void dl_ap_qparqper(const DlObsDev& o, const double* th, double& qpar, double& qper) {
    // theories/galaxy_clustering/base.py:341-350
    switch (o.apmode) {
        case 1: qpar = qper = dl_get(o.qiso, th); break;
        case 2: { double qap = dl_get(o.qap, th); qpar = pow(qap, 1. - o.eta); qper = pow(qap, -o.eta); break; }
        case 3: { double qiso = dl_get(o.qiso, th), qap = dl_get(o.qap, th); qpar = qiso * pow(qap, 1. - o.eta); qper = qiso * pow(qap, -o.eta); break; }
        default: qpar = dl_get(o.qpar, th); qper = dl_get(o.qper, th);
    }
}

// ---- phase 0 + 1: per-point scalars, per-mu AP factors and weights, template at the knots ---------------------------------------------------
// The per-mu work is a LONG dependent chain (division -> log10 / division / square root -> weights: ~150 fp64 operations deep, ~3 us) whose results
// are only needed by phase 3.  It is written in three parts so that the fast kernel can run it in a wave of its own, one part per phase, beside the
// knot / convolution / coefficient work of the other waves (in-kernel stamps: phase 01 took 4.3 us of a 9.2 us workgroup life with the chain inside it).
struct DlMuCarry {
    double qpar, qper, jac, f, b1X, b1Y, sigpar, sigper;   // per-point scalars
    double mu, iq2, x;                                      // part A
    double fac, lq, mup2;                                   // part B
    double w[DL_MAX_ELL + 1];                               // Legendre weights of the node (multipoles, then ell0): requested in part A, used in part C
};

// part A: scalars, x = factorap^2 (m: mu node, clamped by the caller)
DL_HD void dl_fs_mu_partA(const DlObsDev& o, const double* th, int m, DlMuCarry& c) {
    dl_ap_qparqper(o, th, c.qpar, c.qper);
    c.sigpar = dl_get(o.sigpar, th); c.sigper = dl_get(o.sigper, th);
    c.jac = 1. / (c.qpar * c.qper * c.qper);                  // tgc/base.py:217
    c.f = o.f_fid * dl_get(o.df, th);                          // power_template.py:757
    c.b1X = dl_get(o.b1X, th); c.b1Y = dl_get(o.b1Y, th);
    // ap_k_mu, tgc/base.py:216-222: factorap = sqrt(1 + mu^2 (1/qap^2 - 1)); muap = mu / qap / factorap.
    // Written so that the square root, the division and the logarithm all start from x = factorap^2 (short dependent chain):
    // muap^2 = mu^2 / (qap^2 x), log10(factorap / qper) = log10(x) / 2 - log10(qper)
    const double rq = c.qper / c.qpar;                         // 1 / qap
    c.iq2 = rq * rq;
    c.mu = o.mu[m];
    // (unconditional loads: behind `l < n_ell` each was a load-and-wait of its own in part C)
    for (int l = 0; l < DL_MAX_ELL; ++l) c.w[l] = o.wmu[(l < o.n_ell ? l : 0) * o.n_mu + m];
    c.w[DL_MAX_ELL] = o.wmu[(o.ell0 >= 0 ? o.ell0 : 0) * o.n_mu + m];
    c.x = 1. + c.mu * c.mu * (c.iq2 - 1.);
}

// part B: the transcendental part
DL_HD void dl_fs_mu_partB(DlMuCarry& c) {
    c.fac = sqrt(c.x);
    c.mup2 = c.mu * c.mu * c.iq2 / c.x;
    c.lq = 0.5 * log10(c.x) - log10(c.qper);                   // log10(kap) = log10(k) + log10(factorap / qper)
}

// part C: weights of mu node m -> LDS
DL_HD void dl_fs_mu_partC(const DlObsDev& o, const DlFsShared& s, int m, const DlMuCarry& c, bool w3 = true) {
    s.pt[DL_PT_FAC + m] = c.fac;
    s.pt[DL_PT_LQ + m] = c.lq;
    s.pt[DL_PT_LQH + m] = c.lq * o.inv_hx;
    // full_shape.py:492: sigmapar^2 muap^2 + sigmaper^2 (1 - muap^2)
    if (o.damping_fid) {
        // full_shape.py:410-411: exp(-k^2 (sigmapar^2 mu^2 + sigmaper^2 (1 - mu^2)) / 2) at the fiducial (k, mu); phase 3 forms (k / qper * factorap)^2 SD,
        // so SD carries qper^2 / factorap^2 (x = factorap^2)
        const double mu2 = c.mu * c.mu;
        s.pt[DL_PT_SD + m] = (c.sigpar * c.sigpar * mu2 + c.sigper * c.sigper * (1. - mu2)) * (c.qper * c.qper) / c.x;
    } else {
        s.pt[DL_PT_SD + m] = c.sigpar * c.sigpar * c.mup2 + c.sigper * c.sigper * (1. - c.mup2);
    }
    const double fm2 = c.f * c.mup2;
    const double bias = (c.b1X + fm2) * (c.b1Y + fm2);         // = b1X b1Y + (b1X + b1Y) f mu'^2 + f^2 mu'^4, full_shape.py:550
    for (int l = 0; l < DL_MAX_ELL; ++l) {
        double w = (l < o.n_ell) ? c.jac * c.w[l] : 0.;
        s.pt[DL_PT_OM + m * 8 + l] = w * bias;
        if (w3) {
            s.pt[DL_PT_W3 + (m * 5 + l) * 3 + 0] = w;
            s.pt[DL_PT_W3 + (m * 5 + l) * 3 + 1] = w * fm2;
            s.pt[DL_PT_W3 + (m * 5 + l) * 3 + 2] = w * (fm2 * fm2);
        }
    }
    s.pt[DL_PT_OM + m * 8 + 5] = (o.ell0 >= 0) ? c.jac * c.w[DL_MAX_ELL] : 0.;
    s.pt[DL_PT_OM + m * 8 + 6] = 0.;
    s.pt[DL_PT_OM + m * 8 + 7] = 0.;
}

// per-point scalars and the zero padding of the mu nodes (one thread)
DL_HD void dl_fs_scalars(const DlObsDev& o, const double* th, const DlFsShared& s, const DlMuCarry& c, bool w3 = true) {
    for (int mm = o.n_mu; mm < ((o.n_mu + 3) & ~3); ++mm) {   // pad the mu nodes to a multiple of 4 with zero weights (unrolled loops)
        s.pt[DL_PT_LQ + mm] = 0.; s.pt[DL_PT_LQH + mm] = 0.; s.pt[DL_PT_FAC + mm] = 0.; s.pt[DL_PT_SD + mm] = 0.;
        for (int q = 0; q < 8; ++q) s.pt[DL_PT_OM + mm * 8 + q] = 0.;
        if (w3) for (int q = 0; q < 15; ++q) s.pt[DL_PT_W3 + mm * 15 + q] = 0.;
    }
    s.pt[DL_PT_QPAR] = c.qpar;
    s.pt[DL_PT_QPER] = c.qper;
    s.pt[DL_PT_JAC] = c.jac;
    s.pt[DL_PT_F] = c.f;
    s.pt[DL_PT_B1X] = c.b1X;
    s.pt[DL_PT_B1Y] = c.b1Y;
    s.pt[DL_PT_SN0ND] = dl_get(o.sn0, th) / o.nd;              // full_shape.py:549
    s.pt[DL_PT_DAMP] = (c.sigpar != 0. || c.sigper != 0.) ? 1. : 0.;
    for (int q = 0; q < o.n_ct; ++q)                           // full_shape.py:630
        s.pt[DL_PT_CT + q] = 0.5 * (dl_get(o.ct_in[q][0], th) + dl_get(o.ct_in[q][1], th));
    for (int q = 0; q < o.n_sn; ++q)                           // full_shape.py:631
        s.pt[DL_PT_SN + q] = dl_get(o.sn_in[q], th) / o.nd;
}

// template at the knots (or the fixed interval polynomials), nthr threads
DL_HD void dl_fs_knots(int tid, int nthr, const DlObsDev& o, const double* th, const DlFsShared& s) {
    const int n_t = o.n_t;
    if (o.toeplitz && !o.fixed_spline && tid < 2 * DL_FIR_PAD) s.y[tid < DL_FIR_PAD ? tid - DL_FIR_PAD : n_t + tid - DL_FIR_PAD] = 0.;   // zero padding
    if (o.fixed_spline) {
        for (int j = tid; j < 4 * n_t; j += nthr) s.coef[(j & 2) * n_t + 2 * (j >> 2) + (j & 1)] = o.coef_fixed[j];   // host rows [n_t][4] -> the two planes
    } else if (o.templ == 1) {
        // power_template.py:749: exp(dm / a * tanh(a * log(k / kp)) + dn * log(k / kp))
        double dm_a = dl_get(o.dm, th) / o.a, dn = dl_get(o.dn, th);
        for (int j = tid; j < n_t; j += nthr) s.y[j] = o.pk_fid[j] * exp(dm_a * o.sf_th[j] + dn * o.sf_lg[j]);
    } else if (o.templ == 2) {
        // power_template.py:1326-1333: x = log10 k / log10 k_TO - 1, P = P_TO^(1 - m x^2) where x > 0 (below the turn-over), P_TO^(1 - n x^2) above
        const double inv_lkto = 1. / (o.lkto_fid + log10(dl_get(o.qto, th))), lp = o.lpkto_fid + log(dl_get(o.dpto, th));
        const double cm = dl_get(o.to_m, th), cn = dl_get(o.to_n, th);
        for (int j = tid; j < n_t; j += nthr) {
            const double x = o.x_t[j] * inv_lkto - 1.;
            s.y[j] = exp(lp * (1. - (x > 0. ? cm : cn) * (x * x)));
        }
    } else if (o.templ == 3) {
        // power_template.py:955-961: the fiducial P_tt modulated by the bands, over f^2 = (f_fid df)^2 (pk_fid holds P_tt_fid / f_fid^2)
        const double df = dl_get(o.df, th), inv_df2 = 1. / (df * df);
        for (int j = tid; j < n_t; j += nthr) {
            double factor = 1.;
            for (int i = 0; i < o.n_band; ++i) factor += (dl_get(o.band_in[i], th) - 1.) * o.band_tab[(size_t)i * n_t + j];
            s.y[j] = o.pk_fid[j] * factor * inv_df2;
        }
    } else {
        for (int j = tid; j < n_t; j += nthr) s.y[j] = o.pk_fid[j];
    }
}

// the whole of phase 0 + 1 in one call (general kernel, host-side constant folding): mu nodes on the last threads, scalars on thread 0
DL_HD void dl_fs_phase01(int tid, int nthr, const DlObsDev& o, const double* th, const DlFsShared& s) {
    const int mnode = nthr - 1 - tid;
    if (mnode < o.n_mu || tid == 0) {
        DlMuCarry c;
        dl_fs_mu_partA(o, th, mnode < o.n_mu ? mnode : 0, c);
        if (mnode < o.n_mu) { dl_fs_mu_partB(c); dl_fs_mu_partC(o, s, mnode, c); }
        if (tid == 0) dl_fs_scalars(o, th, s, c);
    }
    dl_fs_knots(tid, nthr, o, th, s);
}

// phase 2a: right-hand side of the reduced (n_t - 2 unknowns) not-a-knot system, pre-multiplied by the pivots
DL_HD void dl_fs_phase2a(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    int m = o.n_t - 2;
    for (int i = tid; i < m; i += nthr) {
        double r = 6. * ((s.y[i + 2] - s.y[i + 1]) * o.ih[i + 1] - (s.y[i + 1] - s.y[i]) * o.ih[i]);
        s.M[i + 1] = r * o.sp_inv[i];
    }
}

// phase 2b (two steps): forward sweep z_i = A_i z_{i-1} + B_i.  Segment `seg` covers [start, end); the state entering it,
// z_{start-1} = sum_d gf[d] B_{start-1-d}, is a dot product with precomputed products of the multipliers, computed
// cooperatively by DL_SEG_PARTS threads (independent, fully unrolled loads and FMAs).
DL_HD void dl_fs_phase2b_dot(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    const int m = o.n_t - 2;
    int seg = tid / DL_SEG_PARTS, part = tid % DL_SEG_PARTS;
    int start = seg * o.seg_len;
    double acc = 0.;
    if (start < m) {
#pragma unroll
        for (int q = 0; q < DL_SEG_QMAX; ++q) {
            int idx = start - (DL_SEG_PARTS * q + part);                  // B_{start-1-d} is stored at M[start-d]
            double g = o.sp_gf[q * DL_FS_THREADS + tid];
            double b = (idx >= 1) ? s.M[idx] : 0.;
            acc = fma(g, b, acc);
        }
    }
    s.pt[DL_PT_PART + tid] = acc;
}

DL_HD void dl_fs_phase2b(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    if (tid >= o.n_seg) return;
    const int m = o.n_t - 2;
    int start = tid * o.seg_len, end = start + o.seg_len;
    if (end > m) end = m;
    if (start >= m) return;
    const double* part = s.pt + DL_PT_PART + tid * DL_SEG_PARTS;
    double zz = (part[0] + part[1]) + (part[2] + part[3]);
#pragma unroll 8
    for (int i = start; i < end; ++i) {
        zz = fma(o.sp_A[i], zz, s.M[i + 1]);
        s.z[i] = zz;
    }
}

// phase 2c (two steps): backward sweep u_i = z_i - c'_i u_{i+1}; u_i = M[i + 1]; u_end = sum_d gb[d] z_{end+d}
DL_HD void dl_fs_phase2c_dot(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    const int m = o.n_t - 2;
    int seg = tid / DL_SEG_PARTS, part = tid % DL_SEG_PARTS;
    int start = seg * o.seg_len, end = start + o.seg_len;
    if (end > m) end = m;
    double acc = 0.;
    if (start < m) {
#pragma unroll
        for (int q = 0; q < DL_SEG_QMAX; ++q) {
            int idx = end + DL_SEG_PARTS * q + part;
            double g = o.sp_gb[q * DL_FS_THREADS + tid];
            double zv = (idx < m) ? s.z[idx] : 0.;
            acc = fma(g, zv, acc);
        }
    }
    s.pt[DL_PT_PART + tid] = acc;
}

DL_HD void dl_fs_phase2c(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    if (tid >= o.n_seg) return;
    const int m = o.n_t - 2;
    int start = tid * o.seg_len, end = start + o.seg_len;
    if (end > m) end = m;
    if (start >= m) return;
    const double* part = s.pt + DL_PT_PART + tid * DL_SEG_PARTS;
    double uu = (part[0] + part[1]) + (part[2] + part[3]);
#pragma unroll 8
    for (int i = end - 1; i >= start; --i) {
        uu = fma(o.sp_nC[i], uu, s.z[i]);
        s.M[i + 1] = uu;
    }
}

// phase 2d: moments -> polynomial of each interval j in u = (x - x0) inv_hx - j (uniform knots) or u = (x - x_j) / h_j
DL_HD void dl_fs_phase2d(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    const int n = o.n_t;
    const double hx = 1. / o.inv_hx;
    for (int j = tid; j < n - 1; j += nthr) {
        double Ml = (j == 0) ? (o.end0a * s.M[1] + o.end0b * s.M[2]) : s.M[j];
        double Mr = (j == n - 2) ? (o.end1a * s.M[n - 2] + o.end1b * s.M[n - 3]) : s.M[j + 1];
        double ihj = o.ih[j];
        double h = o.x_t[j + 1] - o.x_t[j];
        double yl = s.y[j], yr = s.y[j + 1];
        // S(t) = c0 + c1 t + c2 t^2 + c3 t^3, t = x - x_j
        double c0 = yl;
        double c1 = (yr - yl) * ihj - h * (2. * Ml + Mr) * (1. / 6.);
        double c2 = 0.5 * Ml;
        double c3 = (Mr - Ml) * ihj * (1. / 6.);
        double d0, d1, d2, d3;
        if (o.uniform_knots) {
            // t = u hx + dlt_j with dlt_j = x_j - (x0 + j hx) (rounding of the knot table, ~1e-16): exact re-expansion in u
            double dl = -o.dlt[j];
            d0 = c0 + dl * (c1 + dl * (c2 + dl * c3));
            d1 = hx * (c1 + dl * (2. * c2 + 3. * dl * c3));
            d2 = hx * hx * (c2 + 3. * dl * c3);
            d3 = hx * hx * hx * c3;
        } else {
            d0 = c0; d1 = c1 * h; d2 = c2 * h * h; d3 = c3 * h * h * h;
        }
        s.coef[2 * j + 0] = d0; s.coef[2 * j + 1] = d1; s.coef[2 * n + 2 * j + 0] = d2; s.coef[2 * n + 2 * j + 1] = d3;
    }
}

// ---- uniform knots (o.toeplitz): the two sweeps of the tridiagonal solve become ONE convolution, without a serial chain ----------------
// Uniform spacing h: M_{j-1} + 4 M_j + M_{j+1} = r_j = 6 / h^2 (y_{j-1} - 2 y_j + y_{j+1}) (2 <= j <= n-3), and the not-a-knot conditions reduce to
// M_1 = r_1 / 6, M_{n-2} = r_{n-2} / 6.  With w = (6 / h^2) sum_e t_|e| y_{j+e} (the solution on the infinite grid; y zero-padded: whatever sits
// outside the table only feeds the homogeneous part), M_j = w_j + a mu^(j-1) + b mu^(n-2-j), a = r_1 / 6 - w_1, b = r_{n-2} / 6 - w_{n-2}:
// the correction matters within ~30 knots of either end.  Each thread produces 4 consecutive w_j from one sliding window of 2 D + 4 knot values.
DL_HD void dl_fs_phase2_fir(int tid, int nthr, const DlObsDev& o, const DlFsShared& s) {
    const double T[DL_FIR_D + 1] = DL_FIR_TAPS;
    const int n = o.n_t;
    const double scale = 6. * o.inv_hx * o.inv_hx;
    for (int j0 = 4 * tid; j0 < n; j0 += 4 * nthr) {
        double acc[4] = {0., 0., 0., 0.};
        const double* yp = s.y + j0 - DL_FIR_D;   // knot j0 - D + p
        // groups of four window positions, the next group's values requested before the current one is consumed
        double cur[4], nxt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = yp[c];
#pragma unroll
        for (int gq = 0; gq < (2 * DL_FIR_D + 4) / 4; ++gq) {
            if (gq + 1 < (2 * DL_FIR_D + 4) / 4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) nxt[c] = yp[4 * (gq + 1) + c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int p = 4 * gq + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = p - DL_FIR_D - q;
                    if (e >= -DL_FIR_D && e <= DL_FIR_D) acc[q] = fma(T[e < 0 ? -e : e], cur[c], acc[q]);
                }
            }
            DL_PIN4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j0 + q < n) s.M[j0 + q] = acc[q] * scale;
    }
    // the last 2 DL_FIR_PAD + 1 threads (no window to convolve while n <= 4 (nthr - 65)) tabulate mu^k for the end corrections
    for (int k = tid - (nthr - 2 * DL_FIR_PAD - 1); k >= 0 && k <= 2 * DL_FIR_PAD; k += nthr) {
        double v = exp((double)k * DL_FIR_LN_ABS_MU);
        s.pt[DL_PT_PART + k] = (k & 1) ? -v : v;
    }
}

DL_HD double dl_fir_mu_pow(const DlFsShared& s, int k) {   // mu^k (tabulated by dl_fs_phase2_fir), 0 beyond the reach of the end corrections
    return (k <= 2 * DL_FIR_PAD) ? s.pt[DL_PT_PART + k] : 0.;
}

// moments -> interval polynomials in u = (x - x0) inv_hx - j, end corrections applied on the fly.  dlt_pref[it] = o.dlt[tid + it nthr], loaded by
// the caller at the top of the kernel (the round trip overlaps the earlier phases); iterations beyond DL_TOEP_PREF read o.dlt directly.
#define DL_TOEP_PREF 3
DL_HD void dl_fs_phase2d_toep(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, const double* dlt_pref) {
    const int n = o.n_t;
    const double hx = 1. / o.inv_hx, ihx = o.inv_hx;
    const double sc6 = o.inv_hx * o.inv_hx;       // r_j / 6
    const double a = sc6 * ((s.y[0] - s.y[1]) - (s.y[1] - s.y[2])) - s.M[1];
    const double b = sc6 * ((s.y[n - 3] - s.y[n - 2]) - (s.y[n - 2] - s.y[n - 1])) - s.M[n - 2];
    int it = 0;
    for (int j = tid; j < n - 1; j += nthr, ++it) {
        // moments of knots jl = max(j, 1), jl + 1 (and jl + 2 at the right end / jl - 1 .. for the not-a-knot extrapolation)
        auto moment = [&](int i) { return s.M[i] + a * dl_fir_mu_pow(s, i - 1) + b * dl_fir_mu_pow(s, n - 2 - i); };   // 1 <= i <= n - 2
        double Ml, Mr;
        if (j == 0) { double m1 = moment(1), m2 = moment(2); Ml = o.end0a * m1 + o.end0b * m2; Mr = m1; }
        else if (j == n - 2) { double m1 = moment(n - 2), m2 = moment(n - 3); Ml = m1; Mr = o.end1a * m1 + o.end1b * m2; }
        else { Ml = moment(j); Mr = moment(j + 1); }
        const double yl = s.y[j], yr = s.y[j + 1];
        const double c0 = yl;
        const double c1 = (yr - yl) * ihx - hx * (2. * Ml + Mr) * (1. / 6.);
        const double c2 = 0.5 * Ml;
        const double c3 = (Mr - Ml) * ihx * (1. / 6.);
        const double dl = -(it < DL_TOEP_PREF ? dlt_pref[it] : o.dlt[j]);
        s.coef[2 * j + 0] = c0 + dl * (c1 + dl * (c2 + dl * c3));
        s.coef[2 * j + 1] = hx * (c1 + dl * (2. * c2 + 3. * dl * c3));
        s.coef[2 * n + 2 * j + 0] = hx * hx * (c2 + 3. * dl * c3);
        s.coef[2 * n + 2 * j + 1] = hx * hx * hx * c3;
    }
}

// interval index j and local coordinate u of abscissa x (log10 k'); extrapolation continues the end pieces like
// scipy's fill_value='extrapolate'
template <bool UNIF>
DL_HD void dl_spline_locate(const DlObsDev& o, double x, int& j, double& u) {
    const int n = o.n_t;
    if (UNIF || o.uniform_knots) {
        // adjacent pieces of a C2 spline agree to O(eps^3) within eps of a knot: no fix-up of the index is needed
        double t = (x - o.x0) * o.inv_hx;
        double tc = t > 0. ? t : 0.;
        j = (int)tc;
        if (j > n - 2) j = n - 2;
        u = t - (double)j;
    } else {
        int lo = 0, hi = n - 1;   // x_t[lo] <= x < x_t[hi] (clamped)
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (x >= o.x_t[mid]) lo = mid; else hi = mid;
        }
        j = lo;
        u = (x - o.x_t[j]) * o.ih[j];
    }
}

template <bool UNIF>
DL_HD double dl_spline_eval(const DlObsDev& o, const DlFsShared& s, double x) {
    int j;
    double u;
    dl_spline_locate<UNIF>(o, x, j, u);
    const double* c = s.coef + 2 * j;
    const double* d = c + 2 * o.n_t;
    return fma(fma(fma(d[1], u, d[0]), u, c[1]), u, c[0]);
}

// uniform knots, abscissa already in units of the knot spacing: t = (x - x0) inv_hx
DL_HD double dl_spline_eval_t(const DlObsDev& o, const DlFsShared& s, double t) {
    int j = (int)t;                       // truncation: 0 for t in (-1, 0); clamped to [0, n_t - 2] as integers: one v_med3_i32 (n_t >= 2) instead of a 64-bit max and a min
#if defined(__HIP_DEVICE_COMPILE__)
    __asm__("v_med3_i32 %0, %1, 0, %2" : "=v"(j) : "v"(j), "s"(o.n_t - 2));
#else
    j = j < 0 ? 0 : j;
    if (j > o.n_t - 2) j = o.n_t - 2;
#endif
    const double u = t - (double)j;
    const double* c = s.coef + 2 * j;
    const double* d = c + 2 * o.n_t;
    return fma(fma(fma(d[1], u, d[0]), u, c[1]), u, c[0]);
}

// phase 3: (k, mu) evaluation, multipole projection, tracer combination; writes power (and tables).
// NL = number of multipole accumulators compiled in (3 or 5; weights of absent multipoles are zero); the mu loop is
// unrolled by 4 (nodes padded with zero weights) so that four independent evaluation chains are in flight per thread.
// FAST: uniform knots, no separate tables (straight-line inner loop); EFT: counter terms present (needs P_dd,l=0).
// The generic instantiation <false, 5, true> decides everything at run time.
// Results go to the LDS tile s.out (dl_fs_phase4 stores it): no global store, hence no store-completion wait, in the loop.
// lk_pref[it] = o.lkin[tid + it nthr] for it < DL_P3_PREF, loaded by the caller at the top of the kernel (nullptr: read here).
#define DL_P3_PREF 2
template <bool FAST, int NL, bool EFT>
DL_HD void dl_fs_phase3(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, double* tables_row, const double* lk_pref = nullptr) {
    const double qper = s.pt[DL_PT_QPER], sn0nd = s.pt[DL_PT_SN0ND];
    const double b1X = s.pt[DL_PT_B1X], b1Y = s.pt[DL_PT_B1Y];
    const bool damp = s.pt[DL_PT_DAMP] != 0.;
    const bool need_dd0 = EFT && o.n_ct > 0;
    const bool TABLES = !FAST && tables_row != nullptr;
    const int n_ell = o.n_ell, n_kin = o.n_kin;
    const int n_mu4 = (o.n_mu + 3) & ~3;
    int it = 0;
    for (int i = tid; i < n_kin; i += nthr, ++it) {
        const double lk = (lk_pref != nullptr && it < DL_P3_PREF) ? lk_pref[it] : o.lkin[i];
        const double kq = damp ? o.kin[i] / qper : 0.;   // tgc/base.py:220: kap = k / qper * factorap
        double p[NL];
        double dd[FAST ? 1 : NL], dt[FAST ? 1 : NL], tt[FAST ? 1 : NL];
        double dd0 = 0.;
#pragma unroll
        for (int l = 0; l < NL; ++l) p[l] = 0.;
#pragma unroll
        for (int l = 0; l < (FAST ? 1 : NL); ++l) dd[l] = dt[l] = tt[l] = 0.;
        for (int m0 = 0; m0 < n_mu4; m0 += 4) {
            double T[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) T[q] = dl_spline_eval<FAST>(o, s, lk + s.pt[DL_PT_LQ + m0 + q]);
            if (damp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double kap = kq * s.pt[DL_PT_FAC + m0 + q];
                    T[q] *= exp(-(kap * kap * s.pt[DL_PT_SD + m0 + q]) / 2.);  // full_shape.py:492-493
                }
            }
            if (FAST || !TABLES) {
                // fused weights: P_l = sum_m Omega_l(m) S(k'_m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double* om = s.pt + DL_PT_OM + (m0 + q) * 8;
#pragma unroll
                    for (int l = 0; l < NL; ++l) p[l] = fma(om[l], T[q], p[l]);
                    if (need_dd0) dd0 = fma(om[5], T[q], dd0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double* w3 = s.pt + DL_PT_W3 + (m0 + q) * 15;
#pragma unroll
                    for (int l = 0; l < (FAST ? 1 : NL); ++l) {
                        dd[l] = fma(w3[3 * l + 0], T[q], dd[l]);
                        dt[l] = fma(w3[3 * l + 1], T[q], dt[l]);
                        tt[l] = fma(w3[3 * l + 2], T[q], tt[l]);
                    }
                }
            }
        }
        if (!FAST && TABLES) {
#pragma unroll
            for (int l = 0; l < (FAST ? 1 : NL); ++l) {
                if (l < n_ell) {
                    p[l] = b1X * b1Y * dd[l] + (b1X + b1Y) * dt[l] + tt[l];   // full_shape.py:550
                    if (l == o.ell0) dd0 = dd[l];
                    tables_row[(size_t)(0 * n_ell + l) * n_kin + i] = dd[l];
                    tables_row[(size_t)(1 * n_ell + l) * n_kin + i] = dt[l];
                    tables_row[(size_t)(2 * n_ell + l) * n_kin + i] = tt[l];
                }
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l < n_ell) {
                double pl = p[l] + (l == o.ell0 ? sn0nd : 0.);
                if (EFT && o.n_ct > 0) {  // full_shape.py:633
                    double acc = 0.;
                    for (int c = 0; c < o.n_ct; ++c) acc += o.ct_matrix[((size_t)l * n_kin + i) * o.n_ct + c] * s.pt[DL_PT_CT + c];
                    pl += acc * dd0;
                }
                if (EFT && o.n_sn > 0) {  // full_shape.py:634
                    double acc = 0.;
                    for (int c = 0; c < o.n_sn; ++c) acc += o.sn_matrix[((size_t)l * n_kin + i) * o.n_sn + c] * s.pt[DL_PT_SN + c];
                    pl += acc;
                }
                s.out[(size_t)l * n_kin + i] = pl;
            }
        }
        if (EFT && o.n_var > 0) s.out[o.n_in + i] = dd0;   // needed by the derivative rows (phase 4)
    }
}

// FAST variant of phase 3 that walks TWO wavenumbers (i, i + nthr) per pass of the mu loop: the per-mu shifts and fused weights are read from LDS
// once for both (phase 3 is bound by LDS bandwidth: 32 B of interval coefficients + ~40 B of weights per evaluation), and eight independent
// evaluation chains are in flight.  Same arithmetic, in the same order, per wavenumber as dl_fs_phase3<true, NL, EFT>.
template <int NL, bool EFT>
DL_HD void dl_fs_phase3_pair(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, const double* lk_pref = nullptr) {
    const double qper = s.pt[DL_PT_QPER], sn0nd = s.pt[DL_PT_SN0ND];
    const bool damp = s.pt[DL_PT_DAMP] != 0.;
    const bool need_dd0 = EFT && o.n_ct > 0;
    const int n_ell = o.n_ell, n_kin = o.n_kin;
    const int n_mu4 = (o.n_mu + 3) & ~3;
    auto finish = [&](int i, const double* p, double dd0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l < n_ell) {
                double pl = p[l] + (l == o.ell0 ? sn0nd : 0.);
                if (EFT && o.n_ct > 0) {  // full_shape.py:633
                    double acc = 0.;
                    for (int c = 0; c < o.n_ct; ++c) acc += o.ct_matrix[((size_t)l * n_kin + i) * o.n_ct + c] * s.pt[DL_PT_CT + c];
                    pl += acc * dd0;
                }
                if (EFT && o.n_sn > 0) {  // full_shape.py:634
                    double acc = 0.;
                    for (int c = 0; c < o.n_sn; ++c) acc += o.sn_matrix[((size_t)l * n_kin + i) * o.n_sn + c] * s.pt[DL_PT_SN + c];
                    pl += acc;
                }
                s.out[(size_t)l * n_kin + i] = pl;
            }
        }
        if (EFT && o.n_var > 0) s.out[o.n_in + i] = dd0;   // needed by the derivative rows (phase 4)
    };
    int it = 0;
    for (int i = tid; i < n_kin; i += 2 * nthr, it += 2) {
        const int i2 = i + nthr;
        const bool two = i2 < n_kin;
        // abscissae in units of the knot spacing: t = (log10 k - x0) inv_hx + log10(F_m / qper) inv_hx
        const double lkA = (((lk_pref != nullptr && it < DL_P3_PREF) ? lk_pref[it] : o.lkin[i]) - o.x0) * o.inv_hx;
        const double lkB = !two ? lkA : (((lk_pref != nullptr && it + 1 < DL_P3_PREF) ? lk_pref[it + 1] : o.lkin[i2]) - o.x0) * o.inv_hx;
        const double kqA = damp ? o.kin[i] / qper : 0.;   // tgc/base.py:220: kap = k / qper * factorap
        const double kqB = (damp && two) ? o.kin[i2] / qper : 0.;
        double pA[NL], pB[NL];
        double dd0A = 0., dd0B = 0.;
#pragma unroll
        for (int l = 0; l < NL; ++l) pA[l] = pB[l] = 0.;
        for (int m0 = 0; m0 < n_mu4; m0 += 4) {
            double lq[4], TA[4], TB[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) lq[q] = s.pt[DL_PT_LQH + m0 + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) TA[q] = dl_spline_eval_t(o, s, lkA + lq[q]);
            if (two) {
#pragma unroll
                for (int q = 0; q < 4; ++q) TB[q] = dl_spline_eval_t(o, s, lkB + lq[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) TB[q] = 0.;
            }
            if (damp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double fac = s.pt[DL_PT_FAC + m0 + q], sd = s.pt[DL_PT_SD + m0 + q];
                    double kap = kqA * fac;
                    TA[q] *= exp(-(kap * kap * sd) / 2.);  // full_shape.py:492-493
                    if (two) { kap = kqB * fac; TB[q] *= exp(-(kap * kap * sd) / 2.); }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* om = s.pt + DL_PT_OM + (m0 + q) * 8;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double w = om[l];
                    pA[l] = fma(w, TA[q], pA[l]);
                    pB[l] = fma(w, TB[q], pB[l]);
                }
                if (need_dd0) { const double w = om[5]; dd0A = fma(w, TA[q], dd0A); dd0B = fma(w, TB[q], dd0B); }
            }
        }
        finish(i, pA, dd0A);
        if (two) finish(i2, pB, dd0B);
    }
}

// phase 4: coalesced store of the staged multipoles
// power_row: row 0 of this point (already offset by col_offset); ld: leading dimension of the power buffer
DL_HD void dl_fs_phase4(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, const double* th, double* power_row, int64_t ld) {
    for (int idx = tid; idx < o.n_in; idx += nthr) DL_STREAM_STORE(&power_row[idx], s.out[idx]);
    for (int c = tid; c < o.n_pass; c += nthr) { const int col = (int)o.pass_tab[2 * c]; power_row[o.n_in + c] = col >= 0 ? th[col] : o.pass_tab[2 * c + 1]; }
    if (o.n_var > 0 && o.n_ct > 0) {
        // d(power)/d(ct) = 0.5 ct_matrix[:, c] P_dd,l=0 per tracer (full_shape.py:630, 633): rows 1 + slot of this point
        for (int c = 0; c < o.n_ct; ++c) {
            for (int t = 0; t < 2; ++t) {
                int slot = o.marg_ct_slot[c][t];
                if (slot < 0) continue;
                if (t == 1 && o.marg_ct_slot[c][0] == slot) continue;   // both tracers feed the same parameter: handled at t = 0 with weight 1
                double wgt = (o.marg_ct_slot[c][0] == o.marg_ct_slot[c][1]) ? 1. : 0.5;
                double* drow = power_row + (size_t)(1 + slot) * ld;
                for (int idx = tid; idx < o.n_in; idx += nthr) {
                    int i = idx % o.n_kin;
                    drow[idx] = wgt * o.ct_matrix[(size_t)idx * o.n_ct + c] * s.out[o.n_in + i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// PNG theory (kind 5; primordial_non_gaussianity.py:75-112): phases of dl_png_kernel (dl_kernels.hip), shared with the CPU emulation.
//   murec [n_mu][8]: log10(factorap / qper), factorap, mu'^2, w_4, w_0 .. w_3;  sc [16]: qper, jac, f, b1X, b1Y, bfnlX, bfnlY, sX^2 / 2, sY^2 / 2, sn0 / nd
// ------------------------------------------------------------------------------------------------------------------------
DL_HD void dl_png_setup(int tid, int nthr, const DlObsDev& o, const double* th, double* murec, double* sc) {
    if (tid == nthr - 1) {
        double qpar, qper;
        dl_ap_qparqper(o, th, qpar, qper);
        const double b1X = dl_get(o.b1X, th), b1Y = dl_get(o.b1Y, th), fnl = dl_get(o.fnl, th);
        sc[0] = qper; sc[1] = 1. / (qpar * qper * qper); sc[2] = o.f_fid * dl_get(o.df, th); sc[3] = b1X; sc[4] = b1Y;
        // primordial_non_gaussianity.py:97-104
        sc[5] = o.png_mode == 0 ? dl_get(o.bphiX, th) * fnl : 2. * 1.686 * (b1X - dl_get(o.pX, th)) * fnl;
        sc[6] = o.png_mode == 0 ? dl_get(o.bphiY, th) * fnl : 2. * 1.686 * (b1Y - dl_get(o.pY, th)) * fnl;
        const double sX = dl_get(o.sigmas, th), sY = dl_get(o.sigmasY, th);
        sc[7] = 0.5 * sX * sX; sc[8] = 0.5 * sY * sY; sc[9] = dl_get(o.sn0, th) / o.nd;
        sc[10] = dl_get(o.bv, th) * sc[2] * o.png_velfac; sc[11] = dl_get(o.sigmau, th);      // velocity variant: bv f 100 / (1 + z), sigma_u
    }
    if (tid >= nthr - 1 - o.n_mu && tid < nthr - 1) {
        const int m = nthr - 2 - tid;
        double qpar, qper;
        dl_ap_qparqper(o, th, qpar, qper);
        const double mu = o.mu[m], rq = qper / qpar;
        const double x = 1. + mu * mu * (rq * rq - 1.);       // factorap^2, tgc/base.py:216-222
        murec[8 * m] = 0.5 * log10(x) - log10(qper);
        murec[8 * m + 1] = sqrt(x);
        murec[8 * m + 2] = mu * mu * rq * rq / x;
        murec[8 * m + 3] = o.n_ell > 4 ? o.wmu[4 * o.n_mu + m] : 0.;
        for (int l = 0; l < 4; ++l) murec[8 * m + 4 + l] = l < o.n_ell ? o.wmu[l * o.n_mu + m] : 0.;
    }
}

// alpha (alpha_fid sqrt(norm / template factor): alpha ~ 1 / sqrt(P), lines 86, 89-93) or the template at the knots
DL_HD void dl_png_knots(int tid, int nthr, const DlObsDev& o, const double* th, const DlFsShared& s, bool alpha) {
    const int n_t = o.n_t;
    const double dm_a = dl_get(o.dm, th) / o.a, dn = dl_get(o.dn, th);
    if (o.toeplitz && tid < 2 * DL_FIR_PAD) s.y[tid < DL_FIR_PAD ? tid - DL_FIR_PAD : n_t + tid - DL_FIR_PAD] = 0.;
    const double norm = (alpha && o.templ == 1) ? exp(dm_a * o.png_th0 + dn * o.png_lg0) : 1.;
    for (int j = tid; j < n_t; j += nthr) {
        const double fac = o.templ == 1 ? exp(dm_a * o.sf_th[j] + dn * o.sf_lg[j]) : 1.;
        s.y[j] = alpha ? o.png_alpha[j] * sqrt(norm / fac) : o.pk_fid[j] * fac;
    }
}

// knot values and second derivatives of the spline just built on s.y (toep: s.M holds the convolution w, the end corrections are applied here as dl_fs_phase2d_toep
// does; otherwise s.M [1 .. n - 2] are final) -> yout [n_t], mout [n_t] (two workgroups per CU instead of one: 2 n_t doubles per spline, not 4 n_t of interval polynomials)
DL_HD void dl_png_keep_spline(int tid, int nthr, const DlObsDev& o, const DlFsShared& s, bool toep, double* yout, double* mout) {
    const int n = o.n_t;
    double a = 0., b = 0.;
    if (toep) {
        const double sc6 = o.inv_hx * o.inv_hx;
        a = sc6 * ((s.y[0] - s.y[1]) - (s.y[1] - s.y[2])) - s.M[1];
        b = sc6 * ((s.y[n - 3] - s.y[n - 2]) - (s.y[n - 2] - s.y[n - 1])) - s.M[n - 2];
    }
    auto moment = [&](int i) { return toep ? s.M[i] + a * dl_fir_mu_pow(s, i - 1) + b * dl_fir_mu_pow(s, n - 2 - i) : s.M[i]; };   // 1 <= i <= n - 2
    for (int j = tid; j < n; j += nthr) {
        double m;
        if (j == 0) m = o.end0a * moment(1) + o.end0b * moment(2);
        else if (j == n - 1) m = o.end1a * moment(n - 2) + o.end1b * moment(n - 3);
        else m = moment(j);
        mout[j] = m;
        yout[j] = s.y[j];
    }
}

DL_HD double dl_png_spline_value(const DlObsDev& o, const double* y, const double* m, int j, double x) {
    const double xl = o.x_t[j], xr = o.x_t[j + 1], h = xr - xl;
    const double a = (xr - x) / h, b = (x - xl) / h;
    return a * y[j] + b * y[j + 1] + ((a * a * a - a) * m[j] + (b * b * b - b) * m[j + 1]) * (h * h) * (1. / 6.);
}

// (k, mu) evaluation of both splines, bias, damping, projection (lines 107-112): out [n_ell][n_kin]; tabs = alpha knots | alpha second derivatives | template knots | ...
DL_HD void dl_png_eval(int tid, int nthr, const DlObsDev& o, const double* tabs, const double* murec, const double* sc, double* out) {
    const int n_t = o.n_t;
    const double qper = sc[0], jac = sc[1], f = sc[2], b1X = sc[3], b1Y = sc[4], bfX = sc[5], bfY = sc[6], hsX = sc[7], hsY = sc[8], sn0nd = sc[9];
    for (int ik = tid; ik < o.n_kin; ik += nthr) {
        const double lk = o.lkin[ik], kq = o.kin[ik] / qper;
        double acc[DL_MAX_ELL] = {0., 0., 0., 0., 0.};
        for (int m = 0; m < o.n_mu; ++m) {
            const double* r = murec + 8 * m;
            const double x = lk + r[0];
            int j; double u;
            dl_spline_locate<false>(o, x, j, u);
            const double al = dl_png_spline_value(o, tabs, tabs + n_t, j, x);
            const double pk = dl_png_spline_value(o, tabs + 2 * n_t, tabs + 3 * n_t, j, x);
            const double kap = kq * r[1], mup2 = r[2];
            const double km2 = kap * kap * mup2;
            const double fog = 1. / ((1. + hsX * km2) * (1. + hsY * km2));
            const double fm2 = f * mup2;
            double pkmu;
            if (o.png_vel) {   // lines 313-319: no stochastic term, one damping scale times sinc(sigma_u k') (numpy's sinc: sin(pi x) / (pi x))
                const double xs = sc[11] * kap, snc = xs == 0. ? 1. : sin(3.141592653589793 * xs) / (3.141592653589793 * xs);
                pkmu = jac * (snc / (1. + hsX * km2)) * (b1X + bfX * al + fm2) * (sc[10] * sqrt(mup2) / kap) * pk;
            } else pkmu = jac * fog * (b1X + bfX * al + fm2) * (b1Y + bfY * al + fm2) * pk + sn0nd;   // lines 108-111
            for (int l = 0; l < 4; ++l) acc[l] = fma(r[4 + l], pkmu, acc[l]);
            acc[4] = fma(r[3], pkmu, acc[4]);
        }
        for (int l = 0; l < o.n_ell; ++l) out[(size_t)l * o.n_kin + ik] = acc[l];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// BAO wiggle model, 'standard' (Chen 2023): DampedBAOWigglesPowerSpectrumMultipoles.calculate bao.py:117-140.
//   P(k, mu) = B(k, mu) P_now(k) + C(k', mu') [P_dd - P_now](k'),  P_ell = sum_mu w_ell(mu) P(k, mu)   (no Jacobian in this model)
// The BAO template does not change P(k) (power_template.py:372-376): both splines are constants, held as interval polynomials in global
// memory (L1 / L2 resident).  Phase A: per-mu AP factors; phase B: one k per thread, mu loop unrolled by 4; output staged in LDS.
// ------------------------------------------------------------------------------------------------------------------------
enum { DL_BAO_QPER = 0, DL_BAO_F, DL_BAO_B1, DL_BAO_SIGS, DL_BAO_D, DL_BAO_LQ = 8, DL_BAO_FAC = DL_BAO_LQ + DL_MAX_MU, DL_BAO_MUP2 = DL_BAO_FAC + DL_MAX_MU,
       DL_BAO_SD = DL_BAO_MUP2 + DL_MAX_MU, DL_BAO_SDF = DL_BAO_SD + DL_MAX_MU, DL_BAO_ML = DL_BAO_SDF + DL_MAX_MU, DL_BAO_REC = DL_BAO_ML + DL_MAX_ML,
       DL_BAO_PT = DL_BAO_REC + 12 * DL_MAX_MU };
// DL_BAO_REC: per-mu records of the 'standard' model's fast path, 12 doubles each: lq / hx, 1/2 fac^2 SD, 1/2 (sigmas mu)^2, f mu^2, f mu'^2, -, w_0 .. w_4, -

DL_HD size_t dl_bao_shared_doubles(int n_in) { return DL_BAO_PT + (size_t)n_in; }

DL_HD void dl_bao_phaseA(int tid, int nthr, const DlObsDev& o, const double* th, double* lds) {
    if (tid < ((o.n_mu + 3) & ~3) || tid == 0) {
        double qpar, qper;
        dl_ap_qparqper(o, th, qpar, qper);
        if (tid < o.n_mu) {
            double sigpar = dl_get(o.sigpar, th), sigper = dl_get(o.sigper, th);
            double qap = qpar / qper;
            double mu = o.mu[tid];
            double fac = sqrt(1. + mu * mu * (1. / (qap * qap) - 1.));   // tgc/base.py:218
            double mup = mu / qap / fac;
            lds[DL_BAO_FAC + tid] = fac;
            lds[DL_BAO_LQ + tid] = log10(fac / qper);
            lds[DL_BAO_MUP2 + tid] = mup * mup;
            lds[DL_BAO_SD + tid] = sigpar * sigpar * (mup * mup) + sigper * sigper * (1. - mup * mup);   // bao.py:129
            lds[DL_BAO_SDF + tid] = sigpar * sigpar * (mu * mu) + sigper * sigper * (1. - mu * mu);       // 'fix-damping': fiducial mu (bao.py:137-138)
            // record of the fast path (dl_bao_phaseB_std): everything of the (k, mu) evaluation that depends on mu only
            double* rec = lds + DL_BAO_REC + 12 * tid;
            const double f = dl_get(o.dbeta, th) * (o.f_fid * dl_get(o.df, th)), sigmas = dl_get(o.sigmas, th);
            rec[0] = lds[DL_BAO_LQ + tid] * o.inv_hx;
            rec[1] = 0.5 * (fac * fac) * lds[DL_BAO_SD + tid];
            rec[2] = 0.5 * (sigmas * mu) * (sigmas * mu);
            rec[3] = f * (mu * mu);
            rec[4] = f * (mup * mup);
            rec[5] = 0.;
            for (int l = 0; l < DL_MAX_ELL; ++l) rec[6 + l] = (l < o.n_ell) ? o.wmu[l * o.n_mu + tid] : 0.;
            rec[11] = 0.;
        } else if (tid < ((o.n_mu + 3) & ~3)) {
            lds[DL_BAO_FAC + tid] = 0.; lds[DL_BAO_LQ + tid] = 0.; lds[DL_BAO_MUP2 + tid] = 0.; lds[DL_BAO_SD + tid] = 0.; lds[DL_BAO_SDF + tid] = 0.;
            for (int q = 0; q < 12; ++q) lds[DL_BAO_REC + 12 * tid + q] = 0.;    // zero weights: the padded nodes add nothing
        }
        if (tid == 0) {
            lds[DL_BAO_QPER] = qper;
            lds[DL_BAO_F] = dl_get(o.dbeta, th) * (o.f_fid * dl_get(o.df, th));   // bao.py:119 with power_template.py:374
            lds[DL_BAO_B1] = dl_get(o.b1X, th);
            lds[DL_BAO_SIGS] = dl_get(o.sigmas, th);
            lds[DL_BAO_D] = dl_get(o.dres, th);
            for (int i = 0; i < o.n_ml; ++i) { const int col = (int)o.ml_tab[3 * i]; lds[DL_BAO_ML + i] = col >= 0 ? th[col] : o.ml_tab[3 * i + 1]; }
        }
    }
}

// 1 / x for x >= 1, to rounding: hardware seed + two Newton steps on the device (6 instructions against ~20 of the IEEE division).  x is clamped to 1e300 first:
// the Newton step of an infinite x would be inf * 0 (the callers square the result: 1e-600 is the same 0 as 1 / inf^2; a NaN x only comes from NaN inputs, which
// the finalize kernels flag on their own)
DL_HD double dl_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = fmin(x, 1e300);
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
#else
    return 1. / x;
#endif
}

// 'standard' model on uniform knots (the reference's BAO templates: geomspace tables), bao.py:117-136 -- same formula as dl_bao_phaseB_m<0>, with everything that
// depends on mu only taken from the per-mu records of phase A (LDS broadcasts: no global loads of nodes / weights in the loop), the abscissa in units of the knot
// spacing (one add, integer clamp), the Finger-of-God factor 1 / (1 + (sigmas k mu)^2 / 2)^2 through dl_rcp.  NL = multipole accumulators compiled in.
template <int NL>
DL_HD void dl_bao_phaseB_std(int tid, int nthr, const DlObsDev& o, double* lds) {
    const double qper = lds[DL_BAO_QPER], b1 = lds[DL_BAO_B1];
    const int n_kin = o.n_kin, nm2 = o.n_t - 2;
    const int reciso = (o.bao_mode & 15) == 1;
    double* out = lds + DL_BAO_PT;
    for (int i = tid; i < n_kin; i += nthr) {
        const double kk = o.kin[i], pknow = o.pknow_k[i];
        const double lkh = (o.lkin[i] - o.x0) * o.inv_hx;
        const double kq = kk / qper, kq2 = kq * kq, kk2 = kk * kk;
        double omsk = 1.;
        if (reciso) { double kr = kk * o.smoothing_radius; omsk = 1. - exp(-0.5 * (kr * kr)); }   // bao.py:131, fiducial coordinates
        double p[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) p[l] = 0.;
        auto evaluate = [&](int m) {
            const double* rec = lds + DL_BAO_REC + 12 * m;
            const double t = lkh + rec[0];
            int j = (int)t;
            j = j < 0 ? 0 : (j > nm2 ? nm2 : j);
            const double u = t - (double)j;
            const double* c = o.coef_w + 4 * (size_t)j;
            const double pkw = fma(fma(fma(c[3], u, c[2]), u, c[1]), u, c[0]);                     // [P_dd - P_now](k')
            const double ca = fma(rec[4], omsk, b1), cb = fma(rec[3], omsk, b1);                   // b1 + f mu'^2 (1 - S(k)), b1 + f mu^2 (1 - S(k))
            const double Cap = ca * ca * exp(-(kq2 * rec[1]));                                     // bao.py:129-132
            const double r = dl_rcp(fma(kk2, rec[2], 1.));                                         // bao.py:133
            return cb * cb * (r * r) * pknow + Cap * pkw;                                          // bao.py:134-136
        };
        auto accumulate = [&](int m, double pkmu) {
            const double* w = lds + DL_BAO_REC + 12 * m + 6;
#pragma unroll
            for (int l = 0; l < NL; ++l) p[l] = fma(w[l], pkmu, p[l]);
        };
        // four nodes at a time (independent chains), then the remainder by two and by one: the reference's 10 nodes are 4 + 4 + 2, not three groups of four with
        // two zero-weight evaluations
        const int n_mu = o.n_mu;
        int m0 = 0;
        for (; m0 + 4 <= n_mu; m0 += 4) {
            double pkmu[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pkmu[q] = evaluate(m0 + q);
#pragma unroll
            for (int q = 0; q < 4; ++q) accumulate(m0 + q, pkmu[q]);
        }
        if (m0 + 2 <= n_mu) {
            const double e0 = evaluate(m0), e1 = evaluate(m0 + 1);
            accumulate(m0, e0); accumulate(m0 + 1, e1);
            m0 += 2;
        }
        if (m0 < n_mu) accumulate(m0, evaluate(m0));
#pragma unroll
        for (int l = 0; l < NL; ++l)
            if (l < o.n_ell) out[(size_t)l * n_kin + i] = p[l];
    }
}

template <int MODEL>   // 0: 'standard', 1: 'fix-damping' / 'move-all' / 'fog-damping' family, 2: resummed wiggles, 3: flexible wiggles
DL_HD void dl_bao_phaseB_m(int tid, int nthr, const DlObsDev& o, double* lds) {
    if (MODEL == 0 && o.uniform_knots && o.n_mu <= DL_MAX_MU - 3) {
        if (o.n_ell <= 3) dl_bao_phaseB_std<3>(tid, nthr, o, lds); else dl_bao_phaseB_std<DL_MAX_ELL>(tid, nthr, o, lds);
        return;
    }
    const double qper = lds[DL_BAO_QPER], f = lds[DL_BAO_F], b1 = lds[DL_BAO_B1], sigmas = lds[DL_BAO_SIGS];
    const int n_ell = o.n_ell, n_mu = o.n_mu, n_kin = o.n_kin, n_mu4 = (o.n_mu + 3) & ~3;
    const int reciso = (o.bao_mode & 15) == 1, model = o.bao_mode >> 4;
    const bool fix_damping = model & 1, move_all = model & 2, fog_damping = model & 4;
    double* out = lds + DL_BAO_PT;
    for (int i = tid; i < n_kin; i += nthr) {
        const double kk = o.kin[i], lk = o.lkin[i], pknow = o.pknow_k[i];
        const double kq = kk / qper;
        double sk = 0.;
        if (reciso) { double kr = kk * o.smoothing_radius; sk = exp(-0.5 * (kr * kr)); }   // bao.py:131, fiducial coordinates
        double p[DL_MAX_ELL];
#pragma unroll
        for (int l = 0; l < DL_MAX_ELL; ++l) p[l] = 0.;
        double mult[DL_MAX_ELL];   // flexible wiggles: delta_{ell 0} + sum_i ml_i K_i(k) of each multipole (bao.py:363-366), at the fiducial k
        if (MODEL == 3) {
#pragma unroll
            for (int l = 0; l < DL_MAX_ELL; ++l) mult[l] = (l == o.ell0) ? 1. : 0.;
            for (int q = 0; q < o.n_ml; ++q) {
                const double v = lds[DL_BAO_ML + q] * o.ct_matrix[(size_t)q * n_kin + i];
                const int lq = (int)o.ml_tab[3 * q + 2];
#pragma unroll
                for (int l = 0; l < DL_MAX_ELL; ++l) if (l == lq) mult[l] += v;
            }
        }
        for (int m0 = 0; m0 < n_mu4; m0 += 4) {
            double pkmu[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int m = m0 + q;
                int j;
                double u;
                dl_spline_locate<false>(o, lk + lds[DL_BAO_LQ + m], j, u);
                const double* c = o.coef_w + 4 * (size_t)j;
                double pkw = fma(fma(fma(c[3], u, c[2]), u, c[1]), u, c[0]);                    // [P_dd - P_now](k')
                double kap = kq * lds[DL_BAO_FAC + m];
                double mup2 = lds[DL_BAO_MUP2 + m];
                double mu = (m < n_mu) ? o.mu[m] : 0.;
                if (MODEL == 0) {   // 'standard' (Chen 2023); compile-time: as a run-time branch inside the unrolled mu loop it cost 27 % of the kernel
                    double ca = b1 + f * mup2 * (1. - sk);
                    double Cap = ca * ca * exp(-(kap * kap * lds[DL_BAO_SD + m]) / 2.);         // bao.py:129-132
                    double sm = sigmas * kk * mu;
                    double den = 1. + sm * sm / 2.;
                    double fog = 1. / (den * den);                                               // bao.py:133
                    double cb = b1 + f * mu * mu * (1. - sk);
                    pkmu[q] = cb * cb * fog * pknow + Cap * pkw;                                 // bao.py:134-136
                } else {            // bao.py:137-150
                    const double* cn = o.coef_n + 4 * (size_t)j;
                    double pknowap = fma(fma(fma(cn[3], u, cn[2]), u, cn[1]), u, cn[0]);        // P_now(k')
                    // damping of the wiggles: at the fiducial (k, mu) ('fix-damping') or at the distorted ones; SD[m] holds the distorted-mu combination
                    double dw;
                    if (MODEL == 3) {
                        // wiggles(k') / P_now(k') times sum_ell mult_ell(k) L_ell(mu), Legendre polynomials at the fiducial mu (bao.py:360-368, 375)
                        double lsum = 0.;
#pragma unroll
                        for (int l = 0; l < DL_MAX_ELL; ++l)
                            if (l < n_ell && m < n_mu) lsum = fma(mult[l], o.sn_matrix[l * n_mu + m], lsum);
                        dw = lsum * pkw / pknowap;
                    } else if (MODEL == 2) {
                        // ResummedPowerSpectrumWiggles.wiggles at (k', mu') (bao.py:201-222): b1 Eulerian bias, d rescales the growth factor
                        const double dd = lds[DL_BAO_D], fm = f * mup2;
                        const double e0 = -0.5 * (1. + f * (f + 2.) * mup2) * kap * kap * dd * dd;
                        const double sdd2 = o.res_sig[0] + o.res_sig[3] / (b1 * b1);
                        double rw;
                        if (reciso) {
                            double kr = kap * o.smoothing_radius;
                            const double skr = exp(-0.5 * (kr * kr)), skc = 1. - skr;
                            const double t = b1 + fm * skc - skr;
                            const double sds2 = (1. + fm) * sdd2 + f * (1. + f) * mup2 * o.res_sig[2];
                            const double sss2 = sdd2 + f * fm * o.res_sig[1] + 2. * fm * o.res_sig[2];
                            rw = t * t * exp(e0 * sdd2) + 2. * t * (1. + fm) * skr * exp(e0 * sds2) + (1. + fm) * (1. + fm) * skr * skr * exp(e0 * sss2);
                        } else {
                            rw = (b1 + fm) * (b1 + fm) * exp(e0 * sdd2);
                        }
                        dw = rw * pkw / pknowap;
                    } else {
                        double snl2 = fix_damping ? kk * kk * lds[DL_BAO_SDF + m] : kap * kap * lds[DL_BAO_SD + m];
                        dw = pkw / pknowap * exp(-snl2 / 2.);
                    }
                    // smooth part: everything at the distorted (k', mu') ('move-all') or at the fiducial ones
                    double ks = move_all ? kap : kk, mus2 = move_all ? mup2 : mu * mu;
                    double pkn = move_all ? pknowap : pknow;
                    double sm2 = sigmas * sigmas * ks * ks * mus2;
                    double den = 1. + sm2 / 2.;
                    double fog = 1. / (den * den);
                    double sks = sk;
                    if (reciso && move_all) { double kr = ks * o.smoothing_radius; sks = exp(-0.5 * (kr * kr)); }
                    double cb = b1 + f * mus2 * (1. - sks);
                    double smooth = cb * cb * pkn;
                    if (MODEL == 3) pkmu[q] = smooth * (1. + dw);                                    // no damping, no Finger-of-God (bao.py:382)
                    else pkmu[q] = fog_damping ? smooth * fog * (1. + dw) : smooth * (fog + dw);     // Beutler 2016 / Howlett 2023
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int m = m0 + q;
                if (m < n_mu) {
#pragma unroll
                    for (int l = 0; l < DL_MAX_ELL; ++l)
                        if (l < n_ell) p[l] = fma(o.wmu[l * n_mu + m], pkmu[q], p[l]);
                }
            }
        }
#pragma unroll
        for (int l = 0; l < DL_MAX_ELL; ++l)
            if (l < n_ell) out[(size_t)l * n_kin + i] = p[l];
    }
}

DL_HD void dl_bao_phaseB(int tid, int nthr, const DlObsDev& o, double* lds) {
    const int model = o.bao_mode >> 4;
    if (model == 0) dl_bao_phaseB_m<0>(tid, nthr, o, lds);
    else if (model & 32) dl_bao_phaseB_m<3>(tid, nthr, o, lds);
    else if (model & 16) dl_bao_phaseB_m<2>(tid, nthr, o, lds);
    else dl_bao_phaseB_m<1>(tid, nthr, o, lds);
}

// coalesced store of the multipoles + pass-through columns
DL_HD void dl_store_with_pass(int tid, int nthr, const DlObsDev& o, const double* th, const double* out, double* power_row) {
    for (int idx = tid; idx < o.n_in; idx += nthr) power_row[idx] = out[idx];
    for (int c = tid; c < o.n_pass; c += nthr) { const int col = (int)o.pass_tab[2 * c]; power_row[o.n_in + c] = col >= 0 ? th[col] : o.pass_tab[2 * c + 1]; }
}

// ------------------------------------------------------------------------------------------------------------------------
// Emulated theory (SURVEY.md rows a12 + the velocileptors part of a5): feature vector of one point.
// LDS: x [DL_MAX_X] | buf0 [DL_MAX_WIDTH] | buf1 [DL_MAX_WIDTH] | scal [4] | mono [(1 + n_var) * DL_N_MONO]
// ------------------------------------------------------------------------------------------------------------------------
enum { DL_EM_X = 0, DL_EM_BUF0 = DL_MAX_X, DL_EM_BUF1 = DL_EM_BUF0 + DL_MAX_WIDTH + 8, DL_EM_SCAL = DL_EM_BUF1 + DL_MAX_WIDTH + 8, DL_EM_MONO = DL_EM_SCAL + 4 };
DL_HD size_t dl_emu_shared_doubles(int n_var) { return DL_EM_MONO + (size_t)(1 + n_var) * DL_N_MONO; }
DL_HD size_t dl_emu_shared_doubles_obs(const DlObsDev& o) {   // (stacked table engine: + the basis of one group + the scaled inputs)
    return dl_emu_shared_doubles(o.n_var) + (o.eng[0].type == 2 ? (size_t)o.stk.max_k + 8 + DL_MAX_X : 0);
}

DL_HD double dl_activation(int act, double v) {
    if (act == 0) return v / (1. + exp(-v));      // silu, conversion.py:29
    if (act == 1) return v > 0. ? v : 0.;         // relu, conversion.py:31
    return tanh(v);                                // tanh, conversion.py:33
}

DL_HD double dl_ipow(double x, int p) {
    double r = 1.;
    for (int i = 0; i < p; ++i) r *= x;
    return r;
}

// one dense layer of engine e: out[j] = act(bias[j] + sum_i in[i] K[i, j]); thread j computes unit j
DL_HD void dl_emu_layer(int tid, const DlObsDev::Engine& e, int layer, const double* w, const double* in, double* out, bool activate) {
    const int nin = e.widths[layer], nout = e.widths[layer + 1];
    if (tid < nout) {
        const double* kernel = w;
        double acc0 = w[(size_t)nin * nout + tid], acc1 = 0.;
        int i = 0;
        for (; i + 2 <= nin; i += 2) {
            acc0 = fma(in[i], kernel[(size_t)i * nout + tid], acc0);
            acc1 = fma(in[i + 1], kernel[(size_t)(i + 1) * nout + tid], acc1);
        }
        if (i < nin) acc0 = fma(in[i], kernel[(size_t)i * nout + tid], acc0);
        double v = acc0 + acc1;
        out[tid] = activate ? dl_activation(e.act, v) : v;
    }
}

// velocileptors 'pars' -> 19 monomials, and the monomials' derivatives w.r.t. analytically solved alpha* / sn* (full_shape.py:1182-1186, 1300-1307, 1577-1592, 1479-1488)
// The arithmetic in three pieces, so that the fused forward pass (dl_emu_batch.h) can spread the writes of a point over lanes with the SAME operations in the same order:
// 'pars' (full_shape.py:1300-1307, 1577-1592; co-evolution 1479-1488) ...
struct DlVeloPre { double pars[DL_N_VPARS], one_b1L, f, sn_scale[3]; bool physical; };
DL_HD void dl_velocileptors_prelude(const DlObsDev& o, const double* v, double sigma8, double fsigma8, DlVeloPre& p) {
    p.one_b1L = 1.; p.f = 0.;
    p.physical = (o.mono_mode == 1 || o.mono_mode == 2);
    const bool rept = (o.mono_mode == 2 || o.mono_mode == 4);
    p.sn_scale[0] = p.sn_scale[1] = p.sn_scale[2] = 1.;
    if (p.physical) {
        p.f = fsigma8 / sigma8;
        double b1L = v[0] / sigma8 - 1., b2L = v[1] / (sigma8 * sigma8), bsL = v[2] / (sigma8 * sigma8), b3L = v[3] / (sigma8 * sigma8 * sigma8);
        p.one_b1L = 1. + b1L;
        if (rept) { p.pars[0] = 1. + b1L; p.pars[1] = 8. / 21. * b1L + b2L; p.pars[2] = bsL; p.pars[3] = b3L; }
        else { p.pars[0] = b1L; p.pars[1] = b2L; p.pars[2] = bsL; p.pars[3] = b3L; }
        p.pars[4] = p.one_b1L * p.one_b1L * v[4];
        p.pars[5] = p.f * p.one_b1L * (v[4] + v[5]);
        p.pars[6] = p.f * (p.f * v[5] + p.one_b1L * v[6]);
        p.pars[7] = p.f * p.f * v[6];
        p.sn_scale[0] = o.snd; p.sn_scale[1] = o.snd * o.fsat * (o.sigv * o.sigv); p.sn_scale[2] = o.snd * o.fsat * (o.sigv * o.sigv) * (o.sigv * o.sigv);
        for (int i = 0; i < 3; ++i) p.pars[8 + i] = v[8 + i] * p.sn_scale[i];
    } else {
        for (int c = 0; c < DL_N_VPARS; ++c) p.pars[c] = v[c];
    }
    if (rept) {   // co-evolution part, full_shape.py:1481-1485
        double b1 = p.pars[0];
        p.pars[2] = p.pars[2] - (2. / 7.) * (b1 - 1.);
        p.pars[3] = 3. * p.pars[3] + (b1 - 1.);
    }
}
// ... the 19 bias monomials (full_shape.py:1182-1186; r0[19] = 0: the padding of a row of 20) ...
DL_HD void dl_velocileptors_row0(const DlObsDev& o, const DlVeloPre& p, double* r0) {
    const double b1 = p.pars[0], b2 = p.pars[1], bs = p.pars[2], b3 = p.pars[3];
    r0[0] = 1.; r0[1] = b1; r0[2] = b1 * b1; r0[3] = b2; r0[4] = b1 * b2; r0[5] = b2 * b2; r0[6] = bs; r0[7] = b1 * bs; r0[8] = b2 * bs; r0[9] = bs * bs;
    r0[10] = b3; r0[11] = b1 * b3; r0[12] = p.pars[4]; r0[13] = p.pars[5]; r0[14] = p.pars[6]; r0[15] = p.pars[7];
    r0[16] = p.pars[8] / o.nd; r0[17] = p.pars[9] / o.nd; r0[18] = p.pars[10] / o.nd; r0[19] = 0.;
}
// ... and the non-zero entries of the derivative row of parameter c (4 .. 10): up to two (monomial, value) pairs, monomials ascending, -1: none
DL_HD void dl_velocileptors_drow(const DlObsDev& o, const DlVeloPre& p, int c, int nz[2], double dv[2]) {
    nz[0] = nz[1] = -1; dv[0] = dv[1] = 0.;
    if (p.physical) {
        if (c == 4) { nz[0] = 12; dv[0] = p.one_b1L * p.one_b1L; nz[1] = 13; dv[1] = p.f * p.one_b1L; }
        else if (c == 5) { nz[0] = 13; dv[0] = p.f * p.one_b1L; nz[1] = 14; dv[1] = p.f * p.f; }
        else if (c == 6) { nz[0] = 14; dv[0] = p.f * p.one_b1L; nz[1] = 15; dv[1] = p.f * p.f; }
        else if (c >= 8) { nz[0] = 16 + (c - 8); dv[0] = (c == 8 ? p.sn_scale[0] : c == 9 ? p.sn_scale[1] : p.sn_scale[2]) / o.nd; }
    } else {
        if (c < 8) { nz[0] = 12 + (c - 4); dv[0] = 1.; }
        else { nz[0] = 16 + (c - 8); dv[0] = 1. / o.nd; }
    }
}

// ``ms``: row stride of ``mono`` (entries DL_N_MONO .. ms - 1 of every row are set to zero); ``vpre``: the eleven 'pars' inputs already fetched (else read from ``th``)
// ``only_row`` >= 0: write that row alone (0: the monomials, r >= 1: the derivative row of slot r - 1)
DL_HD void dl_velocileptors_monomials(const DlObsDev& o, const double* th, double sigma8, double fsigma8, double* mono, int ms = DL_N_MONO, const double* vpre = nullptr,
                                      int only_row = -1) {
    double v[DL_N_VPARS];
    for (int c = 0; c < DL_N_VPARS; ++c) v[c] = vpre != nullptr ? vpre[c] : dl_get(o.vp_in[c], th);
    DlVeloPre p;
    dl_velocileptors_prelude(o, v, sigma8, fsigma8, p);
    if (only_row <= 0) {
        double r0[20];
        dl_velocileptors_row0(o, p, r0);
        for (int m = 0; m < DL_N_MONO; ++m) mono[m] = r0[m];
        for (int m = DL_N_MONO; m < ms; ++m) mono[m] = 0.;
    }
    for (int c = 4; c < DL_N_VPARS; ++c) {
        int slot = o.vp_slot[c];
        if (slot < 0 || (only_row >= 0 && only_row != 1 + slot)) continue;
        double* d = mono + (size_t)(1 + slot) * ms;
        for (int m = 0; m < ms; ++m) d[m] = 0.;
        int nz[2];
        double dv[2];
        dl_velocileptors_drow(o, p, c, nz, dv);
        if (nz[0] >= 0) d[nz[0]] = dv[0];
        if (nz[1] >= 0) d[nz[1]] = dv[1];
    }
}

// which monomials the derivative row of velocileptors parameter c (4 .. 10: alpha0 .. sn4) touches -- the support of the rows dl_velocileptors_monomials writes
// (at most two, ascending; -1: none).  The Gram epilogue of the feature GEMM (dl_feature_gemm.h) multiplies only these.
DL_HD void dl_velocileptors_row_support(const DlObsDev& o, int c, int nz[2]) {
    const bool physical = (o.mono_mode == 1 || o.mono_mode == 2);
    nz[0] = nz[1] = -1;
    if (physical) {
        if (c == 4) { nz[0] = 12; nz[1] = 13; }
        else if (c == 5) { nz[0] = 13; nz[1] = 14; }
        else if (c == 6) { nz[0] = 14; nz[1] = 15; }
        else if (c >= 8) nz[0] = 16 + (c - 8);
    } else {
        if (c < 8) nz[0] = 12 + (c - 4);
        else nz[0] = 16 + (c - 8);
    }
}

// Parallel regions: on the device every thread of the workgroup runs the region once and a barrier follows; in the CPU emulation
// (tests/csrc/emulate.cpp, host pass) the region is a loop over the DL_FS_THREADS threads.
#if defined(__HIP_DEVICE_COMPILE__)
#define DL_PAR_BEGIN { const int tid = threadIdx.x;
#define DL_PAR_END } __syncthreads();
#else
#define DL_PAR_BEGIN for (int tid = 0; tid < DL_FS_THREADS; ++tid) {
#define DL_PAR_END }
#endif

// engine `ie` -> its output vector in LDS (returned pointer: buf0 or buf1); scalar engines leave their value in out[0]
DL_HD double* dl_emu_engine(const DlObsDev& o, int ie, double* lds) {
    const DlObsDev::Engine& e = o.eng[ie];
    double* x = lds + DL_EM_X;
    double* cur = lds + DL_EM_BUF0;
    double* nxt = lds + DL_EM_BUF1;
    if (e.type == 0) {
        DL_PAR_BEGIN
            if (tid < o.n_x) cur[tid] = (x[tid] - e.xlo[tid]) * e.xinv[tid];        // conversion.py:75-77
        DL_PAR_END
        const double* w = e.weights;
        for (int layer = 0; layer < e.n_layers; ++layer) {
            const bool last = (layer == e.n_layers - 1);
            const bool activate = !(last && ie != 0);    // the table engine stops after its last HIDDEN layer (its final linear layer is folded on the host)
            DL_PAR_BEGIN
                dl_emu_layer(tid, e, layer, w, cur, nxt, activate);
            DL_PAR_END
            w += (size_t)e.widths[layer] * e.widths[layer + 1] + e.widths[layer + 1];
            double* t = cur; cur = nxt; nxt = t;
        }
        if (ie != 0) {
            DL_PAR_BEGIN
                if (tid == 0) cur[0] = cur[0] * e.yscale + e.ylo;                      // conversion.py:79 (inverse scaler)
            DL_PAR_END
        }
        return cur;
    }
    // Taylor: monomials prod_p (x_p - c_p)^powers[t, p] (emulators/__init__.py:471-507)
    DL_PAR_BEGIN
        for (int t = tid; t < e.n_terms; t += DL_FS_THREADS) {
            double mon = 1.;
            for (int p = 0; p < o.n_x; ++p) mon *= dl_ipow(x[p] - e.center[p], (int)e.powers[(size_t)t * o.n_x + p]);
            nxt[t] = mon;
        }
    DL_PAR_END
    if (ie != 0) {
        DL_PAR_BEGIN
            if (tid == 0) {
                double sum = 0.;
                for (int t = 0; t < e.n_terms; ++t) sum = fma(e.coef[t], nxt[t], sum);
                nxt[0] = sum;
            }
        DL_PAR_END
    }
    return nxt;
}

// Stacked table engine (emulators/conversion.py:44-98), one point, generic path (the theory vector itself; the batched MFMA form is dl_emu_stacked.h): group by group,
// the networks of the group one after the other (a thread per unit), then columns col + h nm + (m - mfirst) = amplitude_g basis_h mono_m.  Called after the scalar
// engines (lds[DL_EM_SCAL + 1, 2]) and the inputs (lds[DL_EM_X ..]) are in place.
DL_HD void dl_emu_point_stacked(const DlObsDev& o, const double* th, double* lds, double* row0, int64_t ld) {
    const DlObsDev::Engine& e = o.eng[0];
    double* x = lds + DL_EM_X;
    double* mono = lds + DL_EM_MONO;
    double* basis = mono + (size_t)(1 + o.n_var) * DL_N_MONO;        // [max_k + 8]
    double* xs = basis + o.stk.max_k + 8;                            // [DL_MAX_X] scaled inputs (conversion.py:75-77), the same for every network
    const int H = e.widths[e.n_layers];
    DL_PAR_BEGIN
        if (tid < o.n_x) xs[tid] = (x[tid] - e.xlo[tid]) * e.xinv[tid];
        if (tid == 0) {
            const double sigma8 = o.eng[1].type >= 0 ? lds[DL_EM_SCAL + 1] : o.eng[1].cst, fsigma8 = o.eng[2].type >= 0 ? lds[DL_EM_SCAL + 2] : o.eng[2].cst;
            if (o.mono_mode == 0) mono[0] = 1.;
            else dl_velocileptors_monomials(o, th, sigma8, fsigma8, mono);
        }
    DL_PAR_END
    int tb_prev = -1, te_prev = -1;
    for (int gi = 0; gi < o.stk.n_groups; ++gi) {
        const double* rec = o.stk.table + (size_t)gi * DL_STK_REC;
        const int tb = (int)rec[0], te = (int)rec[1], m0 = (int)rec[2], m1 = (int)rec[3], col = (int)rec[4], nm = (int)rec[5], mo = (int)rec[6];
        const int K = (te - tb) * H + 1;
        if (tb != tb_prev || te != te_prev) {
            for (int t = tb; t < te; ++t) {
                double* cur = lds + DL_EM_BUF0;
                double* nxt = lds + DL_EM_BUF1;
                const double* w = e.weights + (size_t)t * o.stk.trunk_doubles;
                for (int layer = 0; layer < e.n_layers; ++layer) {
                    const double* in = layer == 0 ? xs : cur;
                    DL_PAR_BEGIN
                        dl_emu_layer(tid, e, layer, w, in, nxt, true);
                    DL_PAR_END
                    w += (size_t)e.widths[layer] * e.widths[layer + 1] + e.widths[layer + 1];
                    double* sw = cur; cur = nxt; nxt = sw;
                }
                DL_PAR_BEGIN
                    if (tid < H) basis[(size_t)(t - tb) * H + tid] = cur[tid];
                DL_PAR_END
            }
            DL_PAR_BEGIN
                if (tid == 0) basis[K - 1] = 1.;     // the constant parts of the folded final layers of the group, summed on the host
            DL_PAR_END
            tb_prev = tb; te_prev = te;
        }
        double la = o.stk.scale[(size_t)gi * (o.n_x + 1) + o.n_x];
        for (int j = 0; j < o.n_x; ++j) la = fma(o.stk.scale[(size_t)gi * (o.n_x + 1) + j], x[j], la);
        const double amp = la == 0. ? 1. : exp(la);
        DL_PAR_BEGIN
            const int cnt = m1 - m0;
            for (int idx = tid; idx < K * cnt; idx += DL_FS_THREADS) {
                const int h = idx / cnt, mi = idx - h * cnt;
                const double bh = amp * basis[h];
                const size_t c = (size_t)col + (size_t)h * nm + mo + mi;
                row0[c] = bh * mono[m0 + mi];
                for (int v = 0; v < o.n_var; ++v) row0[(size_t)(1 + v) * ld + c] = bh * mono[(size_t)(1 + v) * DL_N_MONO + m0 + mi];
            }
        DL_PAR_END
    }
    DL_PAR_BEGIN
        for (int c = tid; c < o.n_pass; c += DL_FS_THREADS) { const int col = (int)o.pass_tab[2 * c]; row0[o.n_in + c] = col >= 0 ? th[col] : o.pass_tab[2 * c + 1]; }
    DL_PAR_END
}

// features of one point: row0[(h, m)] = basis_h mono_m, derivative rows (1 + slot)[(h, m)] = basis_h dmono_slot,m, pass-through columns.
// feat_rec != nullptr (feature path, dl_feature_gemm.h): only the factors are written, basis [nb_pad] then the monomial rows [(1 + n_var)][20].
DL_HD void dl_emu_point(const DlObsDev& o, const double* th, double* lds, double* row0, int64_t ld, double* feat_rec = nullptr) {
    DL_PAR_BEGIN
        if (tid < o.n_x) lds[DL_EM_X + tid] = dl_get(o.x_in[tid], th);
    DL_PAR_END
    double sigma8 = o.eng[1].cst, fsigma8 = o.eng[2].cst;
    for (int ie = 1; ie <= 2; ++ie) {
        if (o.eng[ie].type < 0) continue;
        double* out = dl_emu_engine(o, ie, lds);
        DL_PAR_BEGIN
            if (tid == 0) lds[DL_EM_SCAL + ie] = out[0];
        DL_PAR_END
    }
    if (o.eng[0].type == 2) { dl_emu_point_stacked(o, th, lds, row0, ld); return; }
    double* basis = dl_emu_engine(o, 0, lds);
    double* mono = lds + DL_EM_MONO;
    DL_PAR_BEGIN
        if (tid == 0) {
            if (o.eng[1].type >= 0) sigma8 = lds[DL_EM_SCAL + 1];
            if (o.eng[2].type >= 0) fsigma8 = lds[DL_EM_SCAL + 2];
            if (o.mono_mode == 0) mono[0] = 1.;
            else dl_velocileptors_monomials(o, th, sigma8, fsigma8, mono);
            if (o.eng[0].type == 0) basis[o.n_basis - 1] = 1.;      // bias row of the folded final layer
        }
    DL_PAR_END
    if (feat_rec != nullptr) {
        DL_PAR_BEGIN
            for (int h = tid; h < o.nb_pad; h += DL_FS_THREADS) feat_rec[h] = h < o.n_basis ? basis[h] : 0.;
            for (int idx = tid; idx < (1 + o.n_var) * 20; idx += DL_FS_THREADS) {
                int r = idx / 20, m = idx - r * 20;
                feat_rec[o.nb_pad + idx] = m < DL_N_MONO ? mono[(size_t)r * DL_N_MONO + m] : 0.;
            }
        DL_PAR_END
        return;
    }
    DL_PAR_BEGIN
        const int nm = o.n_mono;
        for (int idx = tid; idx < o.n_in; idx += DL_FS_THREADS) {
            int h = idx / nm, m = idx - h * nm;
            double bh = basis[h];
            row0[idx] = bh * mono[m];
            for (int v = 0; v < o.n_var; ++v) row0[(size_t)(1 + v) * ld + idx] = bh * mono[(size_t)(1 + v) * DL_N_MONO + m];
        }
        for (int c = tid; c < o.n_pass; c += DL_FS_THREADS) { const int col = (int)o.pass_tab[2 * c]; row0[o.n_in + c] = col >= 0 ? th[col] : o.pass_tab[2 * c + 1]; }
    DL_PAR_END
}
