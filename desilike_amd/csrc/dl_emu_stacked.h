// Stacked table engine (SURVEY 8a row a12 as the reference ships it: emulators/conversion.py:44-98) -- emulator forward pass and feature GEMM of 16 parameter points per
// workgroup in one launch.
//
//   The emulated perturbation-theory node holds FOUR engines ('11', 'loop', 'ct', 'st': the bias monomials 0-2, 3-11, 12-15, 16-18 of full_shape.py:1182-1186), each one
//   network per (z, ell) with the same hidden layers (kernels stacked [n_z, n_ell, in, out], conversion.py:58-66), outputs rescaled by exp(logA) 1e-10 (squared for 'loop',
//   conversion.py:88-92), redshifts selected / blended by full_shape.py:1416-1443.  After the last hidden layers everything is linear with constant coefficients times one
//   amplitude per engine, so the host folds final layers x y-scalers x assembly x redshift blend x k-interpolation x window x L^T into ONE operator per group g of networks:
//       row_r[j] = sum_g amp_g(x) sum_{m in g} mono_r[m] sum_{h < K_g} G_g[(m, j)][h] basis_g[h],     basis_g = (last hidden layers of the group's networks ..., 1)
//   -- the separable form of dl_feature_gemm.h with one A operand (K_g = n_networks_g H + 1 basis functions) and one monomial range per group.
//
//   Workgroup = 16 points x 128 output columns, 512 threads.  Per group: (i) NETWORKS, layer by layer: every dense layer of every network of the group is
//   [16 points x n_in] . [n_in x n_out] by v_mfma_f64_16x16x4_f64, in tasks of one output tile (16 units) dealt evenly to the eight waves; activations in LDS, updated
//   in place (a wave keeps the outputs of its tasks in registers across the barrier that ends the layer's reads); weights stream from L2 in fragment order (a request =
//   the sixteen k-steps of the NEXT task, issued before the MFMAs of the current one; the next layer's first task before the activations); one exponential and one
//   reciprocal per activation; the last hidden layer lands in the group's basis record.  (ii) FEATURE GEMM: wave w = column block w;
//   the operand streams from L2 in fragment order [column block][group][k / 8][m][lane][2], three to seven steps in flight (at most five monomials per pass: a step
//   is only 2 x 5 MFMAs long); the epilogue contracts the accumulators with the amplitude-scaled monomial rows of the lane's four points into the carried output rows
//   (registers; at most 8 rows: residual + the seven alpha* / sn* that can be solved).  (iii) FINALIZE IN THE TAIL (DlStkTail; one observable, N_pad = 128): the carried
//   rows go to LDS, Gram matrices by MFMA, one lane per point solves -- one launch per step; otherwise the residual rows are written for the general finalize kernels.
#pragma once
#include "dl_fullshape.h"
#include "dl_feature_gemm.h"
#include "dl_marg_solve.h"

typedef double dl_stk_double4 __attribute__((ext_vector_type(4)));

#define DL_STK_PTS 16
#define DL_STK_ROWS 8        // rows carried per point: 1 + the solvable alpha0 alpha2 alpha4 alpha6 sn0 sn2 sn4 (full_shape.py:1226)
#define DL_STK_TMAX 8        // output tiles of a layer (widths <= 128)
#define DL_STK_STATIC_LDS 512   // bytes of static LDS the kernels declare beside the dynamic block (lp_lds, nan_lds, counters)

static inline __host__ __device__ int dl_stk_tld(const DlObsDev& o) {          // row stride of a wave's activation buffer: widest layer, multiple of 4, + 2
    int w = 4;
    for (int ie = 0; ie < 3; ++ie) {
        const DlObsDev::Engine& e = o.eng[ie];
        if (e.type == 0 || e.type == 2) for (int l = 0; l <= e.n_layers; ++l) if (e.widths[l] > w) w = e.widths[l];
    }
    return (w + 3) / 4 * 4 + 2;
}
static inline __host__ __device__ int dl_stk_bld(const DlObsDev& o) { return dl_fg_lds_stride((o.stk.max_k + 7) / 8 * 8); }   // row stride of the basis record
// LDS (doubles): x [16][16] | xs [3 engines][16][18] | amp [16][8] | scal [16][4] | vpv [16][12] | mono [16][8][20] | basis [16][bld] | 8 wave buffers [16][tld]
static inline __host__ __device__ size_t dl_stk_shared_doubles(const DlObsDev& o) {
    return (size_t)DL_STK_PTS * (DL_MAX_X + 3 * (DL_MAX_X + 2) + DL_STK_MAX_GROUPS + 4 + 12 + DL_STK_ROWS * DL_FG_MONO_LD + dl_stk_bld(o) + 8 * dl_stk_tld(o));
}
// feature path of the stacked engine: scalar engines constant or MLP, rows within the carried set, every layer within the tile budget
static inline bool dl_stk_feature_ok(const DlObsDev& o) {
    if (o.theory != 3 || o.eng[0].type != 2 || o.n_pass != 0 || 1 + o.n_var > DL_STK_ROWS) return false;
    for (int ie = 1; ie < 3; ++ie) {
        if (o.eng[ie].type == 1) return false;
        if (o.eng[ie].type == 0) for (int l = 0; l <= o.eng[ie].n_layers; ++l) if (o.eng[ie].widths[l] > 128) return false;
    }
    return dl_stk_shared_doubles(o) * sizeof(double) + DL_STK_STATIC_LDS <= 160 * 1024;
}

// The finalize in the kernel's tail (one observable, N_pad = 128, 1 + n_s <= 8 rows of X that fit the LDS the networks and the basis record no longer need): the rows
// X = [residual + bias; derivative rows + tconst_s] of the workgroup's 16 points go from the carried registers to LDS, each wave forms G = X X^T of two points by MFMA
// (dl_stk_gram_phase), lanes 0-15 of wave 0 solve a point each (dl_marg_solve.h) while wave 1 sums the priors -- as in the tail of dl_emulated_feature_gram_kernel.  Against
// rows through memory + dl_finalize_marg_kernel: 25 MB written and read back per 4096 points and a 20 us launch less.
struct DlStkTail {
    int enabled, xr, n_const;          // xr: rows of X (1 + n_s)
    int row_of[DL_STK_ROWS];           // X row of device row r
    const double* cst[DL_STK_ROWS];    // constant part of device row r: bias, or tconst of its solved parameter ([128] each)
    int const_row[DL_MAX_SOLVED];      // X rows that are constants only ...
    const double* const_ptr[DL_MAX_SOLVED];
    int post_mode, pad;
    const double* priors;
    double *loglike, *logprior;
    int32_t* status;
    double *solved, *hessian;
    DlMargDev mg;
};
static inline __host__ __device__ bool dl_stk_tail_fits(const DlObsDev& o, int xr) {
    return xr <= 8 && (size_t)DL_STK_PTS * xr * DL_FG_XLD <= (size_t)DL_STK_PTS * dl_stk_bld(o) + (size_t)8 * DL_STK_PTS * dl_stk_tld(o);
}

#if defined(__HIPCC__)
// Activations of the batched networks: one exponential and one reciprocal (v_rcp_f64 + two Newton steps) each -- silu v / (1 + e^-v), tanh 1 - 2 / (1 + e^2v) (absolute error
// ~1e-16; the library's tanh is 165 vector instructions, this is ~50, and on this chip every fp64 vector instruction beside the MFMAs is added to their time)
__device__ __forceinline__ double dl_stk_rcp(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.), r, r);
    return fma(fma(-v, r, 1.), r, r);
}
// e^x, ~2 ulp, 21 vector instructions (the library's: 42): x = n ln 2 + r, |r| <= ln 2 / 2 (two-term Cody-Waite), Taylor polynomial of degree 13 (first neglected term
// r^14 / 14! < 5e-18), v_ldexp_f64; the argument is clamped to the range of finite results (the activations only need 1 / (1 + e^x))
__device__ __forceinline__ double dl_stk_exp(double x) {
    x = fmin(fmax(x, -708.), 709.);
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1. / 6227020800.;
    p = fma(p, r, 1. / 479001600.); p = fma(p, r, 1. / 39916800.); p = fma(p, r, 1. / 3628800.); p = fma(p, r, 1. / 362880.); p = fma(p, r, 1. / 40320.);
    p = fma(p, r, 1. / 5040.); p = fma(p, r, 1. / 720.); p = fma(p, r, 1. / 120.); p = fma(p, r, 1. / 24.); p = fma(p, r, 1. / 6.); p = fma(p, r, 0.5);
    p = fma(p, r, 1.); p = fma(p, r, 1.);
    return ldexp(p, (int)n);
}
__device__ __forceinline__ double dl_stk_act(int act, double v) {
    if (act == 0) return v * dl_stk_rcp(1. + dl_stk_exp(-v));       // conversion.py:29
    if (act == 1) return v > 0. ? v : 0.;                           // conversion.py:31
    const double t = 1. - 2. * dl_stk_rcp(1. + dl_stk_exp(2. * v)); // conversion.py:33 (large |v|: +-1)
    return v != v ? v : t;                                          // (the clamp inside dl_stk_exp drops a NaN: hand it on as dl_activation does)
}

// The networks of one group, LAYER BY LAYER on all eight waves: a task = one output tile (16 units) of one network; the n_net * tiles tasks of a layer are dealt to the waves
// in contiguous runs (a wave's tasks mostly share a network, i.e. an A operand), so the four SIMDs carry the same load whatever the number of networks (one network per wave:
// six networks on eight waves left two SIMDs with twice the work of the others, and the phase took their time).  Activations of network j live in bufs + j * 16 * tld
// (row stride tld), updated in place: every wave keeps the outputs of its tasks in registers across the barrier that ends the reads of the layer.  `wf`: the weights in
// fragment order (DlObsDev::Stack::wfrag, network `first` + j at wf + j * frag_doubles): a B-operand load is base + lane + immediate, no predicates; the weights of the next
// task (of the next layer's first task) are requested before the MFMAs (the activations) of the current one.  The last hidden layer goes to the basis record:
// dst[point * dst_ld + j * H + unit].
template <int TMAX>
__device__ __forceinline__ void dl_stk_networks(const int32_t* widths, int n_layers, int act, const double* __restrict__ wf, int frag_doubles, int n_net, const double* in0, int ld0,
                                                double* bufs, int tld, double* dst, int dst_ld, int wave, int lane, unsigned long long* lst = nullptr) {
    const int col = lane & 15, g = lane >> 4;
    wf += lane;
    double bw[16];
    int lslot = 0;   // diagnostics (DL_STK_STAMPS): wave 0 stamps layers 1-3 of the first group: start, MFMAs issued, reads done (barrier), activations written, barrier
#define DL_STK_LSTAMP if (lst != nullptr && layer >= 1 && layer <= 3 && threadIdx.x == 0 && lslot < 15) lst[lslot] = __builtin_amdgcn_s_memtime(); if (layer >= 1 && layer <= 3) ++lslot;
    bool have = false;                      // bw holds the weights of this wave's first task of the coming layer
    size_t loff = 0;                        // offset of the layer in a network's fragment-ordered weights
    for (int layer = 0; layer < n_layers; ++layer) {
        const int nin = widths[layer], nout = widths[layer + 1];
        const int ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16;
        const bool last = layer == n_layers - 1;
        const int total = n_net * tiles, per = (total + 7) / 8;
        const int t_begin = wave * per < total ? wave * per : total, t_end = t_begin + per < total ? t_begin + per : total;
        const size_t lnext = loff + (size_t)tiles * ksteps * 64 + 16 * tiles;
        // this wave's first task of the next layer (the same split: tiles may differ)
        const int ntiles = last ? 1 : (widths[layer + 2] + 15) / 16, ntotal = n_net * ntiles, nper = (ntotal + 7) / 8;
        const int nt_begin = wave * nper < ntotal ? wave * nper : ntotal;
        const bool next16 = !last && (nout + 3) / 4 == 16 && nt_begin < ntotal;
 DL_STK_LSTAMP
        dl_stk_double4 res[TMAX];
        if (ksteps == 16) {
            double bwn[16];
            if (!have && t_begin < t_end) {
                const double* wt = wf + (size_t)(t_begin / tiles) * frag_doubles + loff + (size_t)(t_begin % tiles) * 1024;
#pragma unroll
                for (int u = 0; u < 16; ++u) bw[u] = wt[u * 64];
            }
#pragma unroll
            for (int i = 0; i < TMAX; ++i) {
                const int task = t_begin + i;
                if (task >= t_end) break;
                const int jn = task / tiles;
                const double* ap = (layer == 0 ? in0 + col * ld0 : bufs + (size_t)jn * DL_STK_PTS * tld + col * tld) + g;
                double av[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) av[u] = ap[4 * u];
                if (task + 1 < t_end) {
                    const double* wt = wf + (size_t)((task + 1) / tiles) * frag_doubles + loff + (size_t)((task + 1) % tiles) * 1024;
#pragma unroll
                    for (int u = 0; u < 16; ++u) bwn[u] = wt[u * 64];
                } else if (next16) {
                    const double* wt = wf + (size_t)(nt_begin / ntiles) * frag_doubles + lnext + (size_t)(nt_begin % ntiles) * 1024;
#pragma unroll
                    for (int u = 0; u < 16; ++u) bwn[u] = wt[u * 64];
                }
                dl_stk_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};   // two chains: a dependent MFMA waits for its predecessor
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bw[u + 1], acc2, 0, 0, 0);
                }
                res[i] = acc + acc2;
#pragma unroll
                for (int u = 0; u < 16; ++u) bw[u] = bwn[u];
            }
            if (t_begin >= t_end && next16) {     // a wave without a task in this layer that has one in the next: nothing above requested its weights
                const double* wt = wf + (size_t)(nt_begin / ntiles) * frag_doubles + lnext + (size_t)(nt_begin % ntiles) * 1024;
#pragma unroll
                for (int u = 0; u < 16; ++u) bw[u] = wt[u * 64];
            }
        } else {
#pragma unroll
            for (int i = 0; i < TMAX; ++i) {
                const int task = t_begin + i;
                if (task >= t_end) break;
                const int jn = task / tiles, t = task - jn * tiles;
                const double* ap = (layer == 0 ? in0 + col * ld0 : bufs + (size_t)jn * DL_STK_PTS * tld + col * tld) + g;
                const double* wt = wf + (size_t)jn * frag_doubles + loff + (size_t)t * ksteps * 64;
                dl_stk_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};
                for (int u = 0; u < ksteps; u += 2) {
                    const bool two = u + 1 < ksteps;
                    const double b0 = wt[u * 64], b1 = wt[(two ? u + 1 : u) * 64];
                    const double a0 = ap[4 * u], a1 = ap[4 * (two ? u + 1 : u)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                    if (two) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc2, 0, 0, 0);
                }
                res[i] = acc + acc2;
            }
            if (next16) {
                const double* wt = wf + (size_t)(nt_begin / ntiles) * frag_doubles + lnext + (size_t)(nt_begin % ntiles) * 1024;
#pragma unroll
                for (int u = 0; u < 16; ++u) bw[u] = wt[u * 64];
            }
        }
        have = next16;
        DL_STK_LSTAMP
        // every wave has read what it needs of this layer's inputs (LDS reads of this wave are complete: lgkmcnt; the weight requests of the next layer stay in flight)
        if (layer > 0) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
        else asm volatile("" ::: "memory");
        DL_STK_LSTAMP
#pragma unroll
        for (int i = 0; i < TMAX; ++i) {
            const int task = t_begin + i;
            if (task >= t_end) break;
            const int jn = task / tiles, t = task - jn * tiles;
            const int oc = 16 * t + col;
            const double b = (wf - lane)[(size_t)jn * frag_doubles + loff + (size_t)tiles * ksteps * 64 + oc];
            double vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = res[i][r] + b;          // accumulator register r = out[point g + 4 r][oc]
            if (act == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = dl_stk_act(0, vv[r]);
            } else if (act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = dl_stk_act(1, vv[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = dl_stk_act(2, vv[r]);
            }
            // units beyond the layer (zero weights and bias) are written too when they pad the next layer's k-steps: act(0) of silu / relu / tanh is 0
            const int nout4 = (nout + 3) & ~3;
            double* hb = bufs + (size_t)jn * DL_STK_PTS * tld;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!last) { if (oc < nout4) hb[(g + 4 * r) * tld + oc] = vv[r]; }
                else if (oc < nout) dst[(g + 4 * r) * dst_ld + (size_t)jn * nout + oc] = vv[r];
            }
        }
        DL_STK_LSTAMP
        if (!last) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");    // the layer's outputs are in place
        DL_STK_LSTAMP
        loff = lnext;
    }
}

// feature GEMM of one group: CNT monomials, nq operand steps of 8 basis functions, D operand buffers (D - 1 steps in flight: a step is only 2 CNT MFMAs long, and the
// operand comes from L2); then the contraction with the (amplitude-scaled) monomial rows into the carried rows
template <int CNT, int RMAX, int D>
__device__ __forceinline__ void dl_stk_group_gemm(const double* arow, const dl_fg_double2* __restrict__ gw, int nq, const double* mono, int R, int g, double (&outv)[4][RMAX]) {
    dl_fg_double4 acc[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] = (dl_fg_double4){0., 0., 0., 0.};
    dl_fg_double2 b[D][CNT];
#define DL_STK_LOAD(bb, qq) { const int q_ = (qq) < nq ? (qq) : nq - 1; _Pragma("unroll") for (int i = 0; i < CNT; ++i) bb[i] = gw[(size_t)(q_ * CNT + i) * 64]; }
#define DL_STK_MUL(bb, qq) { const dl_fg_double2 a_ = *reinterpret_cast<const dl_fg_double2*>(arow + 8 * (qq)); \
        _Pragma("unroll") for (int i = 0; i < CNT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_.x, bb[i].x, acc[i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < CNT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_.y, bb[i].y, acc[i], 0, 0, 0); }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) DL_STK_LOAD(b[d], d)
    int q = 0;
    for (; q + D <= nq; q += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) { DL_STK_LOAD(b[(d + D - 1) % D], q + d + D - 1) DL_STK_MUL(b[d], q + d) }
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (q + d < nq) { DL_STK_MUL(b[d], q + d) }
#undef DL_STK_LOAD
#undef DL_STK_MUL
    // accumulator register rr of lane (col, g) = U[point g + 4 rr][m][column]; rows of that point += sum_m mono_row[m] U[m]
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const double* mp = mono + (size_t)(g + 4 * rr) * DL_STK_ROWS * DL_FG_MONO_LD;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            if (u < R) {
                double v = outv[rr][u];
#pragma unroll
                for (int i = 0; i < CNT; ++i) v = fma(mp[u * DL_FG_MONO_LD + i], acc[i][rr], v);
                outv[rr][u] = v;
            }
        }
    }
}

// G = X X^T of the workgroup's 16 points from their rows in LDS (xr <= 8 rows per point): the two-points-per-tile form of dl_fg_gram_phase (rows 0-7 of the 16-row MFMA tile:
// point 2 wave, rows 8-15: point 2 wave + 1; the 32 operand values of a lane requested together); the 8 x 8 block of a point lands over its own X rows, at x + pt xr DL_FG_XLD + 8 i + j
__device__ __forceinline__ void dl_stk_gram_phase(double* x, int xr, int wave, int lane, int g) {
    const int j = lane & 15, pp = j >> 3, row = j & 7;
    const bool live = row < xr;
    const double* xp = x + ((size_t)(2 * wave + pp) * xr + (live ? row : 0)) * DL_FG_XLD + g;
    double xv[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) xv[k] = xp[4 * k];
    dl_fg_double4 acc0 = {0., 0., 0., 0.};
#pragma unroll
    for (int k = 0; k < 32; ++k) { const double x0 = live ? xv[k] : 0.; acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc0, 0, 0, 0); }
    double* gl = x + (size_t)(2 * wave + pp) * xr * DL_FG_XLD;     // (every operand read of this wave precedes these writes; no other wave reads these rows)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = g + 4 * r;
        if ((i >> 3) == pp) gl[(i & 7) * 8 + row] = acc0[r];
    }
}

template <int NS>
__device__ __forceinline__ void dl_stk_solve_point(const double* gl, const DlStkTail& tl, int64_t b, double& ll, double& lps, bool& ok) {
    ll = dl_marg_solve_lane<NS>([&](int i, int j) { return gl[i * 8 + j]; }, tl.mg, tl.solved ? tl.solved + (size_t)b * NS : nullptr,
                                tl.hessian ? tl.hessian + (size_t)b * NS * NS : nullptr, lps, ok);
}

// LDS of a workgroup: x [16][16] | xs [3 engines][16][18] | amp [16][8] | scal [16][4] | vpv [16][12] | mono [16][8][20] | work (basis records, activation buffers, X of the tail)
struct DlStkLds { double *x, *xs, *amp, *scal, *vpv, *mono, *work; };
#define DL_STK_XLD (DL_MAX_X + 2)
__device__ __forceinline__ DlStkLds dl_stk_carve(double* lds) {
    DlStkLds s;
    s.x = lds;                                                   // [16][DL_MAX_X] the emulator inputs
    s.xs = s.x + DL_STK_PTS * DL_MAX_X;                          // [3][16][XLD] scaled inputs of the table networks (0) and of the scalar engines (1, 2), zero-padded
    s.amp = s.xs + 3 * DL_STK_PTS * DL_STK_XLD;                  // [16][8] amplitude of every group
    s.scal = s.amp + DL_STK_PTS * DL_STK_MAX_GROUPS;             // [16][4]: sigma8 (1), fsigma8 (2)
    s.vpv = s.scal + DL_STK_PTS * 4;                             // [16][12] velocileptors 'pars' inputs
    s.mono = s.vpv + DL_STK_PTS * 12;                            // [16][DL_STK_ROWS][20] monomial rows, scaled by the amplitude of their group
    s.work = s.mono + DL_STK_PTS * DL_STK_ROWS * DL_FG_MONO_LD;
    return s;
}

// Inputs, scalar engines, amplitudes and the amplitude-scaled monomial rows of the workgroup's 16 points (512 threads; `scratch`: 2 x 16 x tld doubles; the monomial rows are
// NOT yet published by a barrier on return).  st: DL_STK_STAMPS slots 1 (inputs) and 2 (monomial rows) of this workgroup, or null
// th_early / th_val (dl_emulated_stacked_gemm_kernel, n_params <= 32): thread t has ALREADY requested theta[point t / 32][column t % 32] -- before its first access to the descriptor
// `o`, which is a round trip to the kernel-argument segment of its own: the rows go through LDS (the area of the monomial rows, written later) and the inputs are read from there,
// one global round trip at entry instead of two in a row (the entry of dl_emulated_feature_gram_kernel)
// beside(): work of the six waves that form no monomial rows, run beside them (dl_emulated_stacked_gemm_kernel: the requests of the first basis records)
struct DlStkNoop { __device__ __forceinline__ void operator()() const {} };
template <class F = DlStkNoop>
__device__ __forceinline__ void dl_stk_prologue(const DlObsDev& o, const double* __restrict__ theta, int n_params, int64_t B, int64_t p0, int tid, double* lds_base, double* scratch,
                                                int tld, int R, unsigned long long* st, bool th_early = false, double th_val = 0., F&& beside = F()) {
    constexpr int XLD = DL_STK_XLD;
    const DlStkLds s = dl_stk_carve(lds_base);
    double *x = s.x, *xs = s.xs, *amp = s.amp, *scal = s.scal, *vpv = s.vpv, *mono = s.mono;
    // ---- inputs ----
    if (th_early) {
        // descriptors first (lane-dependent fields of `o`: vector loads from the kernel-argument segment), then the theta rows into LDS, a barrier, the inputs from there
        const int pt = tid / XLD, i = tid - pt * XLD;
        const bool xlive = tid < DL_STK_PTS * XLD, xreal = xlive && i < o.n_x;
        const DlInput xin = o.x_in[xreal ? i : 0];
        double xlo[3], xinv[3];
#pragma unroll
        for (int ie = 0; ie < 3; ++ie) {
            const DlObsDev::Engine& en = o.eng[ie];
            const bool mlp = en.type == 0 || en.type == 2;
            xlo[ie] = mlp ? en.xlo[xreal ? i : 0] : 0.;
            xinv[ie] = mlp ? en.xinv[xreal ? i : 0] : 0.;
        }
        const int vpt = tid / DL_N_VPARS, vc = tid - vpt * DL_N_VPARS;
        const bool vlive = tid < DL_STK_PTS * DL_N_VPARS;
        const DlInput vin = o.vp_in[vlive ? vc : 0];
        double* trow = mono;                        // [16][32] (the monomial rows are written after the barrier below)
        if (st != nullptr && tid == 0) st[22] = __builtin_amdgcn_s_memtime();      // (slots 20 - 23: inside the entry of dl_emulated_stacked_gemm_kernel)
        trow[tid] = th_val;
        __syncthreads();
        if (st != nullptr && tid == 0) st[23] = __builtin_amdgcn_s_memtime();
        if (xlive) {
            const double tv = trow[pt * 32 + (xin.col >= 0 ? xin.col : 0)];
            const double v = xreal ? (xin.col >= 0 ? tv : xin.value) : 0.;
            if (xreal) x[pt * DL_MAX_X + i] = v;
#pragma unroll
            for (int ie = 0; ie < 3; ++ie) {
                const DlObsDev::Engine& en = o.eng[ie];
                if (en.type == 0 || en.type == 2) xs[(ie * DL_STK_PTS + pt) * XLD + i] = xreal ? (v - xlo[ie]) * xinv[ie] : 0.;   // conversion.py:75-77
            }
        }
        if (vlive) {
            const double tv = trow[vpt * 32 + (vin.col >= 0 ? vin.col : 0)];
            vpv[vpt * 12 + vc] = vin.col >= 0 ? tv : vin.value;
        }
    } else {
    for (int idx = tid; idx < DL_STK_PTS * XLD; idx += 512) {
        const int pt = idx / XLD, i = idx - pt * XLD;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        double v = 0.;
        if (i < o.n_x) { v = dl_get(o.x_in[i], theta + (size_t)b * n_params); x[pt * DL_MAX_X + i] = v; }
#pragma unroll
        for (int ie = 0; ie < 3; ++ie) {
            const DlObsDev::Engine& en = o.eng[ie];
            if (en.type == 0 || en.type == 2) xs[(ie * DL_STK_PTS + pt) * XLD + i] = i < o.n_x ? (v - en.xlo[i]) * en.xinv[i] : 0.;   // conversion.py:75-77
        }
    }
    for (int idx = tid; idx < DL_STK_PTS * DL_N_VPARS; idx += 512) {
        const int pt = idx / DL_N_VPARS, c = idx - pt * DL_N_VPARS;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        vpv[pt * 12 + c] = dl_get(o.vp_in[c], theta + (size_t)b * n_params);
    }
    }
    __syncthreads();
    if (st != nullptr && tid == 0) st[1] = __builtin_amdgcn_s_memtime();
    // ---- scalar engines (sigma8, fsigma8: the physical prior basis; small networks): a thread per (point, unit), layer by layer, in the still unused work area;
    //      the amplitudes of the groups ----
    for (int ie = 1; ie < 3; ++ie) {
        const DlObsDev::Engine& en = o.eng[ie];
        if (en.type != 0) continue;
        double* cur = scratch;
        double* nxt = scratch + DL_STK_PTS * tld;
        const double* w = en.weights;
        for (int layer = 0; layer < en.n_layers; ++layer) {
            const int nin = en.widths[layer], nout = en.widths[layer + 1];
            const bool last = layer == en.n_layers - 1;
            const double* src = layer == 0 ? xs + (size_t)ie * DL_STK_PTS * XLD : cur;
            const int sld = layer == 0 ? XLD : tld;
            for (int idx = tid; idx < DL_STK_PTS * nout; idx += 512) {
                const int pt = idx / nout, j = idx - pt * nout;
                double acc0 = w[(size_t)nin * nout + j], acc1 = 0.;       // (the summation order of dl_emu_layer)
                int i = 0;
                for (; i + 2 <= nin; i += 2) { acc0 = fma(src[pt * sld + i], w[(size_t)i * nout + j], acc0); acc1 = fma(src[pt * sld + i + 1], w[(size_t)(i + 1) * nout + j], acc1); }
                if (i < nin) acc0 = fma(src[pt * sld + i], w[(size_t)i * nout + j], acc0);
                const double v = acc0 + acc1;
                if (last) { if (j == 0) scal[pt * 4 + ie] = v * en.yscale + en.ylo; }      // inverse scaler, conversion.py:79
                else nxt[pt * tld + j] = dl_activation(en.act, v);
            }
            __syncthreads();
            w += (size_t)nin * nout + nout;
            double* sw = cur; cur = nxt; nxt = sw;
        }
    }
    for (int idx = tid; idx < DL_STK_PTS * o.stk.n_groups; idx += 512) {
        const int pt = idx / o.stk.n_groups, gi = idx - pt * o.stk.n_groups;
        const double* sc = o.stk.scale + (size_t)gi * (o.n_x + 1);
        double la = sc[o.n_x];
        for (int j = 0; j < o.n_x; ++j) la = fma(sc[j], x[pt * DL_MAX_X + j], la);
        amp[pt * DL_STK_MAX_GROUPS + gi] = la == 0. ? 1. : exp(la);
    }
    // the group of every monomial (-1: none), by the last threads beside the amplitudes: the rows below are then scaled by twenty independent multiplications per lane -- walking
    // the group table inside every lane (its records one after the other, a read-multiply-write chain through LDS per monomial) took 45 of the 140 hundred cycles of this phase
    __shared__ int group_of[DL_FG_MONO_LD];
    if (tid >= 512 - DL_FG_MONO_LD) {
        const int m = tid - (512 - DL_FG_MONO_LD);
        int gm = -1;
        for (int gi = 0; gi < o.stk.n_groups; ++gi) {
            const double* rec = o.stk.table + (size_t)gi * DL_STK_REC;
            if (m >= (int)rec[2] && m < (int)rec[3]) gm = gi;
        }
        group_of[m] = gm;
    }
    __syncthreads();
    // ---- monomial rows: one lane per (point, row), then scaled group by group (a monomial belongs to one group; monomials of no group feed nothing) ----
    if (tid >= DL_STK_PTS * DL_STK_ROWS) beside();
    else {
        const int pt = tid & 15, r = tid >> 4;
        if (r < R) {
            const double sigma8 = o.eng[1].type >= 0 ? scal[pt * 4 + 1] : o.eng[1].cst;
            const double fsigma8 = o.eng[2].type >= 0 ? scal[pt * 4 + 2] : o.eng[2].cst;
            double* row = mono + ((size_t)pt * DL_STK_ROWS + r) * DL_FG_MONO_LD;
            dl_velocileptors_monomials(o, nullptr, sigma8, fsigma8, mono + (size_t)pt * DL_STK_ROWS * DL_FG_MONO_LD, DL_FG_MONO_LD, vpv + pt * 12, r);
#pragma unroll
            for (int m = 0; m < DL_FG_MONO_LD; ++m) {
                const int gm = group_of[m];
                const double a = amp[pt * DL_STK_MAX_GROUPS + (gm >= 0 ? gm : 0)];
                if (gm >= 0) row[m] *= a;
            }
        }
    }
    if (st != nullptr && tid == 0) st[2] = __builtin_amdgcn_s_memtime();
}

// the feature GEMM of one device group (cnt = 1 .. 5 monomials) from the basis record `arow` points into
template <int RMAX>
__device__ __forceinline__ void dl_stk_group(int cnt, const double* arow, const dl_fg_double2* gw, int nq, const double* mp, int R, int g, double (&outv)[4][RMAX]) {
    switch (cnt) {
        case 1: dl_stk_group_gemm<1, RMAX, 8>(arow, gw, nq, mp, R, g, outv); break;
        case 2: dl_stk_group_gemm<2, RMAX, 8>(arow, gw, nq, mp, R, g, outv); break;
        case 3: dl_stk_group_gemm<3, RMAX, 6>(arow, gw, nq, mp, R, g, outv); break;
        case 4: dl_stk_group_gemm<4, RMAX, 5>(arow, gw, nq, mp, R, g, outv); break;
        default: dl_stk_group_gemm<5, RMAX, 4>(arow, gw, nq, mp, R, g, outv); break;
    }
}

// the carried rows to memory (no finalize in the tail): out[(p0 + pt) * R + u][jb * 16 + col]
template <int RMAX>
__device__ __forceinline__ void dl_stk_store_rows(const double (&outv)[4][RMAX], int R, double* __restrict__ out, int64_t ldo, int accumulate, int64_t B, int64_t p0, int jb, int col, int g) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int pt = g + 4 * rr;
        if (p0 + pt < B) {
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                if (u < R) {
                    double* dst = out + ((size_t)(p0 + pt) * R + u) * ldo + jb * 16 + col;
                    *dst = accumulate ? *dst + outv[rr][u] : outv[rr][u];
                }
            }
        }
    }
}

// log-priors and NaN flags of the workgroup's 16 points by lanes 0 - 15 of ONE wave (the caller picks it) into lp_lds / nan_lds [16]
__device__ __forceinline__ void dl_stk_priors(const DlStkTail& tl, const double* __restrict__ theta, int n_params, int64_t B, int64_t p0, int lane, double* lp_lds, int* nan_lds) {
    if (n_params <= 32) {
        // four lanes per point, lane (point, q) the terms of the parameters p = q mod 4: every load of the wave goes out at once (a lane per point walking its parameters
        // waited for the prior table four parameters at a time); then the terms are summed IN PARAMETER ORDER (shuffles) -- the sum of dl_marg_priors_lane bit for bit
        const int pt = lane & 15, pq = lane >> 4;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        const double* th = theta + (size_t)b * n_params;
        double xv[8], term[8];
        int nan_in = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = 4 * q + pq < n_params ? th[4 * q + pq] : 0.;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = 4 * q + pq;
            term[q] = p < n_params ? dl_prior_logpdf(tl.priors + 5 * p, xv[q]) : 0.;
            if (p < n_params && xv[q] != xv[q]) nan_in = 1;
        }
        double lp = 0.;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const double t = __shfl(term[p >> 2], pt + 16 * (p & 3), 64);
            if (p < n_params) lp += t;
        }
        nan_in |= __shfl_xor(nan_in, 16, 64);
        nan_in |= __shfl_xor(nan_in, 32, 64);
        if (lane < DL_STK_PTS) { lp_lds[lane] = lp; nan_lds[lane] = nan_in; }
        return;
    }
    if (lane < DL_STK_PTS) {
        const int64_t b = p0 + lane;
        double lp;
        int nan_in;
        dl_marg_priors_lane(theta + (size_t)(b < B ? b : B - 1) * n_params, n_params, tl.priors, lp, nan_in);
        lp_lds[lane] = lp; nan_lds[lane] = nan_in;
    }
}

// the finalize in the tail: rows -> LDS (X [16 points][xr][DL_FG_XLD], over the work area: every wave is past it after the barrier), Gram matrices, solve.
// cpre (or null): the constant parts tl.cst[u][column of this lane] of the rows, requested by the caller ahead of the tail (they are cold: a global round trip at the head
// of the tail otherwise); priors_done: lp_lds / nan_lds were filled by the caller (dl_stk_priors beside the monomial rows) -- else wave 1 fills them here
template <int RMAX>
__device__ __forceinline__ void dl_stk_finalize_tail(const DlStkTail& tl, const double (&outv)[4][RMAX], int R, double* X, const double* __restrict__ theta,
                                                     int n_params, int64_t B, int64_t p0, int tid, int wave, int lane, int col, int g, double* lp_lds, int* nan_lds,
                                                     const double* cpre = nullptr, bool priors_done = false) {
    __syncthreads();
    const int cbase = wave * 16 + col;                   // (N_pad = 128: one workgroup column group, wave = column block)
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        if (u < R) {
            const double c = cpre != nullptr ? cpre[u] : tl.cst[u][cbase];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) X[((size_t)(g + 4 * rr) * tl.xr + tl.row_of[u]) * DL_FG_XLD + cbase] = outv[rr][u] + c;
        }
    }
    for (int c = 0; c < tl.n_const; ++c)
        for (int idx = tid; idx < DL_STK_PTS * 128; idx += 512) {
            const int pt = idx >> 7, cc = idx & 127;
            X[((size_t)pt * tl.xr + tl.const_row[c]) * DL_FG_XLD + cc] = tl.const_ptr[c][cc];
        }
    __syncthreads();
    dl_stk_gram_phase(X, tl.xr, wave, lane, g);            // the 8 x 8 block of point pt at X + pt xr DL_FG_XLD + 8 i + j
    if (!priors_done && wave == 1) dl_stk_priors(tl, theta, n_params, B, p0, lane, lp_lds, nan_lds);
    __syncthreads();
    if (wave == 0 && lane < DL_STK_PTS && p0 + lane < B) {
        const int64_t b = p0 + lane;
        const double* gl = X + (size_t)lane * tl.xr * DL_FG_XLD;
        double ll = 0., lps = 0.;
        bool ok = true;
        switch (tl.mg.n_s) {
            case 0: ll = -0.5 * gl[0]; break;   // no solved parameters: chi2 = |dt|^2 = G[0][0]
            case 1: dl_stk_solve_point<1>(gl, tl, b, ll, lps, ok); break;
            case 2: dl_stk_solve_point<2>(gl, tl, b, ll, lps, ok); break;
            case 3: dl_stk_solve_point<3>(gl, tl, b, ll, lps, ok); break;
            case 4: dl_stk_solve_point<4>(gl, tl, b, ll, lps, ok); break;
            case 5: dl_stk_solve_point<5>(gl, tl, b, ll, lps, ok); break;
            case 6: dl_stk_solve_point<6>(gl, tl, b, ll, lps, ok); break;
            default: dl_stk_solve_point<7>(gl, tl, b, ll, lps, ok); break;
        }
        dl_marg_store_lane(ll, lps, ok, lp_lds[lane], nan_lds[lane] != 0, tl.post_mode, b, tl.loglike, tl.logprior, tl.status);
    }
}

// theta -> residual rows out[B * R, ldo] (+= if accumulate) of one observable; gfrag: [N_pad / 16][sum_g nq_g cnt_g][64][2]; blockIdx.y = group of 8 column blocks
// TMAX: output tiles per layer (4: widths <= 64, 8: <= 128); RMAX: rows carried per point in registers (>= 1 + n_var)
// (the form of round 5: networks of a group, barrier, feature GEMM of the group -- every shape; dl_emu_stacked_ov.h overlaps the two where the shape allows)
template <int TMAX, int RMAX>
__global__ __launch_bounds__(512) void dl_emulated_stacked_kernel(const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ gfrag, const DlObsDev o,
                                                                  double* __restrict__ out, int64_t ldo, int accumulate, int steps_per_block, unsigned long long* stamps, const DlStkTail tl) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * DL_STK_PTS;
    const int R = 1 + o.n_var;
    const int tld = dl_stk_tld(o), bld = dl_stk_bld(o);
    // DL_STK_STAMPS diagnostics (null in production): s_memtime of wave 0 at the phase boundaries, 32 slots per workgroup: 0 entry, 1 inputs, 2 monomial rows, then per device
    // group 3 + 2 gi: networks done (after the barrier), 4 + 2 gi: feature GEMM + epilogue done; 30: rows stored; 31: s_memrealtime at exit (100 MHz)
    unsigned long long* st = stamps != nullptr && blockIdx.y == 0 ? stamps + (size_t)blockIdx.x * 32 : nullptr;
#define DL_STK_STAMP(slot) if (st != nullptr && tid == 0) st[slot] = __builtin_amdgcn_s_memtime();
    DL_STK_STAMP(0)
    constexpr int XLD = DL_STK_XLD;
    const DlStkLds s = dl_stk_carve(lds);
    double* basis = s.work;                                       // [16][bld] basis record of the current group
    double* nbufs = basis + (size_t)DL_STK_PTS * bld;             // [8][16][tld] activation buffers of eight networks
    const int H = o.eng[0].widths[o.eng[0].n_layers];
    dl_stk_prologue(o, theta, n_params, B, p0, tid, lds, basis, tld, R, st);
    // ---- group by group: networks, then the feature GEMM ----
    double outv[4][RMAX];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int u = 0; u < RMAX; ++u) outv[rr][u] = 0.;
    const int jb = blockIdx.y * 8 + wave;
    const dl_fg_double2* gcol = reinterpret_cast<const dl_fg_double2*>(gfrag) + (size_t)jb * steps_per_block * 64 + lane;
    int tb_prev = -1, te_prev = -1;
    for (int gi = 0; gi < o.stk.n_groups; ++gi) {
        const double* rec = o.stk.table + (size_t)gi * DL_STK_REC;
        const int tb = (int)rec[0], te = (int)rec[1], m0 = (int)rec[2], m1 = (int)rec[3], kq = (int)rec[7];
        const int K = (te - tb) * H + 1, nq = (K + 7) / 8;
        if (tb != tb_prev || te != te_prev) {
            __syncthreads();    // the basis record is free (the previous group's GEMM is done); first group: the monomial rows are complete
            for (int t = tb; t < te; t += 8) {     // eight networks at a time (their activation buffers)
                const int n_net = te - t < 8 ? te - t : 8;
                dl_stk_networks<TMAX>(o.eng[0].widths, o.eng[0].n_layers, o.eng[0].act, o.stk.wfrag + (size_t)t * o.stk.frag_doubles, o.stk.frag_doubles, n_net, s.xs, XLD, nbufs, tld,
                                      basis + (size_t)(t - tb) * H, bld, wave, lane, gi == 0 && st != nullptr ? st + 15 : nullptr);
                if (t + 8 < te) __syncthreads();
            }
            for (int idx = tid; idx < DL_STK_PTS * (8 * nq - (K - 1)); idx += 512) {      // the constant basis function and the zero padding of the last step
                const int pt = idx / (8 * nq - (K - 1)), c = K - 1 + (idx - pt * (8 * nq - (K - 1)));
                basis[(size_t)pt * bld + c] = c == K - 1 ? 1. : 0.;
            }
            tb_prev = tb; te_prev = te;
            __syncthreads();
        }
        DL_STK_STAMP(3 + 2 * gi)
        dl_stk_group<RMAX>(m1 - m0, basis + (size_t)col * bld + 2 * g, gcol + (size_t)kq * 64, nq, s.mono + m0, R, g, outv);
        DL_STK_STAMP(4 + 2 * gi)
    }
    if (!tl.enabled) dl_stk_store_rows<RMAX>(outv, R, out, ldo, accumulate, B, p0, jb, col, g);
    else {
        __shared__ double lp_lds[DL_STK_PTS];      // log-priors and NaN flags of the 16 points
        __shared__ int nan_lds[DL_STK_PTS];
        dl_stk_finalize_tail<RMAX>(tl, outv, R, basis, theta, n_params, B, p0, tid, wave, lane, col, g, lp_lds, nan_lds);
    }

    DL_STK_STAMP(30)
    if (st != nullptr && tid == 0) st[31] = __builtin_amdgcn_s_memrealtime();
#undef DL_STK_STAMP
}
#endif
