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
//   Workgroup = 16 points x 128 output columns, 512 threads.  Per group: (i) NETWORKS: wave w runs networks tb + w, tb + w + 8, ... of the group from the scaled inputs
//   to the last hidden layer ALONE -- every dense layer is [16 points x n_in] . [n_in x n_out] by v_mfma_f64_16x16x4_f64, activations of the wave in its own LDS
//   buffer, updated in place (all output tiles of a layer sit in registers before the first is written), so no barrier separates the layers of a network and the
//   networks of a group run side by side on the four SIMDs; weights stream from L2 in their stored [in, out] layout (a request = the sixteen k-steps of the NEXT
//   output tile, issued before the MFMAs of the current one); the last hidden layer lands in the group's basis record.  (ii) FEATURE GEMM: wave w = column block w;
//   the operand streams from L2 in fragment order [column block][group][k / 8][m][lane][2], two steps in flight; the epilogue contracts the accumulators with the
//   amplitude-scaled monomial rows of the lane's four points into the carried output rows (registers; at most 8 rows: residual + the seven alpha* / sn* that can be solved).
#pragma once
#include "dl_fullshape.h"
#include "dl_feature_gemm.h"

typedef double dl_stk_double4 __attribute__((ext_vector_type(4)));

#define DL_STK_PTS 16
#define DL_STK_ROWS 8        // rows carried per point: 1 + the solvable alpha0 alpha2 alpha4 alpha6 sn0 sn2 sn4 (full_shape.py:1226)
#define DL_STK_TMAX 8        // output tiles of a layer (widths <= 128)

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
    return dl_stk_shared_doubles(o) * sizeof(double) <= 160 * 1024;
}

#if defined(__HIPCC__)
// One network on one wave: 16 points from `in0` (row stride ld0; layer 0; columns beyond the inputs zero up to a multiple of 4) through `n_layers` dense layers, every one
// activated (a table network stops after its last HIDDEN layer); hidden activations in the wave's buffer `buf` (row stride tld), the last layer's output to
// dst[point * dst_ld + unit].  `wf`: the network's weights in fragment order (DlObsDev::Stack::wfrag): a B-operand load is base + lane + immediate, no predicates.
template <int TMAX>
__device__ __forceinline__ void dl_stk_network(const int32_t* widths, int n_layers, int act, const double* __restrict__ wf, const double* in0, int ld0, double* buf, int tld,
                                               double* dst, int dst_ld, int lane) {
    const int col = lane & 15, g = lane >> 4;
    wf += lane;
    for (int layer = 0; layer < n_layers; ++layer) {
        const int nin = widths[layer], nout = widths[layer + 1];
        const int ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16;
        const bool last = layer == n_layers - 1;
        const double* ap = (layer == 0 ? in0 + col * ld0 : buf + col * tld) + g;
        dl_stk_double4 res[TMAX];
        if (ksteps == 16) {
            // a 64-input layer (the hidden layers): sixteen k-steps per output tile, the weights of the NEXT tile requested before the MFMAs of the current one
            double bw[16], bwn[16], av[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) bw[u] = wf[u * 64];
#pragma unroll
            for (int u = 0; u < 16; ++u) av[u] = ap[4 * u];              // the A operand is the same for every output tile
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t >= tiles) break;
                if (t + 1 < tiles) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) bwn[u] = wf[((t + 1) * 16 + u) * 64];
                }
                dl_stk_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};   // two chains: a dependent MFMA waits for its predecessor
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bw[u + 1], acc2, 0, 0, 0);
                }
                res[t] = acc + acc2;
#pragma unroll
                for (int u = 0; u < 16; ++u) bw[u] = bwn[u];
            }
        } else {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t >= tiles) break;
                dl_stk_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};
                const double* wt = wf + (size_t)t * ksteps * 64;
                for (int u = 0; u < ksteps; u += 2) {
                    const bool two = u + 1 < ksteps;
                    const double b0 = wt[u * 64], b1 = wt[(two ? u + 1 : u) * 64];
                    const double a0 = ap[4 * u], a1 = ap[4 * (two ? u + 1 : u)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                    if (two) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc2, 0, 0, 0);
                }
                res[t] = acc + acc2;
            }
        }
        const double* bias = wf + (size_t)tiles * ksteps * 64 - lane + col;      // [tile][16]
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t >= tiles) break;
            const int oc = 16 * t + col;
            const double b = bias[16 * t];
            double vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = res[t][r] + b;          // accumulator register r = out[point g + 4 r][oc]
            if (act == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = vv[r] / (1. + exp(-vv[r]));      // silu, conversion.py:29
            } else if (act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = vv[r] > 0. ? vv[r] : 0.;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = tanh(vv[r]);
            }
            // hidden layers: in place -- every read of this layer precedes (one wave: LDS operations complete in order); units beyond the layer (zero weights and bias) are
            // written too when they pad the next layer's k-steps: act(0) of silu / relu / tanh is 0
            const int nout4 = (nout + 3) & ~3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!last) { if (oc < nout4) buf[(g + 4 * r) * tld + oc] = vv[r]; }
                else if (oc < nout) dst[(g + 4 * r) * dst_ld + oc] = vv[r];
            }
        }
        wf += (size_t)tiles * ksteps * 64 + 16 * tiles;
    }
}

// feature GEMM of one group: CNT monomials, nq operand steps of 8 basis functions; then the contraction with the (amplitude-scaled) monomial rows into the carried rows
template <int CNT, int RMAX>
__device__ __forceinline__ void dl_stk_group_gemm(const double* arow, const dl_fg_double2* __restrict__ gw, int nq, const double* mono, int R, int g, double (&outv)[4][RMAX]) {
    dl_fg_double4 acc[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] = (dl_fg_double4){0., 0., 0., 0.};
    dl_fg_double2 b0[CNT], b1[CNT], b2[CNT];
    const int q1 = 1 < nq ? 1 : nq - 1;
#pragma unroll
    for (int i = 0; i < CNT; ++i) { b0[i] = gw[(size_t)i * 64]; b1[i] = gw[(size_t)(q1 * CNT + i) * 64]; }
    for (int q = 0; q < nq; ++q) {
        const int qn = q + 2 < nq ? q + 2 : nq - 1;
#pragma unroll
        for (int i = 0; i < CNT; ++i) b2[i] = gw[(size_t)(qn * CNT + i) * 64];
        const dl_fg_double2 a = *reinterpret_cast<const dl_fg_double2*>(arow + 8 * q);
#pragma unroll
        for (int i = 0; i < CNT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b0[i].x, acc[i], 0, 0, 0);      // (the two MFMAs of an accumulator CNT instructions apart)
#pragma unroll
        for (int i = 0; i < CNT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b0[i].y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CNT; ++i) { b0[i] = b1[i]; b1[i] = b2[i]; }
    }
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

// theta -> residual rows out[B * R, ldo] (+= if accumulate) of one observable; gfrag: [N_pad / 16][sum_g nq_g cnt_g][64][2]; blockIdx.y = group of 8 column blocks
// TMAX: output tiles per layer (4: widths <= 64, 8: <= 128); RMAX: rows carried per point in registers (>= 1 + n_var)
template <int TMAX, int RMAX>
__global__ __launch_bounds__(512) void dl_emulated_stacked_kernel(const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ gfrag, const DlObsDev o,
                                                                  double* __restrict__ out, int64_t ldo, int accumulate, int steps_per_block) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * DL_STK_PTS;
    const int R = 1 + o.n_var;
    const int tld = dl_stk_tld(o), bld = dl_stk_bld(o);
    constexpr int XLD = DL_MAX_X + 2;
    double* x = lds;                                           // [16][DL_MAX_X] the emulator inputs
    double* xs = x + DL_STK_PTS * DL_MAX_X;                    // [3][16][XLD] scaled inputs of the table networks (0) and of the scalar engines (1, 2), zero-padded
    double* amp = xs + 3 * DL_STK_PTS * XLD;                   // [16][8] amplitude of every group
    double* scal = amp + DL_STK_PTS * DL_STK_MAX_GROUPS;       // [16][4]: sigma8 (1), fsigma8 (2)
    double* vpv = scal + DL_STK_PTS * 4;                       // [16][12] velocileptors 'pars' inputs
    double* mono = vpv + DL_STK_PTS * 12;                      // [16][DL_STK_ROWS][20] monomial rows, scaled by the amplitude of their group
    double* basis = mono + DL_STK_PTS * DL_STK_ROWS * DL_FG_MONO_LD;   // [16][bld] basis record of the current group
    double* wbuf = basis + (size_t)DL_STK_PTS * bld + (size_t)wave * DL_STK_PTS * tld;   // this wave's activation buffer
    const int H = o.eng[0].widths[o.eng[0].n_layers];
    // ---- inputs ----
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
    __syncthreads();
    // ---- scalar engines (sigma8, fsigma8: the physical prior basis; small networks): a thread per (point, unit), layer by layer, in the still unused basis record;
    //      the amplitudes of the groups ----
    for (int ie = 1; ie < 3; ++ie) {
        const DlObsDev::Engine& en = o.eng[ie];
        if (en.type != 0) continue;
        double* cur = basis;
        double* nxt = basis + DL_STK_PTS * tld;
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
    __syncthreads();
    // ---- monomial rows: one lane per (point, row), then scaled group by group (a monomial belongs to one group; monomials of no group feed nothing) ----
    if (tid < DL_STK_PTS * DL_STK_ROWS) {
        const int pt = tid & 15, r = tid >> 4;
        if (r < R) {
            const double sigma8 = o.eng[1].type >= 0 ? scal[pt * 4 + 1] : o.eng[1].cst;
            const double fsigma8 = o.eng[2].type >= 0 ? scal[pt * 4 + 2] : o.eng[2].cst;
            double* row = mono + ((size_t)pt * DL_STK_ROWS + r) * DL_FG_MONO_LD;
            dl_velocileptors_monomials(o, nullptr, sigma8, fsigma8, mono + (size_t)pt * DL_STK_ROWS * DL_FG_MONO_LD, DL_FG_MONO_LD, vpv + pt * 12, r);
            for (int gi = 0; gi < o.stk.n_groups; ++gi) {
                const double* rec = o.stk.table + (size_t)gi * DL_STK_REC;
                const double a = amp[pt * DL_STK_MAX_GROUPS + gi];
                for (int m = (int)rec[2]; m < (int)rec[3]; ++m) row[m] *= a;
            }
        }
    }
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
            for (int t = tb + wave; t < te; t += 8)
                dl_stk_network<TMAX>(o.eng[0].widths, o.eng[0].n_layers, o.eng[0].act, o.stk.wfrag + (size_t)t * o.stk.frag_doubles, xs, XLD, wbuf, tld,
                               basis + (size_t)(t - tb) * H, bld, lane);
            for (int idx = tid; idx < DL_STK_PTS * (8 * nq - (K - 1)); idx += 512) {      // the constant basis function and the zero padding of the last step
                const int pt = idx / (8 * nq - (K - 1)), c = K - 1 + (idx - pt * (8 * nq - (K - 1)));
                basis[(size_t)pt * bld + c] = c == K - 1 ? 1. : 0.;
            }
            tb_prev = tb; te_prev = te;
            __syncthreads();
        }
        const double* arow = basis + (size_t)col * bld + 2 * g;
        const dl_fg_double2* gw = gcol + (size_t)kq * 64;
        const double* mp = mono + m0;
        switch (m1 - m0) {
            case 1: dl_stk_group_gemm<1, RMAX>(arow, gw, nq, mp, R, g, outv); break;
            case 2: dl_stk_group_gemm<2, RMAX>(arow, gw, nq, mp, R, g, outv); break;
            case 3: dl_stk_group_gemm<3, RMAX>(arow, gw, nq, mp, R, g, outv); break;
            case 4: dl_stk_group_gemm<4, RMAX>(arow, gw, nq, mp, R, g, outv); break;
            default: dl_stk_group_gemm<5, RMAX>(arow, gw, nq, mp, R, g, outv); break;
        }
    }
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
#endif
