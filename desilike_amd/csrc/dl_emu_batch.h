// Batched emulator forward pass for the feature path: 16 parameter points per workgroup, every dense layer of the MLP engines as an fp64
// MFMA product [16 points x n_in] . [n_in x n_out] (the per-point kernel dl_emulated_kernel evaluates a layer with one thread per output unit and a
// serial chain of n_in FMAs fed by global loads: 57 us per 4096 points against ~10 us here).  Activations live in LDS, weights stream from L2 in their
// stored [in, out] layout (a B-operand load = 16 consecutive outputs of one input row: 128 contiguous bytes).  Taylor engines, the velocileptors bias
// monomials (one thread per point) and the point records of dl_feature_gemm.h follow.  Device-only (MFMA): validated on the GPU against the oracle and
// the reference fixture (tests/test_gpu_emulator.py), like the feature GEMM.
#pragma once
#include "dl_fullshape.h"
#include "dl_feature_gemm.h"
#include "dl_marg_solve.h"

typedef double dl_eb_double4 __attribute__((ext_vector_type(4)));

#define DL_EB_PTS 16

static inline __host__ __device__ int dl_eb_ld(const DlObsDev& o) {   // LDS row length of an activation buffer: widest layer / most Taylor terms, multiple of 4, + 2
    int w = o.n_x;
    for (int ie = 0; ie < 3; ++ie) {
        const DlObsDev::Engine& e = o.eng[ie];
        if (e.type == 0) { for (int l = 0; l <= e.n_layers; ++l) if (e.widths[l] > w) w = e.widths[l]; }
        else if (e.type == 1 && e.n_terms > w) w = e.n_terms;
    }
    if (o.n_basis > w) w = o.n_basis;
    return (w + 3) / 4 * 4 + 2;
}
static inline __host__ __device__ size_t dl_eb_shared_doubles(const DlObsDev& o) {   // inputs | two activation buffers PER ENGINE | scalars | monomial rows
    return (size_t)DL_EB_PTS * (DL_MAX_X + 6 * dl_eb_ld(o) + 4 + (size_t)(1 + o.n_var) * DL_N_MONO) + DL_EB_PTS * 32;   // (+ theta rows of the 16 points, <= 32 columns: fused forward)
}

#if defined(__HIPCC__)
// forward pass of 16 points with NTHR threads; the point records (basis [nb_pad] | monomial rows [(1 + n_var)][20]) go to rec[pt * rec_stride + c]
// (LDS, fused kernel) or, if rec == nullptr, to the feature buffer in global memory.  `lds`: dl_eb_shared_doubles(o) doubles of workspace.
template <int NTHR, bool TO_LDS>
__device__ __forceinline__ void dl_eb_forward(const DlObsDev& o, const double* __restrict__ theta, int n_params, int64_t B, int64_t p0, double* lds, double* rec, int rec_stride,
                                              double* __restrict__ feat, int64_t feat_ld) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int LD = dl_eb_ld(o);
    double* x = lds;                                      // [16][DL_MAX_X]
    double* bufs = x + DL_EB_PTS * DL_MAX_X;              // engine ie: [2][16][LD] at bufs + ie * 2 * 16 * LD
    double* scal = bufs + 6 * DL_EB_PTS * LD;             // [16][4]: sigma8 (1), fsigma8 (2)
    double* mono = scal + DL_EB_PTS * 4;                  // [16][(1 + n_var) * 19]
    for (int idx = tid; idx < DL_EB_PTS * o.n_x; idx += NTHR) {
        int pt = idx / o.n_x, i = idx - pt * o.n_x;
        int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        x[pt * DL_MAX_X + i] = dl_get(o.x_in[i], theta + (size_t)b * n_params);
    }
    __syncthreads();
    // ---- MLP engines, SIDE BY SIDE: the table basis on half of the waves, the two scalar engines (sigma8, fsigma8) on a quarter each, one barrier per layer for all
    //      of them (one after the other the three engines were 15 barrier-separated layers, each an L2 round trip for its weights: 24 of the fused kernel's 70 us).
    //      A layer = [16 points x n_in] . [n_in x n_out] by MFMA, 16 output units per tile; all k-steps of a tile's weights are requested at once.
    constexpr int NW = NTHR / 64;
    int n_mlp = 0, max_layers = 0;
    for (int ie = 0; ie < 3; ++ie) if (o.eng[ie].type == 0) { ++n_mlp; if (o.eng[ie].n_layers > max_layers) max_layers = o.eng[ie].n_layers; }
    // wave range of engine ie: [w0, w0 + nw)
    int my_ie = -1, my_w0 = 0, my_nw = 0;
    {
        int w0 = 0;
        for (int ie = 0; ie < 3; ++ie) {
            if (o.eng[ie].type != 0) continue;
            int nw = n_mlp == 1 ? NW : (n_mlp == 2 ? NW / 2 : (ie == 0 ? NW / 2 : NW / 4));
            if (nw < 1) nw = 1;
            if (wave >= w0 && wave < w0 + nw) { my_ie = ie; my_w0 = w0; my_nw = nw; }
            w0 += nw;
        }
        if (NW < n_mlp) { my_ie = -2; }   // (fewer waves than engines: not launched that way)
    }
    // scaled inputs (conversion.py:75-77) of every MLP engine, zero-padded to a multiple of 4 columns
    const int nin0 = (o.n_x + 3) & ~3;
    for (int ie = 0; ie < 3; ++ie) {
        const DlObsDev::Engine& e = o.eng[ie];
        if (e.type != 0) continue;
        double* cur = bufs + (size_t)ie * 2 * DL_EB_PTS * LD;
        for (int idx = tid; idx < DL_EB_PTS * nin0; idx += NTHR) {
            int pt = idx / nin0, i = idx - pt * nin0;
            cur[pt * LD + i] = i < o.n_x ? (x[pt * DL_MAX_X + i] - e.xlo[i]) * e.xinv[i] : 0.;
        }
    }
    __syncthreads();
    {
        const DlObsDev::Engine& e = o.eng[my_ie >= 0 ? my_ie : 0];
        double* cur = bufs + (size_t)(my_ie >= 0 ? my_ie : 0) * 2 * DL_EB_PTS * LD;
        double* nxt = cur + DL_EB_PTS * LD;
        const double* w = e.weights;
        for (int layer = 0; layer < max_layers; ++layer) {
            if (my_ie >= 0 && layer < e.n_layers) {
                const int nin = e.widths[layer], nout = e.widths[layer + 1];
                const bool last = (layer == e.n_layers - 1);
                const bool activate = !(last && my_ie != 0);   // the table engine stops after its last HIDDEN layer (its final linear layer is folded on the host)
                const int ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16, nout4 = (nout + 3) & ~3;
                for (int t = wave - my_w0; t < tiles; t += my_nw) {
                    const int oc = 16 * t + col;
                    dl_eb_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};   // two chains: a dependent MFMA waits 64 cycles
                    if (nin == 64 && 16 * t + 16 <= nout) {
                        // a full tile of a 64-input layer (the hidden layers): straight-line -- one pointer and a constant stride for the sixteen weight loads,
                        // immediate offsets for the activations.  The general loop below spends ~1000 instructions per tile on predicates and 64-bit addresses
                        // (3 us per layer for 16 MFMAs).
                        const double* wp = w + (size_t)g * nout + oc;
                        const size_t ws = 4 * (size_t)nout;
                        const double* ap = cur + col * LD + g;
                        double bw[16], av[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) { bw[u] = wp[u * ws]; av[u] = ap[4 * u]; }
#pragma unroll
                        for (int u = 0; u < 16; u += 2) {
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bw[u + 1], acc2, 0, 0, 0);
                        }
                    } else
                    for (int ks0 = 0; ks0 < ksteps; ks0 += 16) {   // sixteen k-steps of weight loads in flight: one L2 round trip per tile for layers up to 64 inputs
                        double bw[16], av[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int k = 4 * (ks0 + u) + g;
                            bw[u] = (ks0 + u < ksteps && k < nin && oc < nout) ? w[(size_t)k * nout + oc] : 0.;   // B[k][output unit]
                            av[u] = (ks0 + u < ksteps) ? cur[col * LD + k] : 0.;                                  // A[point = lane & 15][k]   (columns >= nin are zero)
                        }
#pragma unroll
                        for (int u = 0; u < 16; u += 2) {
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bw[u + 1], acc2, 0, 0, 0);
                        }
                    }
                    acc += acc2;
                    const double bias = oc < nout ? w[(size_t)nin * nout + oc] : 0.;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {       // accumulator register r = out[point g + 4 r][oc]
                        double v = acc[r] + bias;
                        if (activate) v = dl_activation(e.act, v);
                        if (oc < nout4) nxt[(g + 4 * r) * LD + oc] = oc < nout ? v : 0.;
                    }
                }
                w += (size_t)nin * nout + nout;
                double* tmp = cur; cur = nxt; nxt = tmp;
            }
            __syncthreads();
        }
    }
    // results of the MLP engines: scalar engines -> scal (inverse scaler, conversion.py:79); table basis stays in LDS
    double* basis = bufs;
    for (int ie = 0; ie < 3; ++ie) {
        const DlObsDev::Engine& e = o.eng[ie];
        if (e.type != 0) continue;
        double* fin = bufs + (size_t)ie * 2 * DL_EB_PTS * LD + ((e.n_layers & 1) ? DL_EB_PTS * LD : 0);   // the buffer the last layer wrote
        if (ie != 0) { if (tid < DL_EB_PTS) scal[tid * 4 + ie] = fin[tid * LD] * e.yscale + e.ylo; }
        else { basis = fin; if (tid < DL_EB_PTS) basis[tid * LD + o.n_basis - 1] = 1.; }                    // bias row of the folded final layer
    }
    // Taylor engines: monomials prod_p (x_p - c_p)^powers[t, p] (emulators/__init__.py:471-507)
    for (int pass = 0; pass < 3; ++pass) {
        const int ie = pass == 2 ? 0 : pass + 1;
        const DlObsDev::Engine& e = o.eng[ie];
        if (e.type != 1) continue;
        double* nxt = bufs + (size_t)ie * 2 * DL_EB_PTS * LD;
        for (int idx = tid; idx < DL_EB_PTS * e.n_terms; idx += NTHR) {
            int pt = idx / e.n_terms, t = idx - pt * e.n_terms;
            double mon = 1.;
            for (int p = 0; p < o.n_x; ++p) mon *= dl_ipow(x[pt * DL_MAX_X + p] - e.center[p], (int)e.powers[(size_t)t * o.n_x + p]);
            nxt[pt * LD + t] = mon;
        }
        __syncthreads();
        if (ie != 0) {
            if (tid < DL_EB_PTS) {
                double sum = 0.;
                for (int t = 0; t < e.n_terms; ++t) sum = fma(e.coef[t], nxt[tid * LD + t], sum);
                scal[tid * 4 + ie] = sum;
            }
        } else basis = nxt;
    }
    __syncthreads();
    // bias monomials and their derivatives w.r.t. the solved parameters: one thread per point
    if (tid < DL_EB_PTS) {
        int64_t b = p0 + tid < B ? p0 + tid : B - 1;
        const double sigma8 = o.eng[1].type >= 0 ? scal[tid * 4 + 1] : o.eng[1].cst;
        const double fsigma8 = o.eng[2].type >= 0 ? scal[tid * 4 + 2] : o.eng[2].cst;
        dl_velocileptors_monomials(o, theta + (size_t)b * n_params, sigma8, fsigma8, mono + (size_t)tid * (1 + o.n_var) * DL_N_MONO);
    }
    __syncthreads();
    // point records of the feature GEMM: basis [nb_pad] | monomial rows [(1 + n_var)][20]
    const int rec_len = o.nb_pad + (1 + o.n_var) * 20;
    for (int idx = tid; idx < DL_EB_PTS * rec_len; idx += NTHR) {
        int pt = idx / rec_len, c = idx - pt * rec_len;
        double v;
        if (c < o.nb_pad) v = c < o.n_basis ? basis[pt * LD + c] : 0.;
        else { int q = c - o.nb_pad, r = q / 20, m = q - r * 20; v = m < DL_N_MONO ? mono[((size_t)pt * (1 + o.n_var) + r) * DL_N_MONO + m] : 0.; }
        if (TO_LDS) rec[pt * rec_stride + c] = v;
        else if (p0 + pt < B) feat[(size_t)(p0 + pt) * feat_ld + o.feat_off + c] = v;
    }
}

// ---- forward pass of the FUSED kernels (records straight into LDS), restructured around its latencies (in-kernel stamps, config 3: the version above took 19 of the
//      57 us of a workgroup's life with the matrix pipe idle): (i) the weights of a layer do not depend on the activations: every wave requests the weights of its NEXT tile
//      before the activation / barrier of the current layer, the first layer's at kernel entry beside theta; (ii) the velocileptors 'pars' are fetched at entry too and
//      the bias monomials are formed by one otherwise idle wave as soon as the scalar engines (sigma8, fsigma8: fewer layers) have finished, beside the remaining layers
//      of the table engine, straight into the records; (iii) the last hidden layer of the table engine writes the basis into the records itself: no copy pass.  Barriers:
//      entry, one per layer, the caller's (nine before).  Same arithmetic in the same order as dl_eb_forward: results are bit-identical.
template <int NTHR>
__device__ __forceinline__ void dl_eb_forward_fused(const DlObsDev& o, const double* __restrict__ theta, int n_params, int64_t B, int64_t p0, double* lds, double* rec,
                                                    int rec_stride, unsigned long long* st = nullptr, bool th_early = false, double th_val = 0.,
                                                    double* keep_theta = nullptr, double* keep_pr = nullptr, double pr_val = 0., int n_keep_pr = 0) {
    // keep_theta / keep_pr (fused finalize): copies of the parameter rows and of the prior table that outlive the workspace, stored where theta is waited for anyway
    int st_slot = 8;   // DL_EF_STAMPS diagnostics: slots 8.. of the workgroup = after the entry barrier, then after every layer's barrier (the monomials before the first of the second run)
#define DL_EB_STAMP if (st != nullptr && threadIdx.x == 0 && st_slot < 14) st[st_slot] = __builtin_amdgcn_s_memtime(); ++st_slot;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a SCALAR: everything that hangs on it -- the engine of this wave, its layer widths and pointers (fields of the kernel-argument struct), the
    // branches on them -- is then scalar loads and scalar branches; as `tid >> 6` it is a vector value to the compiler, and each of those fields was fetched by a
    // vector load from the kernel-argument segment in the middle of the layer chain
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int LD = dl_eb_ld(o);
    double* x = lds;                                      // [16][DL_MAX_X]
    double* bufs = x + DL_EB_PTS * DL_MAX_X;              // engine ie: [2][16][LD] at bufs + ie * 2 * 16 * LD
    double* scal = bufs + 6 * DL_EB_PTS * LD;             // [16][4]: sigma8 (1), fsigma8 (2)
    double* vpv = scal + DL_EB_PTS * 4;                   // [16][12]: velocileptors 'pars' inputs of the 16 points
    constexpr int NW = NTHR / 64;
    int n_mlp = 0, max_layers = 0, mono_at = 0;
    bool any_taylor = false;
    for (int ie = 0; ie < 3; ++ie) {
        if (o.eng[ie].type == 0) { ++n_mlp; if (o.eng[ie].n_layers > max_layers) max_layers = o.eng[ie].n_layers; if (ie && o.eng[ie].n_layers > mono_at) mono_at = o.eng[ie].n_layers; }
        if (o.eng[ie].type == 1) any_taylor = true;
    }
    int my_ie = -1, my_w0 = 0, my_nw = 1;
    {
        int w0 = 0;
        for (int ie = 0; ie < 3; ++ie) {
            if (o.eng[ie].type != 0) continue;
            int nw = n_mlp == 1 ? NW : (n_mlp == 2 ? NW / 2 : (ie == 0 ? NW / 2 : NW / 4));
            if (nw < 1) nw = 1;
            if (wave >= w0 && wave < w0 + nw) { my_ie = ie; my_w0 = w0; my_nw = nw; }
            w0 += nw;
        }
    }
    const DlObsDev::Engine& e = o.eng[my_ie >= 0 ? my_ie : 0];
    const double* w = e.weights;
    const int t0 = wave - my_w0;                          // first tile of this wave in every layer
    double bw[16], bbias = 0.;
    // Two tile shapes are served from registers requested ahead of their use: kind 2 = a full tile of a 64-input layer (the hidden layers: sixteen k-steps, one
    // pointer and a constant stride, no predicates), kind 1 = a layer with <= 16 inputs (input layers, the scalar engines' output layer: four k-steps, predicated by a
    // clamped address and a select).  Everything else takes the general loop of dl_eb_forward, weights requested where they are used.
    auto tile_kind = [&](int nin, int nout, int t) { return (nin == 64 && 16 * t + 16 <= nout) ? 2 : (nin <= 16 ? 1 : 0); };
    double bwn[16], bbn = 0.;                              // second set: the NEXT layer's weights, requested before the MFMAs of the current one
    auto request_to = [&](double (&dst)[16], double& dbias, const double* wl, int nin, int nout, int t, int kind) {
        const int oc = 16 * t + col;
        if (kind == 2) {
            const double* wp = wl + (unsigned)(g * nout + oc);
            const unsigned ws = 4u * (unsigned)nout;
#pragma unroll
            for (int u = 0; u < 16; ++u) dst[u] = wp[u * ws];
            dbias = wl[(unsigned)(nin * nout + oc)];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * u + g;
                const bool ok = k < nin && oc < nout;
                const double v = wl[ok ? (unsigned)(k * nout + oc) : 0u];
                dst[u] = ok ? v : 0.;
            }
            const double bv = wl[(unsigned)(nin * nout + (oc < nout ? oc : 0))];
            dbias = oc < nout ? bv : 0.;
        }
    };
    auto request = [&](const double* wl, int nin, int nout, int t, int kind) { request_to(bw, bbias, wl, nin, nout, t, kind); };
    int have = 0;                                         // kind of the tile (t0 of the coming layer) whose weights bw / bbias hold, or 0
    if (my_ie >= 0 && mono_at > 0 && e.n_layers > 0 && t0 < (e.widths[1] + 15) / 16) {
        have = tile_kind(e.widths[0], e.widths[1], t0);
        if (have) request(w, e.widths[0], e.widths[1], t0, have);
    }
    // inputs: x (Taylor engines), the scaled inputs of every MLP engine (conversion.py:75-77; zero-padded to a multiple of 4 columns), the 'pars' inputs
    const int nin0 = (o.n_x + 3) & ~3;
    // theta was requested before the first access to `o` (th_val = theta[point tid / 32][column tid % 32]): the rows go through LDS, and the input tables
    // (lane-dependent fields of `o`: vector loads from the kernel-argument segment) are fetched meanwhile -- one round trip at entry instead of two in a row
    double* trow = lds + (size_t)DL_EB_PTS * (DL_MAX_X + 6 * LD + 4 + (size_t)(1 + o.n_var) * DL_N_MONO);   // [16][32]
    if (th_early && DL_EB_PTS * nin0 <= NTHR && DL_EB_PTS * DL_N_VPARS <= NTHR) {
        // one (point, input) and one (point, 'pars' entry) per thread: the descriptors -- lane-dependent fields of `o`, the scalers of the engines: a round trip each
        // to the kernel-argument segment / L2 -- are requested BEFORE the barrier that publishes the theta rows, not after it (two round trips in a row at entry)
        const int pt = tid / nin0, i = tid - pt * nin0;
        const bool xlive = tid < DL_EB_PTS * nin0, xreal = xlive && i < o.n_x;
        const DlInput xin = o.x_in[xreal ? i : 0];
        double xlo[3], xinv[3];
#pragma unroll
        for (int ie = 0; ie < 3; ++ie) {
            const DlObsDev::Engine& en = o.eng[ie];
            xlo[ie] = en.type == 0 ? en.xlo[xreal ? i : 0] : 0.;
            xinv[ie] = en.type == 0 ? en.xinv[xreal ? i : 0] : 0.;
        }
        const int vpt = tid / DL_N_VPARS, vc_ = tid - vpt * DL_N_VPARS;
        const bool vlive = tid < DL_EB_PTS * DL_N_VPARS;
        const DlInput vin = o.vp_in[vlive ? vc_ : 0];
        trow[tid] = th_val;
        if (keep_theta != nullptr) { keep_theta[tid] = th_val; if (tid < n_keep_pr) keep_pr[tid] = pr_val; }
        __syncthreads();
        if (xlive) {
            const double tv = trow[pt * 32 + (xin.col >= 0 ? xin.col : 0)];
            const double v = xreal ? (xin.col >= 0 ? tv : xin.value) : 0.;
            if (xreal) x[pt * DL_MAX_X + i] = v;
#pragma unroll
            for (int ie = 0; ie < 3; ++ie)
                if (o.eng[ie].type == 0) bufs[(size_t)ie * 2 * DL_EB_PTS * LD + pt * LD + i] = xreal ? (v - xlo[ie]) * xinv[ie] : 0.;
        }
        if (vlive) {
            const double tv = trow[vpt * 32 + (vin.col >= 0 ? vin.col : 0)];
            vpv[vpt * 12 + vc_] = vin.col >= 0 ? tv : vin.value;
        }
    } else {
    if (th_early) { trow[tid] = th_val; if (keep_theta != nullptr) { keep_theta[tid] = th_val; if (tid < n_keep_pr) keep_pr[tid] = pr_val; } __syncthreads(); }
    for (int idx = tid; idx < DL_EB_PTS * nin0; idx += NTHR) {
        const int pt = idx / nin0, i = idx - pt * nin0;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        const DlInput xin = o.x_in[i < o.n_x ? i : 0];
        const int xc = xin.col >= 0 ? xin.col : 0;
        const double tv = th_early ? trow[pt * 32 + xc] : theta[(size_t)b * n_params + xc];     // (two loads, not one through a pointer that is LDS or global)
        const double v = i < o.n_x ? (xin.col >= 0 ? tv : xin.value) : 0.;
        if (i < o.n_x) x[pt * DL_MAX_X + i] = v;
        for (int ie = 0; ie < 3; ++ie) {
            const DlObsDev::Engine& en = o.eng[ie];
            if (en.type == 0) bufs[(size_t)ie * 2 * DL_EB_PTS * LD + pt * LD + i] = i < o.n_x ? (v - en.xlo[i]) * en.xinv[i] : 0.;
        }
    }
    for (int idx = tid; idx < DL_EB_PTS * DL_N_VPARS; idx += NTHR) {
        const int pt = idx / DL_N_VPARS, c = idx - pt * DL_N_VPARS;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        const DlInput vin = o.vp_in[c];
        const int vc = vin.col >= 0 ? vin.col : 0;
        const double tv = th_early ? trow[pt * 32 + vc] : theta[(size_t)b * n_params + vc];
        vpv[pt * 12 + c] = vin.col >= 0 ? tv : vin.value;
    }
    }
    // bias row of the folded final layer (MLP table engine) and the zero padding of the basis
    for (int idx = tid; idx < DL_EB_PTS * (o.nb_pad - o.n_basis + 1); idx += NTHR) {
        const int pt = idx / (o.nb_pad - o.n_basis + 1), c = o.n_basis - 1 + (idx - pt * (o.nb_pad - o.n_basis + 1));
        if (c >= o.n_basis) rec[pt * rec_stride + c] = 0.;
        else if (o.eng[0].type == 0) rec[pt * rec_stride + c] = 1.;
    }
    __syncthreads();
    DL_EB_STAMP
    if (any_taylor) {   // Taylor engines: monomials prod_p (x_p - c_p)^powers[t, p] (emulators/__init__.py:471-507); table engine: they ARE the basis
        for (int ie = 0; ie < 3; ++ie) {
            const DlObsDev::Engine& en = o.eng[ie];
            if (en.type != 1) continue;
            double* dst = ie == 0 ? rec : bufs + (size_t)ie * 2 * DL_EB_PTS * LD;
            const int ldd = ie == 0 ? rec_stride : LD;
            for (int idx = tid; idx < DL_EB_PTS * en.n_terms; idx += NTHR) {
                const int pt = idx / en.n_terms, t = idx - pt * en.n_terms;
                double mon = 1.;
                for (int p = 0; p < o.n_x; ++p) mon *= dl_ipow(x[pt * DL_MAX_X + p] - en.center[p], (int)en.powers[(size_t)t * o.n_x + p]);
                dst[pt * ldd + t] = mon;
            }
        }
        __syncthreads();
        for (int ie = 1; ie < 3; ++ie) {
            const DlObsDev::Engine& en = o.eng[ie];
            if (en.type != 1 || tid >= DL_EB_PTS) continue;
            const double* src = bufs + (size_t)ie * 2 * DL_EB_PTS * LD;
            double sum = 0.;
            for (int t = 0; t < en.n_terms; ++t) sum = fma(en.coef[t], src[tid * LD + t], sum);
            scal[tid * 4 + ie] = sum;
        }
        __syncthreads();
    }
    auto monomials = [&]() {   // one lane per (point, row) of the last two waves: 'pars' -> 19 monomials / one derivative row, straight into the records (rows of 20)
        if (wave >= NW - 2) {  // (one lane per point writing all rows took longer than a layer of the table engine beside it)
            const int task = (wave - (NW - 2)) * 64 + lane, pt = task & 15, r = task >> 4;
            if (r < 1 + o.n_var) {
                const double sigma8 = o.eng[1].type >= 0 ? scal[pt * 4 + 1] : o.eng[1].cst;
                const double fsigma8 = o.eng[2].type >= 0 ? scal[pt * 4 + 2] : o.eng[2].cst;
                dl_velocileptors_monomials(o, nullptr, sigma8, fsigma8, rec + (size_t)pt * rec_stride + o.nb_pad, DL_FG_MONO_LD, vpv + pt * 12, r);
            }
        }
    };
    double* cur = bufs + (size_t)(my_ie >= 0 ? my_ie : 0) * 2 * DL_EB_PTS * LD;
    double* nxt = cur + DL_EB_PTS * LD;
    if (mono_at > max_layers) mono_at = max_layers;
    // layers [l0, l1): two calls, the monomials between them -- no prefetched weights are alive across that (register-hungry) code
    auto run_layers = [&](int l0, int l1) {
    for (int layer = l0; layer < l1; ++layer) {
        if (my_ie >= 0 && layer < e.n_layers) {
            const int nin = e.widths[layer], nout = e.widths[layer + 1];
            const bool last = (layer == e.n_layers - 1);
            const bool activate = !(last && my_ie != 0);   // the table engine stops after its last HIDDEN layer (its final linear layer is folded on the host)
            const int ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16, nout4 = (nout + 3) & ~3;
            const double* wn = w + (size_t)nin * nout + nout;   // the next layer's weights
            for (int t = t0; t < tiles; t += my_nw) {
                const int oc = 16 * t + col;
                dl_eb_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};   // two chains (even / odd k-steps), as in dl_eb_forward
                const int kind = tile_kind(nin, nout, t);
                // the weights of this wave's first tile of the NEXT layer go out before the MFMAs of this one (a layer's weights do not depend on activations):
                // requested after them they were still 0.55 us away when the next layer wanted them
                int have_next = 0;
                if (t + my_nw >= tiles && !last && layer + 1 < l1 && t0 < (e.widths[layer + 2] + 15) / 16) {
                    have_next = tile_kind(nout, e.widths[layer + 2], t0);
                    if (have_next && kind && (have && t == t0)) request_to(bwn, bbn, wn, nout, e.widths[layer + 2], t0, have_next);
                    else have_next = -have_next;    // (negative: not requested yet -- after the MFMAs, into the first set)
                }
                if (kind) {
                    if (!(have && t == t0)) request(w, nin, nout, t, kind);
                    const double* ap = cur + col * LD + g;
                    if (kind == 2) {
                        double av[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) av[u] = ap[4 * u];
#pragma unroll
                        for (int u = 0; u < 16; u += 2) {
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bw[u + 1], acc2, 0, 0, 0);
                        }
                    } else {
                        double av[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) av[u] = ap[u < ksteps ? 4 * u : 0];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bw[0], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bw[1], acc2, 0, 0, 0);
                        if (ksteps > 2) {
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bw[2], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bw[3], acc2, 0, 0, 0);
                        }
                    }
                } else {
                    for (int ks0 = 0; ks0 < ksteps; ks0 += 4) {   // (four k-steps at a time: sixteen made this rarely taken path the register peak of the kernel)
                        double bq[4], av[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = 4 * (ks0 + u) + g;
                            bq[u] = (ks0 + u < ksteps && k < nin && oc < nout) ? w[(size_t)k * nout + oc] : 0.;
                            av[u] = (ks0 + u < ksteps) ? cur[col * LD + k] : 0.;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u += 2) {
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bq[u], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bq[u + 1], acc2, 0, 0, 0);
                        }
                    }
                    bbias = oc < nout ? w[(size_t)nin * nout + oc] : 0.;
                }
                acc += acc2;
                const double bias = bbias;
                have = 0;
                if (have_next > 0) {
                    have = have_next;
#pragma unroll
                    for (int u = 0; u < 16; ++u) bw[u] = bwn[u];
                    bbias = bbn;
                } else if (have_next < 0) { have = -have_next; request(wn, nout, e.widths[layer + 2], t0, have); }
                // accumulator register r = out[point g + 4 r][oc].  The branch on the activation sits OUTSIDE the four evaluations: inside (dl_activation per value)
                // each exponential was a basic block of its own and the four dependent chains ran one after the other (1.2 us of a 2.2 us layer)
                double vv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = acc[r] + bias;
                if (activate) {
                    if (e.act == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[r] = vv[r] / (1. + exp(-vv[r]));      // silu, conversion.py:29 (the expression of dl_activation)
                    } else if (e.act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[r] = vv[r] > 0. ? vv[r] : 0.;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[r] = tanh(vv[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = vv[r];
                    if (last && my_ie == 0) { if (oc < nout) rec[(g + 4 * r) * rec_stride + oc] = v; }                       // the basis of the table engine
                    else if (last) { if (oc == 0) scal[(g + 4 * r) * 4 + my_ie] = v * e.yscale + e.ylo; }                    // inverse scaler, conversion.py:79
                    else if (oc < nout4) nxt[(g + 4 * r) * LD + oc] = oc < nout ? v : 0.;
                }
            }
            w = wn;
            double* tmp = cur; cur = nxt; nxt = tmp;
        }
        __syncthreads();
        DL_EB_STAMP
    }
    };
    run_layers(0, mono_at);
    have = 0;
    monomials();
    run_layers(mono_at, max_layers);
#undef DL_EB_STAMP
}

// records to the feature buffer (the feature GEMM follows as a separate launch)
__global__ __launch_bounds__(256) void dl_emulated_batch_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, int64_t B, double* __restrict__ feat,
                                                                int64_t feat_ld) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    dl_eb_forward<256, false>(o, theta, n_params, B, (int64_t)blockIdx.x * DL_EB_PTS, lds, lds, 0, feat, feat_ld);
}

// FUSED: theta -> emulator forward (MFMA) -> point records in LDS -> feature GEMM -> residual rows: the records never leave the CU and one launch
// (ramp, completion, cold hand-over) disappears.  LDS: forward workspace, then the 16 records.
static inline __host__ __device__ size_t dl_ef_shared_doubles(const DlObsDev& o) {
    return (dl_eb_shared_doubles(o) + 1) / 2 * 2 + (size_t)DL_FG_PTS * dl_fg_lds_stride(o.nb_pad + (1 + o.n_var) * DL_FG_MONO_LD);
}
__global__ __launch_bounds__(512) void dl_emulated_feature_kernel(const DlObsDev o, const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ gfrag,
                                                                  double* __restrict__ out, int64_t ldo, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t p0 = (int64_t)blockIdx.x * DL_EB_PTS;
    const int R = 1 + o.n_var;
    const int stride = dl_fg_lds_stride(o.nb_pad + R * DL_FG_MONO_LD);
    double* rec = lds + (dl_eb_shared_doubles(o) + 1) / 2 * 2;
    dl_eb_forward_fused<512>(o, theta, n_params, B, p0, lds, rec, stride);
    __syncthreads();
    dl_fg_compute(rec, stride, o.nb_pad, R, gfrag, out, ldo, B, p0, accumulate);
}

// ... with the Gram-matrix epilogue (dl_feature_gemm.h): LDS = forward workspace | records | X rows of the 16 points
struct DlEfGramArgs {
    int xr, n_const;                   // rows of X (1 + n_s); solved parameters without a point-dependent row
    int row_of[6];                     // X row of device row r
    const double* cst[6];              // constant part of device row r
    int const_row[DL_MAX_SOLVED];      // X rows that are constants only ...
    const double* const_ptr[DL_MAX_SOLVED];   // ... and their tconst rows
    double* gram;
    unsigned long long* stamps;        // DL_EF_STAMPS diagnostics
    int nz[6][2];                      // support of the derivative rows (dl_velocileptors_row_support)
    int scaled;                        // DlFgGram::scaled
    int no_early;                      // DL_EF_NO_EARLY_THETA=1 (tests): the path of more than 32 sampled parameters -- theta rows and prior table read from memory where they are used
};
// The marginalised finalize in the tail of the same kernel (n_s <= 7): the Gram blocks of the workgroup's 16 points stay in LDS, lanes 0-15 of wave 0 solve a point
// each (dl_marg_solve.h) while lanes 0-15 of wave 1 sum the priors of the same points; the other waves have left.  Against a separate launch (4.3 us, of which a
// wave lives 1.3 us, plus the stream-order gap before it): no Gram matrix through memory, one kernel boundary less per step.
struct DlEfSolve {
    int enabled, post_mode;
    const double* priors;
    double *loglike, *logprior;
    int32_t* status;
    double *solved, *hessian;
    DlMargDev mg;
};
// LDS: the 16 records | union(forward workspace, X rows): the workspace is dead once the records are written
static inline __host__ __device__ size_t dl_ef_gram_rec_doubles(const DlObsDev& o) { return ((size_t)DL_FG_PTS * dl_fg_lds_stride(o.nb_pad + (1 + o.n_var) * DL_FG_MONO_LD) + 1) / 2 * 2; }
static inline __host__ __device__ size_t dl_ef_gram_shared_doubles(const DlObsDev& o, int xr) {
    const size_t work = dl_eb_shared_doubles(o), x = (size_t)DL_FG_PTS * xr * DL_FG_XLD;
    return dl_ef_gram_rec_doubles(o) + (work > x ? work : x);
}
// (theta, n_params, B first: they arrive in SGPRs with the wave (kernel-argument preload), and the theta rows of the 16 points are requested before the first access to
//  the descriptor `o` -- that access is a round trip to the kernel-argument segment which the theta loads used to wait for: two round trips in a row at entry)
template <int NS>
__device__ __forceinline__ void dl_ef_solve_point(const double* gl, const DlEfSolve& sv, int64_t b, double& ll, double& lps, bool& ok) {
    ll = dl_marg_solve_lane<NS>([&](int i, int j) { return gl[i * 8 + j]; }, sv.mg, sv.solved ? sv.solved + (size_t)b * NS : nullptr,
                                sv.hessian ? sv.hessian + (size_t)b * NS * NS : nullptr, lps, ok);
}

__global__ __launch_bounds__(512) void dl_emulated_feature_gram_kernel(const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ gfrag, const DlObsDev o,
                                                                       const DlEfGramArgs ga, const DlEfSolve sv) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    {   // Every 64-byte line of the kernel-argument segment (the descriptor `o` is 1.5 KB) is touched by one batch of scalar loads at entry: the fields are read
        // where they are first needed -- engine after engine, layer after layer, then `ga` -- and each first touch of a line was a miss of the scalar cache in
        // the middle of a dependent chain; now they are hits.
        constexpr int n_lines = (int)((32 + sizeof(DlObsDev) + sizeof(DlEfGramArgs) + sizeof(DlEfSolve) + 63) / 64);
        const __attribute__((address_space(4))) int* kargs = (const __attribute__((address_space(4))) int*)__builtin_amdgcn_kernarg_segment_ptr();
        int touched = 0;
#pragma unroll
        for (int l = 0; l < n_lines; ++l) touched |= kargs[16 * l];
        asm volatile("; kernel arguments touched: %0" :: "s"(touched));
    }
    const int64_t p0 = (int64_t)blockIdx.x * DL_EB_PTS;
    const bool th_early = n_params <= 32 && !ga.no_early;
    double th_val = 0.;
    if (th_early) {
        const int pt = threadIdx.x >> 5, j = threadIdx.x & 31;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        th_val = theta[(size_t)b * n_params + (j < n_params ? j : 0)];
    }
    // fused finalize: the parameter rows and the prior table stay in LDS for the tail (requested here with the rest of the entry loads, stored by the forward pass
    // where it waits for theta: a store after the forward pass let the compiler sink the table's load there -- a global round trip, 0.6 us, in front of the feature GEMM)
    const bool keep = sv.enabled && th_early;
    __shared__ double keep_theta[DL_FG_PTS * 32], keep_pr[32 * 5], prior_term[DL_FG_PTS][33], lp_lds[DL_FG_PTS];
    __shared__ int nan_lds[DL_FG_PTS];
    double pr_val = 0.;
    if (keep && (int)threadIdx.x < 5 * n_params) pr_val = sv.priors[threadIdx.x];
    const int R = 1 + o.n_var;
    const int stride = dl_fg_lds_stride(o.nb_pad + R * DL_FG_MONO_LD);
    double* rec = lds;
    double* work = lds + dl_ef_gram_rec_doubles(o);
    if (ga.stamps != nullptr && threadIdx.x == 0) { ga.stamps[(size_t)blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memtime(); ga.stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime(); }
    dl_eb_forward_fused<512>(o, theta, n_params, B, p0, work, rec, stride, ga.stamps != nullptr ? ga.stamps + (size_t)blockIdx.x * 16 : nullptr, th_early, th_val,
                             keep ? keep_theta : nullptr, keep_pr, pr_val, 5 * n_params);
    DlFgGram gr;
    gr.x = work;
    gr.xr = ga.xr; gr.gram = sv.enabled ? nullptr : ga.gram; gr.stamps = ga.stamps; gr.scaled = ga.scaled;
#pragma unroll
    for (int r = 0; r < 6; ++r) { gr.nz[r][0] = ga.nz[r][0]; gr.nz[r][1] = ga.nz[r][1]; }
#pragma unroll
    for (int r = 0; r < 6; ++r) { gr.row_of[r] = ga.row_of[r]; gr.cst[r] = ga.cst[r]; }
    // (the first operand request of the feature GEMM goes out, THEN:) the records are complete, the forward workspace is free: X takes its place
    auto after_request = [&]() {
        // a barrier that waits for this wave's LDS traffic only: __syncthreads() also waits for the operand request that was just issued (vmcnt(0)); the records are
        // LDS writes, and what the other waves read after the barrier is LDS
        asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
        if (ga.stamps != nullptr && threadIdx.x == 0) ga.stamps[(size_t)blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memtime();
        // rows of solved parameters whose derivative does not depend on the point: the constant itself (visible to the Gram phase after the barrier before it)
        for (int c = 0; c < ga.n_const; ++c)
            for (int idx = threadIdx.x; idx < DL_FG_PTS * 128; idx += 512) {
                const int pt = idx >> 7, col = idx & 127;
                gr.x[((size_t)pt * gr.xr + ga.const_row[c]) * DL_FG_XLD + col] = ga.const_ptr[c][col];
            }
    };
    // MLP table engine with n_basis = 8 j + 1: the constant basis function sits alone in the last k-step pair of the operand
    const bool bias_pair = o.eng[0].type == 0 && o.n_basis == o.nb_pad - 7 && o.nb_pad >= 16;
    dl_fg_compute_gram(rec, stride, o.nb_pad, R, gfrag, B, p0, &gr, after_request, bias_pair);
    if (sv.enabled) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const int64_t b = p0 + (lane & 15);
        if (wave == 1) {
            // priors of the 16 points, BEFORE the barrier (beside whatever the slower waves still do): one (point, parameter) term per lane and round from LDS, then lanes
            // 0-15 sum the terms of their point in parameter order -- the sum of dl_marg_priors_lane bit for bit
            const int pt = lane & 15;
            if (keep) {
                for (int p = lane >> 4; p < n_params; p += 4) prior_term[pt][p] = dl_prior_logpdf(keep_pr + 5 * p, keep_theta[pt * 32 + p]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < DL_FG_PTS) {
                    double lp = 0.;
                    int nan_in = 0;
                    for (int p = 0; p < n_params; ++p) { const double x = keep_theta[pt * 32 + p]; if (x != x) nan_in = 1; lp += prior_term[pt][p]; }
                    lp_lds[lane] = lp; nan_lds[lane] = nan_in;
                }
            } else if (lane < DL_FG_PTS) {
                double lp;
                int nan_in;
                dl_marg_priors_lane(theta + (size_t)(b < B ? b : B - 1) * n_params, n_params, sv.priors, lp, nan_in);
                lp_lds[lane] = lp; nan_lds[lane] = nan_in;
            }
        }
        __syncthreads();   // the Gram blocks of the 16 points and their priors are in LDS
        if (wave >= 1) return;
        if (lane < DL_FG_PTS && b < B) {
            double ll = 0., lps = 0.;
            bool ok = true;
            const double* gl = gr.x + (size_t)lane * gr.xr * DL_FG_XLD;
            switch (sv.mg.n_s) {
                case 0: ll = -0.5 * gl[0]; break;   // no solved parameters: chi2 = |dt|^2 = G[0][0]
                case 1: dl_ef_solve_point<1>(gl, sv, b, ll, lps, ok); break;
                case 2: dl_ef_solve_point<2>(gl, sv, b, ll, lps, ok); break;
                case 3: dl_ef_solve_point<3>(gl, sv, b, ll, lps, ok); break;
                case 4: dl_ef_solve_point<4>(gl, sv, b, ll, lps, ok); break;
                case 5: dl_ef_solve_point<5>(gl, sv, b, ll, lps, ok); break;
                case 6: dl_ef_solve_point<6>(gl, sv, b, ll, lps, ok); break;
                default: dl_ef_solve_point<7>(gl, sv, b, ll, lps, ok); break;
            }
            dl_marg_store_lane(ll, lps, ok, lp_lds[lane], nan_lds[lane] != 0, sv.post_mode, b, sv.loglike, sv.logprior, sv.status);
        }
    }
    if (ga.stamps != nullptr && threadIdx.x == 0) ga.stamps[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime();   // (100 MHz, chip-wide: calibrates the shader clock)
}
#endif
