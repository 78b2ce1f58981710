// Stacked table engine, round 6: the networks of the NEXT batch run UNDER the feature GEMM of the current group (emulators/conversion.py:44-98,
// full_shape.py:1182-1186, 1416-1443 -- the arithmetic and the operand layouts are those of dl_emu_stacked.h; what changes is WHEN a wave does what).
//
//   dl_emulated_stacked_kernel alternates, group by group, a network phase (a chain of latencies: per layer 18 k cycles for 6 k of MFMA time and 5 k of activations, the
//   matrix pipe idle most of the time) and a feature GEMM (0.90 of its MFMA time).  The networks of batch b + 1 depend on the inputs only, so here they run while the
//   feature GEMM of the first device group of batch b streams: the eight waves of the workgroup split into the halves A = waves 0-3 and B = waves 4-7 (wave w and w + 4 share
//   a SIMD), and a group with a pending batch takes two slots
//       slot 1:  A: feature GEMM of the group, column blocks 0-3      B: layers [0, ls) of the next batch
//       slot 2:  B: feature GEMM of the group, column blocks 4-7      A: layers [ls, n_layers) of the next batch
//   so every SIMD holds one streaming wave and one latency-bound wave (raised priority: it issues whenever it can, the GEMM fills the rest), and both halves carry the same
//   load.  Two basis records in LDS (current / next); the activations of a network live IN its 64 columns of the next record (hidden widths <= the last one), which
//   removes the activation buffers.  The four waves of a half synchronise layer by layer through a counter in LDS (s_barrier would stop the GEMM half); the halves meet at a
//   workgroup barrier per slot.  The first batch has nothing to hide under: all eight waves run it (workgroup barriers).  Groups whose batch is already there, and the empty
//   batch of constant tables ('st'), run the feature GEMM on all eight waves as before.
#pragma once
#include "dl_emu_stacked.h"

#define DL_STKO_TMAX 6          // output-tile tasks of a layer per wave when four waves run the networks (n_networks x tiles <= 24: six networks of 64 units)

static inline __host__ __device__ size_t dl_stko_fixed_doubles() {
    return (size_t)DL_STK_PTS * (DL_MAX_X + 3 * (DL_MAX_X + 2) + DL_STK_MAX_GROUPS + 4 + 12 + DL_STK_ROWS * DL_FG_MONO_LD);
}
// work area: two basis records (or the scratch of the scalar engines, 2 x 16 x tld)
static inline __host__ __device__ size_t dl_stko_work_doubles(const DlObsDev& o) {
    const size_t w = (size_t)2 * DL_STK_PTS * dl_stk_bld(o), s = (size_t)2 * DL_STK_PTS * dl_stk_tld(o);
    return w > s ? w : s;
}
// shapes the overlapped kernel takes: the last hidden width a multiple of 4 and the widest (activations in place in the record), every batch within the task budget of
// four waves, two records within the LDS; everything else stays with dl_emulated_stacked_kernel
static inline bool dl_stko_ok(const DlObsDev& o) {
    if (!dl_stk_feature_ok(o)) return false;
    const DlObsDev::Engine& e = o.eng[0];
    const int H = e.widths[e.n_layers];
    if (H % 4 != 0 || H > 128) return false;
    for (int l = 1; l <= e.n_layers; ++l) if (e.widths[l] > H) return false;
    if (o.stk.max_net * ((H + 15) / 16) > 4 * DL_STKO_TMAX) return false;
    return (dl_stko_fixed_doubles() + dl_stko_work_doubles(o)) * sizeof(double) + DL_STK_STATIC_LDS <= 160 * 1024;
}

#if defined(__HIPCC__)
// The activations of NV values side by side, one instruction per value and step: a value alone is a chain of ~32 DEPENDENT fp64 instructions, and a dependent fp64
// instruction issues ~16 cycles after its predecessor -- written value by value (what the compiler emits from dl_stk_act under register pressure) the activations of a task
// took 2 k cycles for 130 instructions; four chains in step issue back to back (4 cycles per instruction).  The scheduling barriers pin the order.
// v_max_f64 / v_min_f64 as single instructions: fmax / fmin compile to the instruction PLUS a canonicalisation of every operand that is not known to be quiet (v_max_f64 x, x, x:
// two extra instructions per value in front of the clamps below).  In the IEEE mode of compute kernels the instruction itself returns the other operand for a NaN, as fmax / fmin do.
__device__ __forceinline__ double dl_vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double dl_vmax_neg(double a, double b) { double r; asm("v_max_f64 %0, -%1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double dl_vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ double dl_vmax0(double a) { double r; asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }

template <int NV>
__device__ __forceinline__ void dl_stk_act_rows(int act, double (&v)[NV]) {
#define DL_STK_ROW(body) { _Pragma("unroll") for (int r = 0; r < NV; ++r) { body; } __builtin_amdgcn_sched_barrier(0); }
    if (act == 1) { DL_STK_ROW(v[r] = dl_vmax0(v[r])) return; }                                // conversion.py:31 (a NaN gives 0, as v > 0 ? v : 0 does)
    // e^x with full-rate instructions only (round 6: v_rndne_f64, v_cvt_i32_f64 and v_ldexp_f64 issue at a quarter of the rate of an FMA): n = rint(x log2 e) by the
    // 1.5 x 2^52 constant (the integer sits in the low dword of the sum), Cody-Waite reduction, degree-11 polynomial, 2^n by an integer addition to the exponent field
    // (x clamped to [-708, 709]: the result stays a normal number); then one v_rcp_f64 + one cubic step
    double x[NV], n[NV], p[NV];
    int ni[NV];
    // silu v / (1 + e^-v) (conversion.py:29): the exponent is x = -v.  tanh 1 - 2 / (1 + e^2v) (conversion.py:33): the exponent is 2 v -- the factor two sits in the constants (log2 e doubled,
    // ln 2 halved, coefficient k times 2^k: exact scalings), not in a row of its own: x below is HALF the exponent on that path
    if (act == 0) DL_STK_ROW(x[r] = dl_vmax_neg(v[r], -708.))
    else DL_STK_ROW(x[r] = dl_vmax(v[r], -354.))
    DL_STK_ROW(x[r] = dl_vmin(x[r], act == 0 ? 709. : 354.5))
    DL_STK_ROW(n[r] = fma(x[r], act == 0 ? 1.4426950408889634074 : 2.8853900817779268148, 6755399441055744.))
    DL_STK_ROW(ni[r] = __double2loint(n[r]))
    DL_STK_ROW(n[r] = n[r] - 6755399441055744.)
    DL_STK_ROW(x[r] = fma(n[r], act == 0 ? -6.93147180369123816490e-01 : -3.46573590184561908245e-01, x[r]))
    DL_STK_ROW(x[r] = fma(n[r], act == 0 ? -1.90821492927058770002e-10 : -9.54107464635293850010e-11, x[r]))
    // (degree 11, interpolated at the Chebyshev nodes of |x| <= 1.0001 ln 2 / 2 in 60-digit arithmetic: 4.2e-18 off e^x before rounding -- two steps fewer than the
    //  Taylor polynomial of degree 13 for the same 2.1e-16 after it)
    DL_STK_ROW(p[r] = fma((act == 0 ? 2.5110049204818658e-08 : 5.142538077146861e-05), x[r], (act == 0 ? 2.763265472252779e-07 : 0.00028295838435868457)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 2.755724088722987e-06 : 0.0014109307334261693)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 2.4801485441561313e-05 : 0.006349180273039696)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 0.00019841269890076403 : 0.025396825459297796)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 0.0013888888952352863 : 0.08888888929505832)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 0.008333333333319589 : 0.26666666666622685)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 0.04166666666648795 : 0.6666666666638073)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 0.1666666666666668 : 1.3333333333333344)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 0.5000000000000019 : 2.0000000000000075)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 1.0 : 2.0)))
    DL_STK_ROW(p[r] = fma(p[r], x[r], (act == 0 ? 1.0 : 1.0)))
    DL_STK_ROW(p[r] = __hiloint2double(__double2hiint(p[r]) + (ni[r] << 20), __double2loint(p[r])))
    DL_STK_ROW(p[r] = p[r] + 1.)
    // 1 / p: v_rcp_f64 is good to 2^-24.4; ONE cubic step y (1 + e + e^2), e = 1 - p y, takes it to 2^-53 -- what two Newton steps do with one instruction more
    // (tools/probes/rcp_probe.hip, profiles/r06p_rcp_probe.txt)
    DL_STK_ROW(x[r] = __builtin_amdgcn_rcp(p[r]))
    DL_STK_ROW(n[r] = fma(-p[r], x[r], 1.))
    DL_STK_ROW(n[r] = fma(n[r], n[r], n[r]))
    DL_STK_ROW(x[r] = fma(x[r], n[r], x[r]))
    if (act == 0) DL_STK_ROW(v[r] = v[r] * x[r])
    else { DL_STK_ROW(x[r] = fma(x[r], -2., 1.)) DL_STK_ROW(v[r] = fma(0., v[r], x[r])) }      // (+ 0 v: the clamp drops a NaN input, this hands it on -- as dl_activation does)
#undef DL_STK_ROW
}

#define DL_STK_WGBAR asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory")

// barrier among the NW waves that run the networks.  NW = 8: the workgroup barrier.  NW = 4 (dl_emu_stacked_ov.h: the other half of the workgroup streams a feature GEMM
// meanwhile, s_barrier would stop it): a monotonic counter in LDS -- a wave arrives (its LDS traffic complete) with one ds_add, then polls until all four have.  LDS
// operations of a CU execute in arrival order: what a wave wrote before its add is there for whoever sees the count.
template <int NW>
__device__ __forceinline__ void dl_stk_sync(unsigned* ctr, unsigned& target, int lane) {
    if (NW == 8) { DL_STK_WGBAR; return; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    target += NW;
    while ((int)(__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// where the activations of the networks of a batch live: network j at act + j * act_net (row stride act_ld), the LAST hidden layer at dst + j * dst_net (row stride dst_ld:
// the batch's basis record).  dl_emulated_stacked_kernel: eight activation buffers [16][tld] beside the record; dl_emulated_stacked_ov_kernel: in place in the record
// (act = dst, act_net = dst_net = H, act_ld = dst_ld = bld).  Strides = 2 mod 32: the A-operand reads of the 16 points fall on distinct banks.
struct DlStkActs { double* act; int act_net, act_ld; double* dst; int dst_net, dst_ld; };

// Layers [l0, l1) of the n_net networks of a batch, LAYER BY LAYER on NW waves (sw: this wave among them; n_layers: hidden layers of a network).  A task = one output tile
// (16 units) of one network; the n_net * tiles tasks of a layer are dealt to the waves in contiguous runs (a wave's tasks mostly share a network, i.e. an A operand, which is
// then read once), so the SIMDs carry the same load whatever the number of networks.  Activations are updated in place: every wave keeps the outputs of its tasks in registers
// across the sync that ends the reads of the layer (TM: tasks per wave and layer).  `wf`: the weights in fragment order (DlObsDev::Stack::wfrag, network j of the batch at
// wf + j * frag_doubles): a B-operand load is base + lane + immediate, no predicates.  The sixteen k-steps of the NEXT task (of the next layer's first task) are requested
// before the MFMAs (the activations) of the current one, into the OTHER of two register sets (round 6: with one set and a copy the copy waited for the request it had just
// issued -- a task then took the L2 round trip of its successor's weights, 3 k cycles for 1 k of MFMA time).  Per layer: MFMAs -> sync (every wave has read its inputs) ->
// bias + activation -> write -> sync; the sync after layer l1 - 1 is the caller's.
template <int NW, int TM>
__device__ __forceinline__ void dl_stk_layers(const int32_t* widths, int n_layers, int l0, int l1, int act, const double* __restrict__ wf, int frag_doubles, int n_net,
                                              const double* in0, int ld0, const DlStkActs& ab, int sw, int lane, unsigned* ctr, unsigned& target,
                                              unsigned long long* fst = nullptr, int* fidx = nullptr) {
    // fst / fidx: DL_STK_STAMPS diagnostics of one lane (null in production): (0x80 + 4 layer + k) << 56 | s_memtime after the MFMAs (k = 0), the sync (1), the activations (2), the sync (3)
#ifdef DL_STK_FINE_STAMPS
#define DL_STK_FSTAMP(k) { if (fst != nullptr) { if (*fidx < 63) fst[*fidx] = ((unsigned long long)(0x80 + 4 * layer + (k)) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); ++*fidx; } }
#else
#define DL_STK_FSTAMP(k)
#endif
    const int col = lane & 15, g = lane >> 4;
    wf += lane;
    size_t loff = 0;                        // offset of the layer in a network's fragment-ordered weights
    for (int l = 0; l < l0; ++l) loff += (size_t)((widths[l + 1] + 15) / 16) * (size_t)(((widths[l] + 3) / 4) * 64 + 16);
    double bwa[16], bwb[16];                // the two weight sets of the 16-step form
    bool have = false, in_b = false;        // the weights of this wave's first task of the coming layer are in flight (in bwb: in_b)
    for (int layer = l0; layer < l1; ++layer) {
        const int nin = widths[layer], nout = widths[layer + 1];
        const int ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16;
        const bool more = layer + 1 < l1, last = layer == n_layers - 1;
        const int total = n_net * tiles, per = (total + NW - 1) / NW;
        const int t_begin = sw * per < total ? sw * per : total, t_end = t_begin + per < total ? t_begin + per : total;
        const size_t lnext = loff + (size_t)tiles * ksteps * 64 + 16 * tiles;
        const int ntiles = more ? (widths[layer + 2] + 15) / 16 : 1, ntotal = n_net * ntiles, nper = (ntotal + NW - 1) / NW;
        const int nt_begin = sw * nper < ntotal ? sw * nper : ntotal;
        const bool next16 = more && (nout + 3) / 4 == 16 && nt_begin < ntotal;
        const double* wnext = wf + (size_t)(nt_begin / ntiles) * frag_doubles + lnext + (size_t)(nt_begin % ntiles) * 1024;
        dl_stk_double4 res[TM];
        if (ksteps == 16) {
            if (have && in_b) {
#pragma unroll
                for (int u = 0; u < 16; ++u) bwa[u] = bwb[u];         // (requested a whole activation phase ago)
            }
            if (!have && t_begin < t_end) {
                const double* wt = wf + (size_t)(t_begin / tiles) * frag_doubles + loff + (size_t)(t_begin % tiles) * 1024;
#pragma unroll
                for (int u = 0; u < 16; ++u) bwa[u] = wt[u * 64];
            }
            double av[16];
            int jprev = -1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int task = t_begin + i;
                if (task >= t_end) break;
                const int jn = task / tiles;
                if (jn != jprev) {      // (the tiles of a network share its A operand)
                    const double* ap = (layer == 0 ? in0 + col * ld0 : ab.act + (size_t)jn * ab.act_net + (size_t)col * ab.act_ld) + g;
#pragma unroll
                    for (int u = 0; u < 16; ++u) av[u] = ap[4 * u];
                    jprev = jn;
                }
                double (&bc)[16] = (i & 1) ? bwb : bwa;
                double (&bn)[16] = (i & 1) ? bwa : bwb;
                if (task + 1 < t_end) {
                    const double* wt = wf + (size_t)((task + 1) / tiles) * frag_doubles + loff + (size_t)((task + 1) % tiles) * 1024;
#pragma unroll
                    for (int u = 0; u < 16; ++u) bn[u] = wt[u * 64];
                } else if (next16) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) bn[u] = wnext[u * 64];
                }
                dl_stk_double4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};   // two chains: a dependent MFMA waits for its predecessor
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bc[u], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bc[u + 1], acc2, 0, 0, 0);
                }
                res[i] = acc + acc2;
            }
            in_b = ((t_end - t_begin) & 1) != 0;          // the set the last task requested into
            if (t_begin >= t_end && next16) {             // no task in this layer, one in the next: nothing above requested its weights
#pragma unroll
                for (int u = 0; u < 16; ++u) bwa[u] = wnext[u * 64];
                in_b = false;
            }
        } else if (ksteps <= 4) {
            // a short layer (the first one: n_x inputs): the weights of ALL the wave's tasks are requested before the first MFMA -- one round trip, not one per task
            // (task by task the six tasks of a wave took 33 k cycles for twelve MFMAs, every load cold)
            double ws[TM][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int task = t_begin + i < t_end ? t_begin + i : (t_end > 0 ? t_end - 1 : 0);      // (clamped: the load count is the same on every path)
                const double* wt = wf + (size_t)(task / tiles) * frag_doubles + loff + (size_t)(task % tiles) * ksteps * 64;
#pragma unroll
                for (int u = 0; u < 4; ++u) ws[i][u] = wt[(u < ksteps ? u : ksteps - 1) * 64];
            }
            double av[4];
            int jprev = -1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int task = t_begin + i;
                if (task >= t_end) break;
                const int jn = task / tiles;
                if (layer != 0 ? jn != jprev : jprev < 0) {      // (layer 0: every network reads the same scaled inputs)
                    const double* ap = (layer == 0 ? in0 + col * ld0 : ab.act + (size_t)jn * ab.act_net + (size_t)col * ab.act_ld) + g;
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = ap[4 * (u < ksteps ? u : ksteps - 1)];
                    jprev = jn;
                }
                dl_stk_double4 acc = {0., 0., 0., 0.};
#pragma unroll
                for (int u = 0; u < 4; ++u) if (u < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], ws[i][u], acc, 0, 0, 0);
                res[i] = acc;
            }
            if (next16) {
#pragma unroll
                for (int u = 0; u < 16; ++u) bwa[u] = wnext[u * 64];
                in_b = false;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int task = t_begin + i;
                if (task >= t_end) break;
                const int jn = task / tiles, t = task - jn * tiles;
                const double* ap = (layer == 0 ? in0 + col * ld0 : ab.act + (size_t)jn * ab.act_net + (size_t)col * ab.act_ld) + g;
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
#pragma unroll
                for (int u = 0; u < 16; ++u) bwa[u] = wnext[u * 64];
                in_b = false;
            }
        }
        have = next16;
        DL_STK_FSTAMP(0)
        // every wave has read what it needs of this layer's inputs (layer 0 reads the scaled inputs, which nobody overwrites; the weight requests stay in flight)
        if (layer > 0) dl_stk_sync<NW>(ctr, target, lane);
        DL_STK_FSTAMP(1)
        // bias, activation (four values side by side: dl_stk_act_rows), write
        const int nlim = last ? nout : (nout + 3) & ~3;  // units beyond the layer (zero weights and bias) are written too when they pad the next layer's k-steps: act(0) = 0
        const int hld = last ? ab.dst_ld : ab.act_ld;
        const double* bias = (wf - lane) + loff + (size_t)tiles * ksteps * 64 + col;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int task = t_begin + i;
            if (task >= t_end) break;
            const int jn = task / tiles, t = task - jn * tiles;
            double* hb = (last ? ab.dst + (size_t)jn * ab.dst_net : ab.act + (size_t)jn * ab.act_net) + 16 * t + col;
            const double b = bias[(size_t)jn * frag_doubles + 16 * t];
            double vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = res[i][r] + b;          // accumulator register r = out[point g + 4 r][oc]
            dl_stk_act_rows<4>(act, vv);
            if (16 * t + col < nlim) {
#pragma unroll
                for (int r = 0; r < 4; ++r) hb[(size_t)(g + 4 * r) * hld] = vv[r];
            }
        }
        DL_STK_FSTAMP(2)
        if (more) dl_stk_sync<NW>(ctr, target, lane);    // the layer's outputs are in place
        DL_STK_FSTAMP(3)
        loff = lnext;
    }
#undef DL_STK_FSTAMP
}

// the constant basis function and the zero padding of the last operand step of a record with K basis functions, by the `nthr` threads t = 0 .. nthr - 1
__device__ __forceinline__ void dl_stko_fill(double* rec, int bld, int K, int t, int nthr) {
    const int nq = (K + 7) / 8, w = 8 * nq - (K - 1);
    for (int idx = t; idx < DL_STK_PTS * w; idx += nthr) {
        const int pt = idx / w, c = K - 1 + (idx - pt * w);
        rec[(size_t)pt * bld + c] = c == K - 1 ? 1. : 0.;
    }
}

// theta -> residual rows / finalize in the tail, as dl_emulated_stacked_kernel (same arguments, same operands); TM: output-tile tasks of a layer per wave when four waves
// run the networks (ceil(max_net x tiles / 4): 6 at the size of BASELINE configs[2], at most DL_STKO_TMAX).  stamps: DL_STK_STAMPS diagnostics, 64 slots of wave 0 then
// 64 of wave 4 per workgroup: slots 0-2 entry / inputs / monomial rows (wave 0), then (code << 56 | s_memtime) in program order -- 0x10 + gi: batch of group gi in place,
// 0x20 + gi / 0x30 + gi: slot 1 own work done / barrier passed, 0x40 + gi / 0x50 + gi: slot 2, 0x60 + gi: feature GEMM on all waves done, 0x70: tail done;
// last slot: HW_ID register (SIMD of the wave)
template <int RMAX, int TM>
__global__ __launch_bounds__(512) void dl_emulated_stacked_ov_kernel(const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ gfrag, const DlObsDev o,
                                                                     double* __restrict__ out, int64_t ldo, int accumulate, int steps_per_block, unsigned long long* stamps, const DlStkTail tl,
                                                                     int mode) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned sync_ctr[2];           // arrival counters of the halves A and B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * DL_STK_PTS;
    const int R = 1 + o.n_var;
    const int tld = dl_stk_tld(o), bld = dl_stk_bld(o);
    unsigned long long* st0 = stamps != nullptr && blockIdx.y == 0 ? stamps + (size_t)blockIdx.x * 128 : nullptr;
    unsigned long long* st = st0 != nullptr && (wave == 0 || wave == 4) && lane == 0 ? st0 + (wave == 4 ? 64 : 0) : nullptr;
    int sidx = 3;
#define DL_STKO_STAMP(code) { if (st != nullptr && sidx < 63) st[sidx] = ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); ++sidx; }
    if (st0 != nullptr && tid == 0) st0[0] = __builtin_amdgcn_s_memtime();
    if (tid == 0) { sync_ctr[0] = 0u; sync_ctr[1] = 0u; }
    const DlStkLds s = dl_stk_carve(lds);
    double* recs = s.work;                                        // [2][16][bld] basis records: current / next batch
    const DlObsDev::Engine& e0 = o.eng[0];
    const int n_layers = e0.n_layers, H = e0.widths[n_layers];
    dl_stk_prologue(o, theta, n_params, B, p0, tid, lds, recs, tld, R, st0);
    double outv[4][RMAX];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int u = 0; u < RMAX; ++u) outv[rr][u] = 0.;
    const int jb = blockIdx.y * 8 + wave;
    const dl_fg_double2* gcol = reinterpret_cast<const dl_fg_double2*>(gfrag) + (size_t)jb * steps_per_block * 64 + lane;
    const bool half_a = wave < 4;
    const int sw = wave & 3;
    unsigned target_a = 0u, target_b = 0u;
    int cur = 0, tb_cur = -1, te_cur = -1;
    bool next_ready = false;
    const int ls = (n_layers + 1) / 2;           // layers of slot 1 (layer 0 is the short one: n_x inputs)
    for (int gi = 0; gi < o.stk.n_groups; ++gi) {
        const double* rec = o.stk.table + (size_t)gi * DL_STK_REC;
        const int tb = (int)rec[0], te = (int)rec[1], m0 = (int)rec[2], m1 = (int)rec[3], kq = (int)rec[7];
        const int K = (te - tb) * H + 1, nq = (K + 7) / 8;
        if ((mode & 4) && (tb != tb_cur || te != te_cur)) {
            // split mode: no overlap with the feature GEMM; the two halves run HALF of the batch's networks each, every half on its own counter -- the two waves of a SIMD
            // drift apart (one in its MFMAs while the other is in its activations) instead of stalling together
            DL_STK_WGBAR;
            double* rc = recs;
            const int n = te - tb, na = (n + 1) / 2;
            if (half_a) {
                const DlStkActs ab = {rc, H, bld, rc, H, bld};
                if (na > 0) dl_stk_layers<4, TM>(e0.widths, n_layers, 0, n_layers, e0.act, o.stk.wfrag + (size_t)tb * o.stk.frag_doubles, o.stk.frag_doubles, na, s.xs, DL_STK_XLD, ab, sw, lane,
                                                 &sync_ctr[0], target_a);
            } else {
                const DlStkActs ab = {rc + (size_t)na * H, H, bld, rc + (size_t)na * H, H, bld};
                if (n - na > 0) dl_stk_layers<4, TM>(e0.widths, n_layers, 0, n_layers, e0.act, o.stk.wfrag + (size_t)(tb + na) * o.stk.frag_doubles, o.stk.frag_doubles, n - na, s.xs, DL_STK_XLD,
                                                     ab, sw, lane, &sync_ctr[1], target_b);
            }
            dl_stko_fill(rc, bld, K, tid, 512);
            DL_STK_WGBAR;
            tb_cur = tb; te_cur = te; cur = 0; next_ready = true;     // (next_ready: no overlapped slots in this mode)
        }
        if (tb != tb_cur || te != te_cur) {
            if (next_ready) { cur ^= 1; next_ready = false; }
            else {
                // the first batch: nothing to hide it under -- all eight waves, workgroup barriers
                DL_STK_WGBAR;         // the monomial rows are complete, the record is free
                unsigned dummy = 0u;
                const DlStkActs ab = {recs + (size_t)cur * DL_STK_PTS * bld, H, bld, recs + (size_t)cur * DL_STK_PTS * bld, H, bld};
                if (te > tb) dl_stk_layers<8, (TM + 1) / 2>(e0.widths, n_layers, 0, n_layers, e0.act, o.stk.wfrag + (size_t)tb * o.stk.frag_doubles, o.stk.frag_doubles, te - tb, s.xs, DL_STK_XLD,
                                                            ab, wave, lane, nullptr, dummy);
                dl_stko_fill(recs + (size_t)cur * DL_STK_PTS * bld, bld, K, tid, 512);
                DL_STK_WGBAR;
            }
            tb_cur = tb; te_cur = te;
        }
        DL_STKO_STAMP(0x10 + gi)
        const double* arow = recs + (size_t)cur * DL_STK_PTS * bld + (size_t)col * bld + 2 * g;
        const dl_fg_double2* gw = gcol + (size_t)kq * 64;
        // the next batch: the first later group on other networks
        int tbn = -1, ten = -1;
        for (int gj = gi + 1; gj < o.stk.n_groups; ++gj) {
            const double* rj = o.stk.table + (size_t)gj * DL_STK_REC;
            if ((int)rj[0] != tb || (int)rj[1] != te) { tbn = (int)rj[0]; ten = (int)rj[1]; break; }
        }
        if (!next_ready && tbn >= 0) {
            double* rn = recs + (size_t)(cur ^ 1) * DL_STK_PTS * bld;
            const int Kn = (ten - tbn) * H + 1;
            DL_STK_WGBAR;             // every wave is past the other record (the feature GEMMs of the batch before this one)
            if (ten == tbn) {          // a batch without networks (constant tables): its record is the constant basis function
                dl_stko_fill(rn, bld, Kn, tid, 512);
                DL_STK_WGBAR;
                dl_stk_group<RMAX>(m1 - m0, arow, gw, nq, s.mono + m0, R, g, outv);
                DL_STKO_STAMP(0x60 + gi)
            } else {
                const double* wfn = o.stk.wfrag + (size_t)tbn * o.stk.frag_doubles;
                const DlStkActs abn = {rn, H, bld, rn, H, bld};           // activations in place in the next record
                if (half_a) dl_stk_group<RMAX>(m1 - m0, arow, gw, nq, s.mono + m0, R, g, outv);
                else {
                    if (mode & 1) __builtin_amdgcn_s_setprio(2);
                    dl_stk_layers<4, TM>(e0.widths, n_layers, 0, ls, e0.act, wfn, o.stk.frag_doubles, ten - tbn, s.xs, DL_STK_XLD, abn, sw, lane, &sync_ctr[1], target_b, gi == 0 ? st : nullptr, &sidx);
                    dl_stko_fill(rn, bld, Kn, tid - 256, 256);
                    if (mode & 1) __builtin_amdgcn_s_setprio(0);
                }
                DL_STKO_STAMP(0x20 + gi)
                DL_STK_WGBAR;
                DL_STKO_STAMP(0x30 + gi)
                if (!half_a) dl_stk_group<RMAX>(m1 - m0, arow, gw, nq, s.mono + m0, R, g, outv);
                else if (ls < n_layers) {
                    if (mode & 1) __builtin_amdgcn_s_setprio(2);
                    dl_stk_layers<4, TM>(e0.widths, n_layers, ls, n_layers, e0.act, wfn, o.stk.frag_doubles, ten - tbn, s.xs, DL_STK_XLD, abn, sw, lane, &sync_ctr[0], target_a, gi == 0 ? st : nullptr, &sidx);
                    if (mode & 1) __builtin_amdgcn_s_setprio(0);
                }
                DL_STKO_STAMP(0x40 + gi)
                DL_STK_WGBAR;
                DL_STKO_STAMP(0x50 + gi)
            }
            next_ready = true;
        } else {
            dl_stk_group<RMAX>(m1 - m0, arow, gw, nq, s.mono + m0, R, g, outv);
            DL_STKO_STAMP(0x60 + gi)
        }
    }
    if (!tl.enabled) dl_stk_store_rows<RMAX>(outv, R, out, ldo, accumulate, B, p0, jb, col, g);
    else {
        __shared__ double lp_lds[DL_STK_PTS];
        __shared__ int nan_lds[DL_STK_PTS];
        dl_stk_finalize_tail<RMAX>(tl, outv, R, recs, theta, n_params, B, p0, tid, wave, lane, col, g, lp_lds, nan_lds);
    }
    DL_STKO_STAMP(0x70)
    if (st != nullptr) st[63] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
#undef DL_STKO_STAMP
}
#endif
