// Stacked table engine, round 6: the step as TWO launches -- every network as a WAVE-PRIVATE chain, then the feature GEMMs (emulators/conversion.py:44-98,
// full_shape.py:1182-1186, 1416-1443; same arithmetic and operands as dl_emu_stacked.h).
//
//   What the round-6 measurements of dl_emulated_stacked_kernel say (docs/EXPERIMENTS.md): its network phases (3 x 41 us of 244) are chains of latencies PER WAVE -- ~5.8 k
//   cycles per output-tile task whoever shares the SIMD -- because a layer of a group's six networks is dealt tile by tile to eight waves that meet at two barriers per
//   layer; no single cost dominates (activations 27 %, MFMAs 14 %, weight round trips 13 %, syncs 7 %, the rest of the chain 40 %), and running the phase under the
//   feature GEMM of another batch does not work on this chip (a vector-instruction chain beside an MFMA stream gets one issue slot per MFMA).  So the networks leave the
//   workgroup structure of the feature GEMM altogether:
//
//   (A) dl_stk_chain_kernel: one WAVE per (network, 16-point tile) -- 18 x 256 = 4608 independent chains at the size of BASELINE configs[2].  A wave walks all layers of
//       its network alone: the 16 x 64 activations go through 8 KB of wave-private LDS only to change from the accumulator layout to the A-operand layout (no barrier: a
//       wave's LDS operations execute in order), a layer goes output tile by output tile -- sixteen MFMAs in two chains, then the tile's four activations per lane side by
//       side (dl_stk_act_rows<4>: 156 registers, three waves per SIMD) -- and the next tile's weights are requested a tile ahead into the other of two register sets.  Networks are dealt
//       network-major, so the four waves of a workgroup -- and the workgroups that run together -- stream the same weights.  The last hidden layers go to memory,
//       basis [points][n_networks x H] (38 MB per 4096 points; read back at once: Infinity Cache).
//   (B) dl_emulated_stacked_gemm_kernel: the feature GEMMs of dl_emulated_stacked_kernel (same operand stream, same epilogues, same tail) with the basis records of a batch
//       arriving by LDS-DMA into one of two record buffers while the previous batch multiplies.
#pragma once
#include "dl_emu_stacked_ov.h"

#define DL_STKS_LD 66            // row stride (doubles) of a wave's activation tile [16][64]: 2 mod 32
#define DL_STKS_TILES 4          // output tiles per layer: hidden widths <= 64

// shapes the split form takes: hidden widths within four output tiles, inputs within four k-steps, two basis records within the LDS
static inline bool dl_stks_ok(const DlObsDev& o) {
    if (!dl_stk_feature_ok(o)) return false;
    const DlObsDev::Engine& e = o.eng[0];
    if (o.n_x > 16 || e.n_layers < 1) return false;
    for (int l = 1; l <= e.n_layers; ++l) if (e.widths[l] > 16 * DL_STKS_TILES) return false;
    if (e.widths[e.n_layers] % 2 != 0) return false;
    return (dl_stko_fixed_doubles() + dl_stko_work_doubles(o)) * sizeof(double) + DL_STK_STATIC_LDS <= 160 * 1024;
}

#if defined(__HIPCC__)
// (A) one wave = one (network, 16-point tile) chain.  basis_out [B][ldk]: the last hidden layer of network t of the observable's stack at columns [t H, (t + 1) H).
template <int ACT>      // the activation of the hidden layers (0 silu, 1 relu, 2 tanh) at compile time: at run time a tile carried three copies of the activation rows behind scalar branches
__global__ __launch_bounds__(256, 3) void dl_stk_chain_kernel(const double* __restrict__ theta, int n_params, int64_t B, const DlObsDev o, double* __restrict__ basis_out, int64_t ldk,
                                                            int n_pt_tiles) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chain = blockIdx.x * 4 + wave;
    if (chain >= o.stk.n_trunks * n_pt_tiles) return;          // (no barrier in this kernel: a wave may leave alone)
    const int net = chain / n_pt_tiles, tile = chain - net * n_pt_tiles;   // network-major: the waves of a workgroup share their weights
    const int col = lane & 15, g = lane >> 4;
    const DlObsDev::Engine& e = o.eng[0];
    const int n_layers = e.n_layers, H = e.widths[n_layers];
    constexpr int act = ACT;
    const double* __restrict__ wf = o.stk.wfrag + (size_t)net * o.stk.frag_doubles + lane;
    double* hb = lds + (size_t)wave * DL_STK_PTS * DL_STKS_LD;
    const int64_t p0 = (int64_t)tile * DL_STK_PTS;
    // A operand of the first layer: the scaled inputs (conversion.py:75-77) of point p0 + col, elements 4 u + g
    double a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = 0.;
    {
        const int64_t b = p0 + col < B ? p0 + col : B - 1;
        const double* th = theta + (size_t)b * n_params;
        for (int i = 0; i < o.n_x; ++i) {      // (uniform index: scalar loads of the descriptor; the lane keeps the elements of its own k-group)
            const double v = (dl_get(o.x_in[i], th) - e.xlo[i]) * e.xinv[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = (i == 4 * u + g) ? v : a[u];
        }
    }
    size_t loff = 0;                        // offset of the layer in the network's fragment-ordered weights
    double wa[16], wb[16];                  // the two weight sets of the 16-step form: tile t multiplies from set t & 1
    for (int layer = 0; layer < n_layers; ++layer) {
        const int nin = e.widths[layer], nout = e.widths[layer + 1];
        const int ksteps = (nin + 3) / 4, tiles = (nout + 15) / 16;
        const bool last = layer == n_layers - 1;
        const double* wl = wf + loff;
        const double* bl = (wl - lane) + (size_t)tiles * ksteps * 64 + col;      // the layer's biases
        const size_t lnext = loff + (size_t)tiles * ksteps * 64 + 16 * tiles;
        if (ksteps == 16 && layer == 0) {      // (after the first layer set A holds tile 0 already: requested before the last activations of the layer before)
#pragma unroll
            for (int u = 0; u < 16; ++u) wa[u] = wl[u * 64];
        }
        // tile by tile: MFMAs, bias + activation of the tile's four values per lane (four chains side by side: three waves share the SIMD), write.  Accumulator register r of
        // tile t = out[point g + 4 r][16 t + col]
#pragma unroll
        for (int t = 0; t < DL_STKS_TILES; ++t) {
            if (t < tiles) {
                const double bias = bl[16 * t];
                dl_stk_double4 c0 = {bias, bias, bias, bias};      // ONE accumulator chain, started from the bias: a dependent v_mfma_f64_16x16x4 issues as fast as an independent one
                                                                    // (64 cycles either way, profiles/mfma_f64_probe.txt) -- a second chain was four zeros and four additions per tile
                if (ksteps == 16) {
                    double (&wc)[16] = (t & 1) ? wb : wa;
                    double (&wn)[16] = (t & 1) ? wa : wb;
                    if (t + 1 < tiles) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) wn[u] = wl[(size_t)(t + 1) * 1024 + u * 64];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], wc[u], c0, 0, 0, 0);
                } else if (ksteps <= 4) {      // short layers (the first one: n_x inputs)
                    double w4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) w4[u] = wl[(size_t)t * ksteps * 64 + (u < ksteps ? u : ksteps - 1) * 64];
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (u < ksteps) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], w4[u], c0, 0, 0, 0);
                } else {
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (u < ksteps) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], wl[(size_t)t * ksteps * 64 + u * 64], c0, 0, 0, 0);
                }
                // tile 0 of the next layer into set A (the layer after starts with t = 0), behind the last MFMAs of this one (which may read set A) and ahead of its activations
                if (t + 1 == tiles && !last && (nout + 3) / 4 == 16) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) wa[u] = wf[lnext + u * 64];
                }
                double vv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] = c0[r];
                dl_stk_act_rows<4>(act, vv);
                const int oc = 16 * t + col;
                if (!last) {
                    // units beyond the layer (zero weights and bias) are written too: they pad the next layer's k-steps, act(0) = 0 for silu / relu / tanh
#pragma unroll
                    for (int r = 0; r < 4; ++r) hb[(g + 4 * r) * DL_STKS_LD + oc] = vv[r];
                } else if (oc < nout) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (p0 + g + 4 * r < B) basis_out[(size_t)(p0 + g + 4 * r) * ldk + (size_t)net * H + oc] = vv[r];
                }
            }
        }
        if (!last) {
            // accumulator layout -> A-operand layout through the wave's own tile (LDS operations of a wave execute in order: no barrier)
            const int ks2 = (nout + 3) / 4;
            const double* ap = hb + col * DL_STKS_LD + g;
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = u < ks2 ? ap[4 * u] : 0.;
        }
        loff = lnext;
    }
}

// a batch's basis records basis [B][ldk] (columns [tb H, te H)) -> record `rec` [16][bld] in LDS, by LDS-DMA where a row is whole 1 KB pieces (wave w takes pieces w, w + 8, ...),
// else by loads and ds_writes; asynchronous in the first case: the caller waits (vmcnt) before the barrier that publishes the record.  Waves w0 .. w0 + nw - 1 share the work
// (the others form the monomial rows / the log-priors meanwhile)
__device__ __forceinline__ void dl_stks_fetch(const double* __restrict__ basis_in, int64_t ldk, int64_t B, int64_t p0, int tb, int te, int H, double* rec, int bld, int tid, int wave, int lane, int w0 = 0, int nw = 8) {
    const int row_doubles = (te - tb) * H;
    if (row_doubles <= 0 || wave < w0 || wave >= w0 + nw) return;
    if (row_doubles % 128 == 0) {
        const int segs = row_doubles / 128;
        for (int q = wave - w0; q < DL_STK_PTS * segs; q += nw) {
            const int r = q / segs, sg = q - r * segs;
            const int64_t b = p0 + r < B ? p0 + r : B - 1;
            const double* src = basis_in + (size_t)b * ldk + (size_t)tb * H + (size_t)sg * 128 + 2 * lane;
            double* dst = rec + (size_t)r * bld + (size_t)sg * 128;       // (wave-uniform: the piece of lane l lands at dst + 2 l)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    } else {
        const int c2n = row_doubles / 2;      // (H even: dl_stks_ok)
        for (int idx = tid - 64 * w0; idx < DL_STK_PTS * c2n; idx += 64 * nw) {
            const int r = idx / c2n, c2 = idx - r * c2n;
            const int64_t b = p0 + r < B ? p0 + r : B - 1;
            *reinterpret_cast<dl_fg_double2*>(rec + (size_t)r * bld + 2 * c2) = *reinterpret_cast<const dl_fg_double2*>(basis_in + (size_t)b * ldk + (size_t)tb * H + 2 * c2);
        }
    }
}

// (B) theta + basis records -> residual rows / finalize in the tail (arguments of dl_emulated_stacked_kernel + the basis records of dl_stk_chain_kernel)
template <int RMAX>
__global__ __launch_bounds__(512) void dl_emulated_stacked_gemm_kernel(const double* __restrict__ theta, int n_params, int64_t B, const double* __restrict__ gfrag, const DlObsDev o,
                                                                       double* __restrict__ out, int64_t ldo, int accumulate, int steps_per_block, unsigned long long* stamps, const DlStkTail tl,
                                                                       const double* __restrict__ basis_in, int64_t ldk) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t p0 = (int64_t)blockIdx.x * DL_STK_PTS;
    // theta, n_params, B arrive in SGPRs with the wave (kernel-argument preload): the theta rows of the 16 points are requested before the first access to the descriptor `o`
    // (thread t: point t / 32, column t % 32; dl_stk_prologue), and every 64-byte line of the kernel-argument segment is touched by one batch of scalar loads -- the fields
    // are read where they are first needed, and each first touch of a line was a miss of the scalar cache inside a dependent chain (as in dl_emulated_feature_gram_kernel)
    const bool th_early = n_params <= 32;
    double th_val = 0.;
    if (th_early) {
        const int pt = tid >> 5, j = tid & 31;
        const int64_t b = p0 + pt < B ? p0 + pt : B - 1;
        th_val = theta[(size_t)b * n_params + (j < n_params ? j : 0)];
    }
    {
        constexpr int n_lines = (int)((32 + sizeof(DlObsDev) + 32 + sizeof(DlStkTail) + 16) / 64);      // (whole lines inside the segment)
        const __attribute__((address_space(4))) int* kargs = (const __attribute__((address_space(4))) int*)__builtin_amdgcn_kernarg_segment_ptr();
        int touched = 0;
#pragma unroll
        for (int l = 0; l < n_lines; ++l) touched |= kargs[16 * l];
        asm volatile("; kernel arguments touched: %0" :: "s"(touched));
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int R = 1 + o.n_var;
    const int tld = dl_stk_tld(o), bld = dl_stk_bld(o);
    // DL_STK_STAMPS (null in production): the slots of dl_emulated_stacked_kernel -- 0 entry, 1 inputs, 2 monomial rows, 3 + 2 gi: the group's record in place, 4 + 2 gi: its feature GEMM done, 30: tail done
    // (128 slots per workgroup: 32 + 12 w + gi / + 6 + gi: wave w at the start / the end of the feature GEMM of group gi < 6 -- the skew between the eight column blocks)
    unsigned long long* st = stamps != nullptr && blockIdx.y == 0 ? stamps + (size_t)blockIdx.x * 128 : nullptr;
#define DL_STKS_STAMP(slot) if (st != nullptr && tid == 0) st[slot] = __builtin_amdgcn_s_memtime();
#define DL_STKS_WSTAMP(off) if (st != nullptr && lane == 0 && gi < 6) st[32 + 12 * wave + (off) + gi] = __builtin_amdgcn_s_memtime();
    DL_STKS_STAMP(0)
    const DlStkLds s = dl_stk_carve(lds);
    double* recs = s.work;                                        // [2][16][bld] basis records: batch b in record b & 1
    const int H = o.eng[0].widths[o.eng[0].n_layers];
    // the records of the first two batches are requested by the six waves that form no monomial rows, beside them, when the work area is free (no scalar engine uses it as
    // scratch: the standard prior basis): the group table behind the request is a cold global round trip of its own -- asked for before the prologue it sat in front of the
    // inputs, on the path to the first feature GEMM (stamps: 60 of the entry's 206 hundred cycles)
    const bool early = o.eng[1].type != 0 && o.eng[2].type != 0;
    int requested = 0;                                            // batches whose record has been asked for
    auto request_first_two = [&](int w0, int nw) {
        int tb_p = -1, te_p = -1;
        for (int gj = 0; gj < o.stk.n_groups && requested < 2; ++gj) {
            const double* rj = o.stk.table + (size_t)gj * DL_STK_REC;
            if ((int)rj[0] != tb_p || (int)rj[1] != te_p) {
                tb_p = (int)rj[0]; te_p = (int)rj[1];
                dl_stks_fetch(basis_in, ldk, B, p0, tb_p, te_p, H, recs + (size_t)(requested & 1) * DL_STK_PTS * bld, bld, tid, wave, lane, w0, nw);
                ++requested;
            }
        }
    };
    constexpr int mono_waves = DL_STK_PTS * DL_STK_ROWS / 64;     // (waves of the monomial rows: dl_stk_prologue)
    // the log-priors of the 16 points (the finalize in the tail needs them; they depend on theta only) by the last wave, also beside the monomial rows: in the tail they were two
    // cold round trips (prior table, theta rows) of one wave between the Gram phase and the barrier in front of the solve
    __shared__ double lp_lds[DL_STK_PTS];
    __shared__ int nan_lds[DL_STK_PTS];
    auto beside = [&]() {
        if (st != nullptr && tid == 448) st[24] = __builtin_amdgcn_s_memtime();      // (slots 24 - 26: wave 7 beside the monomial rows)
        if (tl.enabled && wave == 7) dl_stk_priors(tl, theta, n_params, B, p0, lane, lp_lds, nan_lds);      // (first: its cold loads beside the others' wait for the group table)
        if (st != nullptr && tid == 448) st[25] = __builtin_amdgcn_s_memtime();
        if (early) request_first_two(mono_waves, tl.enabled ? 7 - mono_waves : 8 - mono_waves);      // (wave 7 keeps to the log-priors: with a share of the requests behind them it was the last to arrive, 50 hundred cycles after the monomial rows)
        if (st != nullptr && tid == 448) st[26] = __builtin_amdgcn_s_memtime();
    };
    dl_stk_prologue(o, theta, n_params, B, p0, tid, lds, recs, tld, R, st, th_early, th_val, beside);
    double outv[4][RMAX], cpre[RMAX];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int u = 0; u < RMAX; ++u) outv[rr][u] = 0.;
#pragma unroll
    for (int u = 0; u < RMAX; ++u) cpre[u] = 0.;
    const int jb = blockIdx.y * 8 + wave;
    const dl_fg_double2* gcol = reinterpret_cast<const dl_fg_double2*>(gfrag) + (size_t)jb * steps_per_block * 64 + lane;
    // the batches (runs of groups on the same networks), in order; record of batch b: b & 1
    DL_STK_WGBAR;
    int ibatch = -1, tb_cur = -1, te_cur = -1;
    if (!early) request_first_two(0, 8);                             // (the prologue's scratch in the work area is done with)
    for (int gi = 0; gi < o.stk.n_groups; ++gi) {
        const double* rec = o.stk.table + (size_t)gi * DL_STK_REC;
        const int tb = (int)rec[0], te = (int)rec[1], m0 = (int)rec[2], m1 = (int)rec[3], kq = (int)rec[7];
        const int K = (te - tb) * H + 1, nq = (K + 7) / 8;
        if (tb != tb_cur || te != te_cur) {
            ++ibatch; tb_cur = tb; te_cur = te;
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of the record have landed ...
            dl_stko_fill(recs + (size_t)(ibatch & 1) * DL_STK_PTS * bld, bld, K, tid, 512);
            DL_STK_WGBAR;                                                   // ... and everybody's
        }
        DL_STKS_STAMP(3 + 2 * gi) DL_STKS_WSTAMP(0)
        if (RMAX <= 6 && tl.enabled && gi + 1 == o.stk.n_groups) {      // the constant parts of the rows (cold) travel under the last feature GEMM (eight rows: no registers left for them)
#pragma unroll
            for (int u = 0; u < RMAX; ++u) cpre[u] = u < R ? tl.cst[u][wave * 16 + col] : 0.;
        }
        const double* rc = recs + (size_t)(ibatch & 1) * DL_STK_PTS * bld;
        dl_stk_group<RMAX>(m1 - m0, rc + (size_t)col * bld + 2 * g, gcol + (size_t)kq * 64, nq, s.mono + m0, R, g, outv);
        DL_STKS_STAMP(4 + 2 * gi) DL_STKS_WSTAMP(6)
        // the last group of its batch: the record is free for the batch after the next one
        bool last_of_batch = gi + 1 == o.stk.n_groups;
        int tbn = -1, ten = -1, seen = 0;                                   // the (ibatch + 2)-th batch, if any
        if (!last_of_batch) {
            const double* rj = o.stk.table + (size_t)(gi + 1) * DL_STK_REC;
            last_of_batch = (int)rj[0] != tb || (int)rj[1] != te;
        }
        if (last_of_batch) {
            int tb_p = tb, te_p = te;
            for (int gj = gi + 1; gj < o.stk.n_groups && seen < 2; ++gj) {
                const double* rj = o.stk.table + (size_t)gj * DL_STK_REC;
                if ((int)rj[0] != tb_p || (int)rj[1] != te_p) { tb_p = (int)rj[0]; te_p = (int)rj[1]; if (++seen == 2) { tbn = tb_p; ten = te_p; } }
            }
            if (tbn >= 0) {
                DL_STK_WGBAR;                                               // every wave is past this record
                dl_stks_fetch(basis_in, ldk, B, p0, tbn, ten, H, recs + (size_t)(ibatch & 1) * DL_STK_PTS * bld, bld, tid, wave, lane);
            }
        }
    }
    if (!tl.enabled) dl_stk_store_rows<RMAX>(outv, R, out, ldo, accumulate, B, p0, jb, col, g);
    else dl_stk_finalize_tail<RMAX>(tl, outv, R, recs, theta, n_params, B, p0, tid, wave, lane, col, g, lp_lds, nan_lds, RMAX <= 6 ? cpre : nullptr, true);
    DL_STKS_STAMP(30)
    if (st != nullptr && tid == 0) st[31] = __builtin_amdgcn_s_memrealtime();
#undef DL_STKS_STAMP
#undef DL_STKS_WSTAMP
}
#endif
