// chi2 GEMM: partial chi2 of the whitened residual, without ever writing the residual.
//
//   chi2_b = sum_j (sum_k A[b, k] Wt[j, k] + bias[j])^2 is ADDITIVE over the output columns j, not over k: the launch is therefore split over
//   (32-row block) x (16-column block), every workgroup runs over the FULL K and emits one partial chi2 per row and column block,
//   part[b, nt] -- M x N_pad / 16 doubles instead of the M x N_pad x (K splits) residual slabs of the split-K kernel (dl_gemm_tiled.h),
//   which the plain-likelihood finalize then sums in a fixed order (deterministic).
//
//   workgroup = DL_CG_WAVES waves; K advances in 128-wide panels: 32 A rows + 16 Wt rows x 128 k = 48 KB, fetched as full 1 KB row segments (one wave =
//   one row), staged through LDS (rows padded to 130 doubles: bank stride 4 mod 64 for the MFMA operand layout; measured, round 6: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42,
//   all of it tied to the Wt rows -- with the B operand in registers (dl_chi2_gemm_tile_bf) the counter reads 0 and the main loop takes the same time: not what bounds it;
//   profiles/r06c_chi2_lds_conflicts.txt),
//   by LDS-DMA (global_load_lds_dwordx4), triple-buffered in LDS with two panels in flight (operands come cold from the theory kernel).
//   Within a panel the 32 k-steps of v_mfma_f64_16x16x4_f64 are dealt round-robin to the 8 waves (in-workgroup split-K, reduced through LDS at
//   the end); each k-step feeds two MFMAs (rows 0-15, 16-31) that share the Wt operand.
//   The 8 column blocks that share a row block are mapped to the same XCD (blockIdx % 8), so A is fetched once per XCD L2.
#pragma once
#include "dl_scalar_prefetch.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double dl_cg_double2 __attribute__((ext_vector_type(2)));
typedef double dl_cg_double4 __attribute__((ext_vector_type(4)));

#define DL_CG_M 32
#define DL_CG_N 16
#define DL_CG_KP 128                       // panel width
#define DL_CG_LD (DL_CG_KP + 2)            // padded LDS row (doubles)
#define DL_CG_ROWS (DL_CG_M + DL_CG_N)
#ifndef DL_CG_WAVES
#define DL_CG_WAVES 16                     // waves per workgroup: 4 per SIMD hide the LDS-read -> MFMA latency of each other (8: 8.0 us main loop; 16: see DESIGN)
#endif
#define DL_CG_VPT (DL_CG_ROWS / DL_CG_WAVES)   // row segments (1 KB LDS-DMA pieces) per wave and panel
#define DL_CG_NBUF 3                       // LDS panel buffers
#define DL_CG_LDS_BYTES (DL_CG_NBUF * DL_CG_ROWS * DL_CG_LD * 8)   // (32-row tile; the 16-row tile uses the same allocation)

// Fused finalize (fin.counters != nullptr): the workgroup that completes a row block's last column block (device-scope counter) sums the partials of its 32
// points in a fixed order (deterministic whoever arrives last), adds the priors and writes loglike / logprior / status -- no separate finalize launch
// (a launch ramp plus a cold read of the partials, 4.3 us of a 34 us step).  The counter resets itself for the next launch.
struct DlChi2Fin {
    int32_t* counters;       // [row blocks], zero before the first launch; nullptr: partials only (dl_finalize_part_kernel follows)
    const double* theta;     // [M, n_params]
    const double* priors;    // [n_params, 5]
    double* loglike;         // may be null
    double* logprior;        // may be null
    int32_t* status;         // may be null
    int32_t n_params, post_mode;
    unsigned long long* stamps;   // DL_CG_STAMPS diagnostics (nullptr in production): 8 slots per workgroup, see dl_fullshape_kernel
    int32_t* ready;          // dl_step_kernel: arrival counters of the row blocks' producers, reset by the last column block of a row block (nullptr otherwise)
};

// Panels of K that hold non-zero entries of the 16 Wt rows of each column block (by value in the kernarg segment: scalar registers).  With a block-diagonal
// precision (several observables with independent covariances, SumLikelihood) Wt = L^T blockdiag(W_obs) is block diagonal: a column block of observable i
// only meets the K range of observable i -- the other panels multiply zeros and are skipped (two config-2 tracers: 10 of 19 panels per column block).
#define DL_CG_MAX_TILES 32
struct DlChi2Panels {
    uint32_t range[DL_CG_MAX_TILES];   // p_lo | p_hi << 8: panels [p_lo, p_hi) of the column block; p_hi = 0: all panels.  One dword per block: a uniform index into the
};                                     // kernel arguments is then ONE scalar load (byte arrays were two dependent vector loads ahead of the first panel request)

#define DL_CG_STAMP(slot, fn) if (fin.stamps != nullptr && tid == 0) fin.stamps[(size_t)blockIdx.x * 8 + (slot)] = fn();
// The end of a tile, shared by the two forms of the main loop: in-workgroup reduction of the k-slices of the DL_CG_WAVES waves, bias, square, sum over the 16 columns, partial chi2
// per row (and the fused finalize of the last-arriving column block when fin.counters is set).  fin_lp / fin_nan: the priors of the row block (lanes 0 .. MT - 1 of wave 0).
template <int MT, bool RESID>
__device__ __forceinline__ void dl_chi2_gemm_finish(dl_cg_double4 acc0, dl_cg_double4 acc1, double bj, double* __restrict__ part, int M, int n_tiles, const DlChi2Fin& fin,
                                                    double* __restrict__ resid, int64_t ldr, double* lds, int mb, int nt, double fin_lp, int fin_nan) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int MTILES = MT / 16;
    const int m0 = mb * MT, n0 = nt * DL_CG_N;
    DL_CG_STAMP(2, __builtin_amdgcn_s_memtime)
    // in-workgroup reduction of the 8 k-slices, then bias, square, sum over the 16 columns
    __syncthreads();
    double* red = lds;   // [waves][2 tiles][4 regs][64 lanes]
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[((wave * 2 + 0) * 4 + r) * 64 + lane] = acc0[r]; if (MTILES > 1) red[((wave * 2 + 1) * 4 + r) * 64 + lane] = acc1[r]; }
    __syncthreads();
    if (wave < 4 * MTILES) {   // wave (t, r) = (wave % MTILES, wave / MTILES): accumulator register r of row tile t (rows 16 t + (lane >> 4) + 4 r), all 16 columns
        const int t = wave % MTILES, r = wave / MTILES;
        double v = bj;
#pragma unroll
        for (int w = 0; w < DL_CG_WAVES; ++w) v += red[((w * 2 + t) * 4 + r) * 64 + lane];   // fixed order: deterministic
        if (RESID) { const int rrow = m0 + 16 * t + g + 4 * r; if (rrow < M) resid[(size_t)rrow * ldr + n0 + r16] = v; }
        double sq = v * v;
        // C layout: reg r of lane l = C[row (l >> 4) + 4 r][col l & 15]: sum the 16 lanes of a lane group
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        sq += __shfl_xor(sq, 4, 64);
        sq += __shfl_xor(sq, 8, 64);
        const int row = m0 + 16 * t + g + 4 * r;
        // agent-scope (write-through, sc1) store: performed device-wide once the wave's vmcnt drains -- no L2 write-back fence is needed before the counter
        if (r16 == 0 && row < M) __hip_atomic_store(part + (size_t)row * n_tiles + nt, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DL_CG_STAMP(3, __builtin_amdgcn_s_memtime) DL_CG_STAMP(4, __builtin_amdgcn_s_memtime) DL_CG_STAMP(5, __builtin_amdgcn_s_memtime) DL_CG_STAMP(7, __builtin_amdgcn_s_memrealtime)
    if (fin.counters == nullptr) return;
    // Hand-over without cache-maintenance fences: an agent-scope release (buffer_wbl2) walks the XCD's L2 -- 256 of them cost +20 us -- and a full
    // __threadfence() also invalidates it under the workgroups still streaming `power` and W~ (+40 us).  Instead every access to the partials and the counter
    // is itself an agent-scope (sc1) access: the stores above are performed once vmcnt has drained, the counter is a memory-side atomic, the last
    // arriver's loads bypass non-coherent lines.  One barrier, then wave 0 alone: counter round trip, 8 partial loads, outputs.
    __asm__ volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __syncthreads();
    if (wave != 0) return;
    int done = 0;
    if (lane == 0) done = __hip_atomic_fetch_add(fin.counters + mb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    done = __builtin_amdgcn_readfirstlane(done);
    if (done != n_tiles - 1) return;
    if (lane == 0) __hip_atomic_store(fin.counters + mb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream-ordered)
    if (lane == 0 && fin.ready != nullptr) __hip_atomic_store(fin.ready + mb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (every consumer of the row block is past its wait)
    const int row = m0 + lane;
    if (lane < MT && row < M) {
        double chi2 = 0.;
        for (int t = 0; t < n_tiles; ++t) chi2 += __hip_atomic_load(part + (size_t)row * n_tiles + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // fixed order
        const double inf = __builtin_huge_val();
        const double lp = fin_lp, ll = -0.5 * chi2;
        int st = 0;                    // DL_STATUS_OK
        if (fin_nan) st = 3;           // DL_STATUS_NAN_INPUT
        else if (lp == -inf) st = 1;   // DL_STATUS_OUT_OF_PRIOR
        else if (!(ll == ll) || ll == inf || ll == -inf) st = 2;   // DL_STATUS_NONFINITE
        if (fin.loglike) fin.loglike[row] = fin.post_mode ? (st == 0 ? ll + lp : -inf) : ll;
        if (fin.logprior) fin.logprior[row] = lp;
        if (fin.status) fin.status[row] = st;
    }
}

// MT: rows per workgroup, 32 or 16 (16: batches whose 32-row blocks would leave CUs without a workgroup -- 256 walkers x 16 column blocks = 128 workgroups of 32 rows,
// 256 of 16 rows, each moving 32 instead of 48 KB per panel; the partial sums of a row do not depend on the tile height)
// RESID: the residual itself is ALSO written, resid [M, ldr] (the analytic gradient needs d~ as well as chi2: dl_eval_logposterior_grad)
// The tile (row block mb, column block nt) by one workgroup of DL_CG_WAVES waves; `lds`: DL_CG_LDS_BYTES of workspace.  WAIT (dl_step_kernel: the rows of A are produced
// by other workgroups of the SAME launch): the operand rows of Wt of the first two panels are requested, then `wait()` returns once the producers have published row block mb,
// then the rows of A follow.
template <bool DO_LOAD, bool DO_MMA, int MT, bool RESID, class Wait>
__device__ __forceinline__ void dl_chi2_gemm_tile(const double* __restrict__ A, int64_t lda, const double* __restrict__ Wt, int64_t ldw, const double* __restrict__ bias,
                                                  double* __restrict__ part, int M, int K_pad, int n_tiles, const DlChi2Fin& fin, const DlChi2Panels& panels, int k_live,
                                                  double* __restrict__ resid, int64_t ldr, double* lds, int mb, int nt, Wait&& wait) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int ROWS = MT + DL_CG_N, VPT = ROWS / DL_CG_WAVES, MTILES = MT / 16;
    const int m0 = mb * MT, n0 = nt * DL_CG_N;
    if (m0 >= M) return;
    DL_CG_STAMP(0, __builtin_amdgcn_s_memtime) DL_CG_STAMP(6, __builtin_amdgcn_s_memrealtime)
    // staging by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write): piece i of wave w is the 1 KB segment of row w + 8 i of the
    // panel (rows 0-31 = A, 32-47 = Wt); the LDS destination of a piece is wave-uniform base + 16 B x lane
    int p_lo = 0, p_hi = K_pad / DL_CG_KP;   // K_pad is a multiple of the panel width (padding columns are zero in A and Wt)
    const uint32_t range = panels.range[nt < DL_CG_MAX_TILES ? nt : 0];
    if (nt < DL_CG_MAX_TILES && (range >> 8) != 0) { p_lo = (int)(range & 0xffu); p_hi = (int)(range >> 8); }
    const char* src[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int row = wave + DL_CG_WAVES * i;
        const double* base;
        if (row < MT) { int ar = m0 + row; if (ar > M - 1) ar = M - 1; base = A + (size_t)ar * lda; }
        else base = Wt + (size_t)(n0 + row - MT) * ldw;
        src[i] = reinterpret_cast<const char*>(base) + 16 * lane + (size_t)p_lo * (DL_CG_KP * 8);
    }
    const double bj = bias[n0 + r16];        // requested now, used in the epilogue
    const int n_panels = p_hi - p_lo;
    constexpr int BUF = ROWS * DL_CG_LD, NJ = DL_CG_KP / 4 / DL_CG_WAVES;   // doubles per LDS buffer; k-steps per wave and panel
#define DL_CG_DMA(p)                                                                                                                 \
    {   const size_t off = (size_t)(p) * (DL_CG_KP * 8);                                                                             \
        double* dst = lds + ((p) % DL_CG_NBUF) * BUF + wave * DL_CG_LD;                                                              \
        _Pragma("unroll") for (int i = 0; i < VPT; ++i)                                                                               \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + off),                          \
                                             (__attribute__((address_space(3))) void*)(dst + DL_CG_WAVES * i * DL_CG_LD), 16, 0, 0); }
    dl_cg_double4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
    const double* la = lds + r16 * DL_CG_LD + g;
    // prologue: panels 0 and 1 requested; panel 0 landed (counted wait: the pieces of panel 1 may still fly) and visible (barrier)
    wait();
    if (DO_LOAD) { DL_CG_DMA(0) if (n_panels > 1) { DL_CG_DMA(1) } }
    // fused finalize: the priors depend on theta only -- lanes 0-31 of wave 0 evaluate them for the 32 points of the row block now, while the first panels
    // are in flight (whichever column block arrives last will need them; two registers are carried through the main loop)
    double fin_lp = 0.;
    int fin_nan = 0;
    if (fin.counters != nullptr && wave == 0 && lane < MT && m0 + lane < M) {
        const double inf = __builtin_huge_val();
        for (int p = 0; p < fin.n_params; ++p) {
            double x = fin.theta[(size_t)(m0 + lane) * fin.n_params + p];
            const double* pr = fin.priors + 5 * p;
            if (x != x) fin_nan = 1;
            bool isin = (pr[1] <= x) && (x <= pr[2]);
            double v = 0.;
            if (pr[0] == 1.) { double t = x - pr[3]; v = -0.5 * (t * t) / (pr[4] * pr[4]); }   // parameter.py:2007
            fin_lp += isin ? v : -inf;
        }
    }
    if (n_panels > 1) __asm__ volatile("s_waitcnt vmcnt(%0)" : : "n"(VPT) : "memory");
    else __asm__ volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __builtin_amdgcn_s_barrier();
    DL_CG_STAMP(1, __builtin_amdgcn_s_memtime)
#define DL_CG_MULTIPLY(p)                                                                                                            \
    {   const double* lb = la + ((p) % DL_CG_NBUF) * BUF;                                                                            \
        double a0[NJ], a1[NJ], bb[NJ];                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                                             \
            const int ks = wave + DL_CG_WAVES * j; a0[j] = lb[4 * ks]; a1[j] = MTILES > 1 ? lb[16 * DL_CG_LD + 4 * ks] : 0.; bb[j] = lb[MT * DL_CG_LD + 4 * ks]; } \
        if (DO_MMA) { _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                               \
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], bb[j], acc0, 0, 0, 0);                                                \
            if (MTILES > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], bb[j], acc1, 0, 0, 0); } } }
    // the last panel of K is partly padding (1200 of 1280 columns: 12 of its 32 k-steps are live): waves whose k-steps are padding skip their MFMAs there
#define DL_CG_MULTIPLY_LIM(p, lim)                                                                                                   \
    {   const double* lb = la + ((p) % DL_CG_NBUF) * BUF;                                                                            \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                                             \
            const int ks = wave + DL_CG_WAVES * j;                                                                                   \
            if (DO_MMA && ks < (lim)) {                                                                                              \
                const double a0 = lb[4 * ks], a1 = MTILES > 1 ? lb[16 * DL_CG_LD + 4 * ks] : 0., bb = lb[MT * DL_CG_LD + 4 * ks];    \
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bb, acc0, 0, 0, 0);                                                  \
                if (MTILES > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bb, acc1, 0, 0, 0); } } }
    // iteration p: request panel p + 2 (its buffer held panel p - 1, whose reads every wave retired before the last barrier), read the operands of
    // panel p, multiply, then wait until the wave's own pieces of panel p + 1 have landed (the youngest requests, panel p + 2, stay in flight)
    // and its LDS reads are back; one raw barrier per panel, no vmcnt(0) in the steady loop; the last two panels are peeled (nothing left to request)
    int p = 0;
    for (; p + 2 < n_panels; ++p) {
        if (DO_LOAD) { DL_CG_DMA(p + 2) }
        DL_CG_MULTIPLY(p)
        __asm__ volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(VPT) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (p + 1 < n_panels) {
        DL_CG_MULTIPLY(p)
        __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();
        ++p;
    }
    {
        int lim = (k_live - (p_lo + p) * DL_CG_KP + 3) / 4;        // live k-steps of this (last) panel
        lim = lim < 0 ? 0 : (lim > DL_CG_KP / 4 ? DL_CG_KP / 4 : lim);
        DL_CG_MULTIPLY_LIM(p, lim)
    }
#undef DL_CG_MULTIPLY
#undef DL_CG_MULTIPLY_LIM
#undef DL_CG_DMA
    dl_chi2_gemm_finish<MT, RESID>(acc0, acc1, bj, part, M, n_tiles, fin, resid, ldr, lds, mb, nt, fin_lp, fin_nan);
}

// Round 6: the same tile with the B OPERAND OF THE WHOLE TILE IN REGISTERS.  W~ is constant: dl_create lays it out in MFMA fragment order, wfrag [N_pad / 16][K_pad / 4][64]
// (k-step ks of column block nt: lane l holds W~[16 nt + (l & 15)][4 ks + (l >> 4)]), so the B operand of a k-step is ONE 512-byte load per wave.  A wave owns two k-steps
// of every panel: its share of B over the full K is 2 x n_panels doubles per lane (20 at K_pad = 1280) -- requested at entry, all of it at once, before the first DMA piece.
// LDS-DMA then carries only the MT rows of `power`, 32 instead of 48 KB per panel (the panel time of the round-5 loop was its 48 KB at the rate of L2-served LDS-DMA,
// 37 B / clock / CU), and the B-side LDS reads disappear.  The panel loop is unrolled over DL_CG_PMAX panels (static registers and LDS buffers; tiles with more live panels take
// dl_chi2_gemm_tile).  Tried first: B of panel p + 2 requested with the panel's DMA pieces into three rotating register sets -- as inline assembly the compiler moves the
// "results" between registers before they have landed; as plain loads in a loop unrolled by three it rotates the sets by copies behind a vmcnt(0) at the back-edge.
#define DL_CG_PMAX 12
template <bool DO_LOAD, bool DO_MMA, int MT, bool RESID>
__device__ __forceinline__ void dl_chi2_gemm_tile_bf(const double* __restrict__ A, int64_t lda, const double* __restrict__ wfrag, const double* __restrict__ bias,
                                                     double* __restrict__ part, int M, int K_pad, int n_tiles, const DlChi2Fin& fin, const DlChi2Panels& panels, int k_live,
                                                     double* __restrict__ resid, int64_t ldr, double* lds, int mb, int nt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int VPT = MT / DL_CG_WAVES, MTILES = MT / 16;
    static_assert(VPT >= 1 && DL_CG_KP / 4 / DL_CG_WAVES == 2, "16 waves: one or two row pieces and two k-steps per wave and panel");
    const int m0 = mb * MT, n0 = nt * DL_CG_N;
    if (m0 >= M) return;
    DL_CG_STAMP(0, __builtin_amdgcn_s_memtime) DL_CG_STAMP(6, __builtin_amdgcn_s_memrealtime)
    int p_lo = 0, p_hi = K_pad / DL_CG_KP;
    const uint32_t range = panels.range[nt < DL_CG_MAX_TILES ? nt : 0];
    if (nt < DL_CG_MAX_TILES && (range >> 8) != 0) { p_lo = (int)(range & 0xffu); p_hi = (int)(range >> 8); }
    const int n_panels = p_hi - p_lo;        // (<= DL_CG_PMAX: the launcher's condition)
    // the wave's share of B: k-steps wave and wave + 16 of every panel (a panel = 32 k-steps = 2048 doubles of the column block's fragment stream); panels beyond the tile's
    // last one are clamped (loaded, not used: no branch around a request)
    const double* bsrc = wfrag + ((size_t)nt * (K_pad / 4) + (size_t)p_lo * (DL_CG_KP / 4) + wave) * 64 + lane;
    double bq[DL_CG_PMAX][2];
#pragma unroll
    for (int q = 0; q < DL_CG_PMAX; ++q) {
        const double* bp = bsrc + (size_t)(q < n_panels ? q : n_panels - 1) * (DL_CG_KP / 4 * 64);
        bq[q][0] = bp[0]; bq[q][1] = bp[DL_CG_WAVES * 64];
    }
    const char* src[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        int ar = m0 + wave + DL_CG_WAVES * i; if (ar > M - 1) ar = M - 1;
        src[i] = reinterpret_cast<const char*>(A + (size_t)ar * lda) + 16 * lane + (size_t)p_lo * (DL_CG_KP * 8);
    }
    const double bj = bias[n0 + r16];        // requested now, used in the epilogue
    constexpr int BUF = MT * DL_CG_LD;       // doubles per LDS buffer (rows of A only)
#define DL_CG_DMA_A(p)                                                                                                               \
    {   const size_t off = (size_t)(p) * (DL_CG_KP * 8);                                                                             \
        double* dst = lds + ((p) % DL_CG_NBUF) * BUF + wave * DL_CG_LD;                                                              \
        _Pragma("unroll") for (int i = 0; i < VPT; ++i)                                                                               \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + off),                          \
                                             (__attribute__((address_space(3))) void*)(dst + DL_CG_WAVES * i * DL_CG_LD), 16, 0, 0); }
    dl_cg_double4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
    const double* la = lds + r16 * DL_CG_LD + g;
    if (DO_LOAD) { DL_CG_DMA_A(0) if (n_panels > 1) { DL_CG_DMA_A(1) } }
    double fin_lp = 0.;
    int fin_nan = 0;
    if (fin.counters != nullptr && wave == 0 && lane < MT && m0 + lane < M) {     // (fused finalize: the priors of the row block while the first panels fly)
        const double inf = __builtin_huge_val();
        for (int p = 0; p < fin.n_params; ++p) {
            double x = fin.theta[(size_t)(m0 + lane) * fin.n_params + p];
            const double* pr = fin.priors + 5 * p;
            if (x != x) fin_nan = 1;
            bool isin = (pr[1] <= x) && (x <= pr[2]);
            double v = 0.;
            if (pr[0] == 1.) { double t = x - pr[3]; v = -0.5 * (t * t) / (pr[4] * pr[4]); }   // parameter.py:2007
            fin_lp += isin ? v : -inf;
        }
    }
    // panel 0 landed (counted wait: the pieces of panel 1 may still fly; B was requested before either: it is back as well) and visible (barrier)
    if (n_panels > 1) __asm__ volatile("s_waitcnt vmcnt(%0)" : : "n"(VPT) : "memory");
    else __asm__ volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __builtin_amdgcn_s_barrier();
    DL_CG_STAMP(1, __builtin_amdgcn_s_memtime)
    int lim = (k_live - (p_lo + n_panels - 1) * DL_CG_KP + 3) / 4;        // live k-steps of the last panel (1200 of 1280 columns: 12 of its 32 k-steps are padding)
    lim = lim < 0 ? 0 : (lim > DL_CG_KP / 4 ? DL_CG_KP / 4 : lim);
    // iteration p: request panel p + 2 (its buffer held panel p - 1, whose reads every wave retired before the last barrier), multiply panel p, then wait until the wave's own
    // pieces of panel p + 1 have landed (panel p + 2 stays in flight) and its LDS reads are back; one raw barrier per panel
#pragma unroll
    for (int p = 0; p < DL_CG_PMAX; ++p) {
        if (p < n_panels) {       // (no `break`: the loop must unroll completely -- bq is indexed by p)
        if (DO_LOAD && p + 2 < n_panels) { DL_CG_DMA_A(p + 2) }
        const double* lb = la + (p % DL_CG_NBUF) * BUF;
        const int klim = p + 1 < n_panels ? DL_CG_KP / 4 : lim;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ks = wave + DL_CG_WAVES * j;
            if (DO_MMA && ks < klim) {      // (uniform per wave; false only in the padding of the last panel)
                const double a0 = lb[4 * ks], a1 = MTILES > 1 ? lb[16 * DL_CG_LD + 4 * ks] : 0.;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq[p][j], acc0, 0, 0, 0);
                if (MTILES > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq[p][j], acc1, 0, 0, 0);
            }
        }
        if (p + 1 < n_panels) {
            if (p + 2 < n_panels) __asm__ volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(VPT) : "memory");
            else __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
            __builtin_amdgcn_s_barrier();
        }
        }
    }
#undef DL_CG_DMA_A
    dl_chi2_gemm_finish<MT, RESID>(acc0, acc1, bj, part, M, n_tiles, fin, resid, ldr, lds, mb, nt, fin_lp, fin_nan);
}

template <bool DO_LOAD, bool DO_MMA, int MT = DL_CG_M, bool RESID = false>
__global__ __launch_bounds__(64 * DL_CG_WAVES) void dl_chi2_gemm_bf_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ wfrag, const double* __restrict__ bias,
                                                              double* __restrict__ part, int M, int K_pad, int n_tiles, DlChi2Fin fin, DlChi2Panels panels, int k_live,
                                                              double* __restrict__ resid, int64_t ldr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    dl_kernarg_prefetch<256>();
    const int L = blockIdx.x;                // XCD-aware decode as in dl_chi2_gemm_kernel: row block = xcd + 8 q
    const int xcd = L & 7, rest = L >> 3;
    const int nt = rest % n_tiles, mb = xcd + 8 * (rest / n_tiles);
    dl_chi2_gemm_tile_bf<DO_LOAD, DO_MMA, MT, RESID>(A, lda, wfrag, bias, part, M, K_pad, n_tiles, fin, panels, k_live, resid, ldr, lds, mb, nt);
}

template <bool DO_LOAD, bool DO_MMA, int MT = DL_CG_M, bool RESID = false>
__global__ __launch_bounds__(64 * DL_CG_WAVES) void dl_chi2_gemm_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ Wt, int64_t ldw,
                                                           const double* __restrict__ bias, double* __restrict__ part, int M, int K_pad, int n_tiles, DlChi2Fin fin,
                                                           DlChi2Panels panels, int k_live, double* __restrict__ resid, int64_t ldr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    dl_kernarg_prefetch<256>();   // pointers, sizes, finalize block, panel ranges: four lines, one round trip
    // XCD-aware decode of the linear workgroup id L = xcd + 8 (nt + n_tiles q): row block = xcd + 8 q
    const int L = blockIdx.x;
    const int xcd = L & 7, rest = L >> 3;
    const int nt = rest % n_tiles, mb = xcd + 8 * (rest / n_tiles);
    dl_chi2_gemm_tile<DO_LOAD, DO_MMA, MT, RESID>(A, lda, Wt, ldw, bias, part, M, K_pad, n_tiles, fin, panels, k_live, resid, ldr, lds, mb, nt, []() {});
}
