// Split-K fp64 MFMA GEMM with LDS-DMA staging (successor of dl_gemm_tiled.h for the residual-slab path: marginalised likelihoods, batches above 2048 points):
//   slab[s][M, N] = A[M, Ks] . Wt[N, Ks]^T over the K-slice Ks of split s;  workgroup = 64 (M) x 128 (N) tile, 16 waves 2 (M) x 8 (N), each 32 x 16 (or 8 waves of 32 x 32).
//   K advances in 32-wide panels: 64 A rows + 128 Wt rows x 256 B = 48 KB, three LDS buffers, two panels in flight (global_load_lds_dwordx4, counted
//   vmcnt, raw barrier: same pipeline as dl_chi2_gemm.h).  One LDS-DMA wave-instruction moves 1 KB to CONTIGUOUS LDS = four 256-byte rows here, so rows
//   cannot be padded individually; instead the 16-byte chunks of row sr (0-3) of a piece are XOR-permuted by 4 sr on the way in -- the global source address
//   is per lane, the LDS image stays linear -- and pieces are 16 bytes apart: the 16 rows of an MFMA operand read then fall on 16 distinct bank groups
//   (bank = 4 (row >> 2) + 16 ((chunk >> 2) ^ (row & 3)) + ...).  The reader applies the same permutation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double dl_gd_double4 __attribute__((ext_vector_type(4)));

#define DL_GD_M 64
#define DL_GD_N 128
#define DL_GD_KP 32                                   // panel width (doubles)
#define DL_GD_PIECES ((DL_GD_M + DL_GD_N) / 4)        // 1 KB pieces (4 rows) per panel = 48
#define DL_GD_PLD 130                                 // piece stride in doubles (1024 B + 16 B)
#define DL_GD_BUF (DL_GD_PIECES * DL_GD_PLD)          // doubles per LDS buffer
#define DL_GD_NBUF 3
#ifndef DL_GD_WAVES
#define DL_GD_WAVES 16                                 // 8: waves 2 x 4 of 32 x 32; 16: waves 2 x 8 of 32 x 16 (four per SIMD)
#endif
#define DL_GD_WN (DL_GD_WAVES / 2)                    // waves along N
#define DL_GD_TJ (DL_GD_N / DL_GD_WN / 16)            // MFMA column tiles per wave (2 or 1)
#define DL_GD_VPT (DL_GD_PIECES / DL_GD_WAVES)        // pieces per wave and panel
#define DL_GD_LDS_BYTES (DL_GD_NBUF * DL_GD_BUF * 8)

// panels_per_split panels of K per blockIdx.z; n_panels = K_pad / 32.
// CHI2 (single split only): instead of the residual slab, bias is added and the squares are summed over each wave's 16 DL_GD_TJ columns: part[M, N_pad / (16 TJ)]
// partial chi2 per row, finished by dl_finalize_part_kernel (plain likelihood: the residual itself is never needed).
template <bool CHI2>
__global__ __launch_bounds__(64 * DL_GD_WAVES) void dl_window_gemm_dma_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ Wt, int64_t ldw,
                                                                 double* __restrict__ slabs, int64_t slab_stride, int64_t ldc, int M, int panels_per_split, int n_panels,
                                                                 const double* __restrict__ bias, int n_live) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave / DL_GD_WN, wn = wave % DL_GD_WN;
    const int m0 = blockIdx.x * DL_GD_M, n0 = blockIdx.y * DL_GD_N, split = blockIdx.z;
    const int pa = split * panels_per_split;
    int pb = pa + panels_per_split;
    if (pb > n_panels) pb = n_panels;
    if (pa >= pb) return;
    // DMA sources: piece q = wave + 8 i; lane l fetches row 4 q' + (l >> 4), 16-byte chunk (l & 15) ^ (4 (l >> 4)) of the panel
    const int sr_l = lane >> 4, cd = (lane & 15) ^ (sr_l << 2);
    const char* src[DL_GD_VPT];
#pragma unroll
    for (int i = 0; i < DL_GD_VPT; ++i) {
        const int q = wave + DL_GD_WAVES * i;
        const double* base;
        if (q < DL_GD_M / 4) { int ar = m0 + 4 * q + sr_l; if (ar > M - 1) ar = M - 1; base = A + (size_t)ar * lda; }
        else base = Wt + (size_t)(n0 + 4 * (q - DL_GD_M / 4) + sr_l) * ldw;
        src[i] = reinterpret_cast<const char*>(base) + 16 * cd;
    }
#define DL_GD_DMA(p)                                                                                                              \
    {   const size_t off = (size_t)(p) * (DL_GD_KP * 8);                                                                          \
        double* dst = lds + (((p) - pa) % DL_GD_NBUF) * DL_GD_BUF + wave * DL_GD_PLD;                                             \
        _Pragma("unroll") for (int i = 0; i < DL_GD_VPT; ++i)                                                                     \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + off),                       \
                                             (__attribute__((address_space(3))) void*)(dst + DL_GD_WAVES * i * DL_GD_PLD), 16, 0, 0); }
    // operand addresses: tile row R = base + r16 -> piece R >> 2, row-in-piece sr = r16 & 3, k = 4 ks + g -> chunk ((2 ks) ^ (4 sr)) + (g >> 1), double g & 1
    const int sr = r16 & 3, s4 = sr << 2, gh = g >> 1;
    const double* la = lds + ((wm * 32) / 4 + (r16 >> 2)) * DL_GD_PLD + sr * 32 + (g & 1);                       // A tile i: + 4 i pieces
    const double* lw = lds + (DL_GD_M / 4 + (wn * 16 * DL_GD_TJ) / 4 + (r16 >> 2)) * DL_GD_PLD + sr * 32 + (g & 1);   // Wt tile j: + 4 j pieces
    // columns >= n_live are padding (all-zero rows of Wt: 60 data points padded to the 128-wide tile): a wave whose columns are all padding stages its pieces
    // and meets the barriers like the others, but issues no MFMA -- the accumulators stay zero, which is the product.  (The waves of a column group sit on the
    // same SIMDs as their live twins: the MFMA work per SIMD halves with half of the tile live.)
    const bool live = n0 + wn * 16 * DL_GD_TJ < n_live;
    dl_gd_double4 acc[2][DL_GD_TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < DL_GD_TJ; ++j) acc[i][j] = (dl_gd_double4){0., 0., 0., 0.};
#define DL_GD_MULTIPLY(p)                                                                                                         \
    if (live) {   const int bo = (((p) - pa) % DL_GD_NBUF) * DL_GD_BUF;                                                           \
        _Pragma("unroll") for (int ks = 0; ks < DL_GD_KP / 4; ++ks) {                                                             \
            const int off = bo + ((((2 * ks) ^ s4) + gh) << 1);                                                                   \
            const double a0 = la[off], a1 = la[off + 4 * DL_GD_PLD];                                                              \
            _Pragma("unroll") for (int j = 0; j < DL_GD_TJ; ++j) {                                                                \
                const double bj = lw[off + 4 * j * DL_GD_PLD];                                                                    \
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bj, acc[0][j], 0, 0, 0);                                     \
                acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bj, acc[1][j], 0, 0, 0);                                     \
            } } }
    DL_GD_DMA(pa)
    if (pa + 1 < pb) { DL_GD_DMA(pa + 1) }
    if (pa + 1 < pb) __asm__ volatile("s_waitcnt vmcnt(%0)" : : "n"(DL_GD_VPT) : "memory");
    else __asm__ volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __builtin_amdgcn_s_barrier();
    int p = pa;
    // Steady state, three panels per trip with the buffer of each known at compile time.  With a run-time buffer index every k-step pays vector instructions for the
    // permuted operand offset and every request for its 64-bit address (20 + 3 per 16 MFMAs: `v_mfma_f64` runs at the vector unit's own rate, those instructions are
    // not free beside it).  Here the eight permuted offsets of a lane are registers computed once (two sets: buffers 0 / 1 within the 16-bit immediate of a DS read,
    // buffer 2 beyond it), the per-lane request pointers advance once per trip and the panel inside the trip is the request's immediate offset -- which an LDS-DMA
    // instruction adds to BOTH its addresses, so the LDS base handed over in M0 is moved back by as much.  The generic loop below takes what is left.
    static_assert(DL_GD_NBUF == 3 && DL_GD_KP == 32, "the unrolled loop rotates three buffers of eight k-steps");
    {
        const double* pa01[DL_GD_KP / 4]; const double* pa2[DL_GD_KP / 4]; const double* pw01[DL_GD_KP / 4]; const double* pw2[DL_GD_KP / 4];
#pragma unroll
        for (int ks = 0; ks < DL_GD_KP / 4; ++ks) {
            const int off = ((((2 * ks) ^ s4) + gh) << 1);
            pa01[ks] = la + off; pa2[ks] = la + 2 * DL_GD_BUF + off; pw01[ks] = lw + off; pw2[ks] = lw + 2 * DL_GD_BUF + off;
        }
        const char* srcp[DL_GD_VPT];
#pragma unroll
        for (int i = 0; i < DL_GD_VPT; ++i) srcp[i] = src[i] + (size_t)(pa + 3) * (DL_GD_KP * 8);   // panel pa + 3: the middle request of the first trip (immediates -1, 0, +1 panels)
#define DL_GD_DMA_B(j, B)                                                                                                           \
    {   constexpr int imm = ((j) - 1) * (DL_GD_KP * 8);                                                                             \
        double* dst = lds + (B) * DL_GD_BUF + wave * DL_GD_PLD - imm / 8;                                                           \
        _Pragma("unroll") for (int i = 0; i < DL_GD_VPT; ++i)                                                                       \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcp[i],                                \
                                             (__attribute__((address_space(3))) void*)(dst + DL_GD_WAVES * i * DL_GD_PLD), 16, imm, 0); }
#define DL_GD_MULTIPLY_B(PA, PW, boff)                                                                                              \
    if (live) {                                                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < DL_GD_KP / 4; ++ks) {                                                               \
            const double a0 = (PA)[ks][(boff)], a1 = (PA)[ks][(boff) + 4 * DL_GD_PLD];                                              \
            _Pragma("unroll") for (int j = 0; j < DL_GD_TJ; ++j) {                                                                  \
                const double bj = (PW)[ks][(boff) + 4 * j * DL_GD_PLD];                                                             \
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bj, acc[0][j], 0, 0, 0);                                       \
                acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bj, acc[1][j], 0, 0, 0);                                       \
            } } }
#define DL_GD_STEP_B(j, B, PA, PW, boff)                                                                                            \
    {   DL_GD_DMA_B(j, ((B) + 2) % DL_GD_NBUF)                                                                                      \
        DL_GD_MULTIPLY_B(PA, PW, boff)                                                                                              \
        __asm__ volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(DL_GD_VPT) : "memory");                                           \
        __builtin_amdgcn_s_barrier(); }
        for (; p + 4 < pb; p += 3) {     // (p - pa is a multiple of 3 here: panel p sits in buffer 0)
            DL_GD_STEP_B(0, 0, pa01, pw01, 0) DL_GD_STEP_B(1, 1, pa01, pw01, DL_GD_BUF) DL_GD_STEP_B(2, 2, pa2, pw2, 0)
#pragma unroll
            for (int i = 0; i < DL_GD_VPT; ++i) srcp[i] += 3 * (DL_GD_KP * 8);
        }
#undef DL_GD_STEP_B
#undef DL_GD_MULTIPLY_B
#undef DL_GD_DMA_B
    }
    for (; p + 2 < pb; ++p) {
        DL_GD_DMA(p + 2)
        DL_GD_MULTIPLY(p)
        __asm__ volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(DL_GD_VPT) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (p + 1 < pb) {
        DL_GD_MULTIPLY(p)
        __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();
        ++p;
    }
    DL_GD_MULTIPLY(p)
#undef DL_GD_DMA
#undef DL_GD_MULTIPLY
    if (CHI2) {
        const int n_parts = (int)(gridDim.y * DL_GD_WN);
        double bj[DL_GD_TJ];
#pragma unroll
        for (int j = 0; j < DL_GD_TJ; ++j) bj[j] = bias[n0 + wn * 16 * DL_GD_TJ + 16 * j + r16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sq = 0.;
#pragma unroll
                for (int j = 0; j < DL_GD_TJ; ++j) { const double v = acc[i][j][r] + bj[j]; sq = fma(v, v, sq); }
                // C layout: reg r of lane l = C[row (l >> 4) + 4 r][col l & 15]: sum the 16 lanes of a lane group
                sq += __shfl_xor(sq, 1, 64);
                sq += __shfl_xor(sq, 2, 64);
                sq += __shfl_xor(sq, 4, 64);
                sq += __shfl_xor(sq, 8, 64);
                const int row = m0 + wm * 32 + 16 * i + g + 4 * r;
                if (r16 == 0 && row < M) slabs[(size_t)row * n_parts + blockIdx.y * DL_GD_WN + wn] = sq;
            }
        return;
    }
    double* out = slabs + (size_t)split * slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = m0 + wm * 32 + 16 * i + g + 4 * r;
            if (row < M) {
#pragma unroll
                for (int j = 0; j < DL_GD_TJ; ++j) out[(size_t)row * ldc + n0 + wn * 16 * DL_GD_TJ + 16 * j + r16] = acc[i][j][r];
            }
        }
}
