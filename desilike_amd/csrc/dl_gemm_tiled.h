// Tiled split-K fp64 MFMA GEMM kernel (see dl_kernels.hip for the description).  Kept in a header so that tools/gemm_probe.hip can
// instantiate it with parts switched off (DO_LOAD / DO_MMA / DO_STORE) to attribute its time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double dl_gt_double2 __attribute__((ext_vector_type(2)));
typedef double dl_gt_double4 __attribute__((ext_vector_type(4)));

#define DL_GT_M 64
#define DL_GT_N 128
#define DL_GT_K 16                  // granularity of the K split
#ifndef DL_GT_PANEL
#define DL_GT_PANEL 6               // K chunks per panel
#endif
#define DL_GT_LD (DL_GT_PANEL * DL_GT_K + 2)   // padded LDS row (doubles)
#define DL_GT_LDS_BYTES ((DL_GT_M + DL_GT_N) * DL_GT_LD * 8)

template <bool DO_LOAD, bool DO_MMA, bool DO_STORE>
__global__ __launch_bounds__(512) void dl_window_gemm_tiled_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ Wt, int64_t ldw,
                                                                   double* __restrict__ slabs, int64_t slab_stride, int64_t ldc, int M, int chunks_per_split, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * DL_GT_M, n0 = blockIdx.y * DL_GT_N, split = blockIdx.z;
    const int c0 = split * chunks_per_split;
    int c1 = c0 + chunks_per_split;
    if (c1 > nchunks) c1 = nchunks;
    // global -> register staging: thread t moves 16 B of row (t >> 3) [A] and of rows (t >> 3), (t >> 3) + 64 [Wt], per 16-wide chunk
    const int lrow = tid >> 3, kp = tid & 7;
    int arow = m0 + lrow;
    if (arow > M - 1) arow = M - 1;
    const dl_gt_double2* ag = reinterpret_cast<const dl_gt_double2*>(A + (size_t)arow * lda + kp * 2);
    const dl_gt_double2* w0g = reinterpret_cast<const dl_gt_double2*>(Wt + (size_t)(n0 + lrow) * ldw + kp * 2);
    const dl_gt_double2* w1g = reinterpret_cast<const dl_gt_double2*>(Wt + (size_t)(n0 + lrow + 64) * ldw + kp * 2);
    double* sa = lds + lrow * DL_GT_LD + kp * 2;
    double* sw0 = lds + (DL_GT_M + lrow) * DL_GT_LD + kp * 2;
    double* sw1 = lds + (DL_GT_M + lrow + 64) * DL_GT_LD + kp * 2;
    const double* la = lds + (wm * 32 + r16) * DL_GT_LD + g;
    const double* lw = lds + (DL_GT_M + wn * 32 + r16) * DL_GT_LD + g;
    dl_gt_double4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (dl_gt_double4){0., 0., 0., 0.};
    dl_gt_double2 ra[DL_GT_PANEL], rw0[DL_GT_PANEL], rw1[DL_GT_PANEL];
    const dl_gt_double2 zero2 = {0., 0.};
#define DL_GT_LOAD(cb)                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < DL_GT_PANEL; ++q) {                                                           \
        if (DO_LOAD && (cb) + q < c1) { size_t o = (size_t)((cb) + q) * (DL_GT_K / 2); ra[q] = ag[o]; rw0[q] = w0g[o]; rw1[q] = w1g[o]; } \
        else { ra[q] = zero2; rw0[q] = zero2; rw1[q] = zero2; }                                                         \
    }
    if (c0 < c1) { DL_GT_LOAD(c0) }
    for (int cb = c0; cb < c1; cb += DL_GT_PANEL) {
#pragma unroll
        for (int q = 0; q < DL_GT_PANEL; ++q) {
            *reinterpret_cast<dl_gt_double2*>(sa + q * DL_GT_K) = ra[q];
            *reinterpret_cast<dl_gt_double2*>(sw0 + q * DL_GT_K) = rw0[q];
            *reinterpret_cast<dl_gt_double2*>(sw1 + q * DL_GT_K) = rw1[q];
        }
        __syncthreads();
        const int cn = cb + DL_GT_PANEL;
        if (cn < c1) { DL_GT_LOAD(cn) }
        int nk = DO_MMA ? (c1 - cb < DL_GT_PANEL ? c1 - cb : DL_GT_PANEL) * (DL_GT_K / 4) : 0;
        if (nk == DL_GT_PANEL * (DL_GT_K / 4)) {
#pragma unroll
            for (int kk = 0; kk < DL_GT_PANEL * (DL_GT_K / 4); ++kk) {
                double a0 = la[4 * kk], a1 = la[16 * DL_GT_LD + 4 * kk];
                double b0 = lw[4 * kk], b1 = lw[16 * DL_GT_LD + 4 * kk];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else {
            for (int kk = 0; kk < nk; ++kk) {
                double a0 = la[4 * kk], a1 = la[16 * DL_GT_LD + 4 * kk];
                double b0 = lw[4 * kk], b1 = lw[16 * DL_GT_LD + 4 * kk];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (cn < c1) __syncthreads();
    }
#undef DL_GT_LOAD
    double* out = slabs + (size_t)split * slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = m0 + wm * 32 + 16 * i + g + 4 * r;
            if (row < M && (DO_STORE || acc[i][0][r] == 1.2345e300)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) out[(size_t)row * ldc + n0 + wn * 32 + 16 * j + r16] = acc[i][j][r];
            }
        }
}

