// dl_kernels.hip -- gfx950 (CDNA4) kernels of the full-shape likelihood path.
//
//   dl_fullshape_kernel : one workgroup per (point, observable): template -> spline -> AP -> multipoles
//                         -> tracer combination, everything staged in LDS.
//   dl_window_gemm      : C[B, N] = A[B, K] . Wt[N, K]^T + bias, fp64 MFMA v_mfma_f64_16x16x4_f64;
//                         used for the (precision-whitened) window convolution (rows a6 + a8).
//   dl_finalize_kernel  : chi2 = |whitened residual|^2 by wavefront shuffles, priors, status (rows a8 + a9).
#include <hip/hip_runtime.h>

#include "dl_fullshape.h"
#include "dl_kernels.h"

// ------------------------------------------------------------------------------------------------
// theory kernel
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DL_FS_THREADS) void dl_fullshape_kernel(const DlObsDev* __restrict__ obs, int n_obs, const double* __restrict__ theta,
                                                                     int n_params, double* __restrict__ power, int64_t ld_power,
                                                                     double* __restrict__ tables, int64_t ld_tables) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, iobs = blockIdx.y;
    const DlObsDev& o = obs[iobs];
    DlFsShared s;
    s.y = lds;
    s.M = lds + o.n_t;
    s.z = lds + 2 * (size_t)o.n_t;
    s.pt = lds + 3 * (size_t)o.n_t;
    const double* th = theta + (size_t)b * n_params;
    const int tid = threadIdx.x, nthr = blockDim.x;
    dl_fs_phase01(tid, nthr, o, th, s);
    __syncthreads();
    if (!o.fixed_spline) {
        dl_fs_phase2a(tid, nthr, o, s);
        __syncthreads();
        dl_fs_phase2b(tid, nthr, o, s);
        __syncthreads();
        dl_fs_phase2c(tid, nthr, o, s);
        __syncthreads();
    }
    double* prow = power + (size_t)b * ld_power + o.col_offset;
    double* trow = tables ? tables + (size_t)b * ld_tables : nullptr;
    dl_fs_phase3(tid, nthr, o, s, prow, trow);
}

void dl_launch_fullshape(const DlObsDev* obs_dev, int n_obs, int max_n_t, const double* theta, int n_params, int64_t B, double* power, int64_t ld_power,
                         double* tables, int64_t ld_tables, hipStream_t stream) {
    dim3 grid((unsigned)B, (unsigned)n_obs);
    size_t shmem = dl_fs_shared_doubles(max_n_t) * sizeof(double);
    static size_t shmem_optin = 0;
    if (shmem > 48 * 1024 && shmem > shmem_optin) {  // large templates (e.g. 2000-knot BAO tables) need the dynamic-LDS opt-in
        (void)hipFuncSetAttribute((const void*)dl_fullshape_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        shmem_optin = shmem;
    }
    hipLaunchKernelGGL(dl_fullshape_kernel, grid, dim3(DL_FS_THREADS), shmem, stream, obs_dev, n_obs, theta, n_params, power, ld_power, tables, ld_tables);
}

// ------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM:  C[M, N] = A[M, K] . Wt[N, K]^T + bias[N]
//   A  row-major, leading dimension lda (multiple of 32, padding columns zero)
//   Wt row-major, leading dimension ldw (multiple of 32, padding zero), N_pad rows (multiple of 32)
//   one workgroup = 4 waves computes a 16 (M) x 32 (N) tile; the 4 waves split K in 32-wide chunks
//   (round-robin) and are summed through LDS.  v_mfma_f64_16x16x4_f64 operand layout (guide section 3):
//   A operand lane l = A[row l&15][k l>>4], B operand lane l = B[k l>>4][col l&15],
//   C/D reg r of lane l = C[row (l>>4) + 4 r][col l&15].
//   Inside a 32-chunk lane group g = l>>4 owns k = 8 g .. 8 g + 7 (contiguous 64 B per lane), i.e. the
//   k -> (mfma step, lane group) assignment is permuted identically for A and Wt: the sum is unchanged.
// ------------------------------------------------------------------------------------------------
typedef double dl_double4 __attribute__((ext_vector_type(4)));
typedef double dl_double2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void dl_window_gemm_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ Wt, int64_t ldw,
                                                             const double* __restrict__ bias, double* __restrict__ C, int64_t ldc, int M, int N_valid, int K_pad) {
    __shared__ __attribute__((aligned(16))) double red[3][2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 32;
    int arow = m0 + r16;
    if (arow > M - 1) arow = M - 1;
    const double* ap = A + (size_t)arow * lda + g * 8;
    const double* b0p = Wt + (size_t)(n0 + r16) * ldw + g * 8;
    const double* b1p = Wt + (size_t)(n0 + 16 + r16) * ldw + g * 8;
    dl_double4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
    const int nchunks = K_pad / 32;
    for (int kc = wave; kc < nchunks; kc += 4) {
        const dl_double2* a2 = reinterpret_cast<const dl_double2*>(ap + (size_t)kc * 32);
        const dl_double2* b02 = reinterpret_cast<const dl_double2*>(b0p + (size_t)kc * 32);
        const dl_double2* b12 = reinterpret_cast<const dl_double2*>(b1p + (size_t)kc * 32);
        dl_double2 av[4], bv0[4], bv1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = a2[q]; bv0[q] = b02[q]; bv1[q] = b12[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q].x, bv0[q].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q].x, bv1[q].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q].y, bv0[q].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q].y, bv1[q].y, acc1, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[wave - 1][0][r][lane] = acc0[r]; red[wave - 1][1][r][lane] = acc1[r]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s0 = acc0[r], s1 = acc1[r];
#pragma unroll
            for (int w = 0; w < 3; ++w) { s0 += red[w][0][r][lane]; s1 += red[w][1][r][lane]; }
            int row = m0 + g + 4 * r;
            if (row < M) {
                if (n0 + r16 < N_valid) C[(size_t)row * ldc + n0 + r16] = s0 + bias[n0 + r16];
                if (n0 + 16 + r16 < N_valid) C[(size_t)row * ldc + n0 + 16 + r16] = s1 + bias[n0 + 16 + r16];
            }
        }
    }
}

void dl_launch_window_gemm(const double* A, int64_t lda, const double* Wt, int64_t ldw, const double* bias, double* C, int64_t ldc, int64_t M, int N_valid, int N_pad,
                           int K_pad, hipStream_t stream) {
    dim3 grid((unsigned)((M + 15) / 16), (unsigned)(N_pad / 32));
    hipLaunchKernelGGL(dl_window_gemm_kernel, grid, dim3(256), 0, stream, A, lda, Wt, ldw, bias, C, ldc, (int)M, N_valid, K_pad);
}

// ------------------------------------------------------------------------------------------------
// observable transform (power_spectrum.py:402-404): flat -> (3 (flat / data)^(1/3) - 2) data, in place
// ------------------------------------------------------------------------------------------------
__global__ void dl_transform_kernel(double* __restrict__ flat, int64_t ld, const double* __restrict__ data, const int32_t* __restrict__ transform, int n, int64_t B) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    int64_t b = idx / n;
    int j = (int)(idx - b * n);
    if (transform[j] == 1) {
        double d = data[j], t = flat[b * ld + j];
        flat[b * ld + j] = (3. * pow(t / d, 1. / 3.) - 2.) * d;
    }
}

void dl_launch_transform(double* flat, int64_t ld, const double* data, const int32_t* transform, int n, int64_t B, hipStream_t stream) {
    int64_t total = B * n;
    hipLaunchKernelGGL(dl_transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, flat, ld, data, transform, n, B);
}

// ------------------------------------------------------------------------------------------------
// finalize: one wavefront per point.  loglike = -1/2 sum_j dtilde_j^2 (likelihoods/base.py:13-17, 660 with the
// precision folded in as its Cholesky factor), logprior (parameter.py:1889-1897, 1994-2007), status.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dl_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void dl_finalize_kernel(const double* __restrict__ dtilde, int64_t ld, int n, const double* __restrict__ theta, int n_params,
                                                          const double* __restrict__ priors, int64_t B, double* __restrict__ loglike,
                                                          double* __restrict__ logprior, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const double* row = dtilde + (size_t)b * ld;
    double sum = 0.;
    for (int j = lane; j < n; j += 64) { double v = row[j]; sum = fma(v, v, sum); }
    sum = dl_wave_sum(sum);
    // priors: lanes stride over parameters
    double lp = 0.;
    int nan_in = 0;
    const double inf = __builtin_huge_val();
    for (int p = lane; p < n_params; p += 64) {
        double x = theta[(size_t)b * n_params + p];
        const double* pr = priors + 5 * p;
        if (x != x) nan_in = 1;
        bool isin = (pr[1] <= x) && (x <= pr[2]);
        double v = 0.;
        if (pr[0] == 1.) { double t = x - pr[3]; v = -0.5 * (t * t) / (pr[4] * pr[4]); }   // parameter.py:2007
        lp += isin ? v : -inf;
    }
    lp = dl_wave_sum(lp);
    nan_in = __any(nan_in);
    if (lane == 0) {
        double ll = -0.5 * sum;
        int st = DL_ST_OK;
        if (nan_in) st = DL_ST_NAN_INPUT;
        else if (lp == -inf) st = DL_ST_OUT_OF_PRIOR;
        else if (!(ll == ll) || ll == inf || ll == -inf) st = DL_ST_NONFINITE;
        if (loglike) loglike[b] = ll;
        if (logprior) logprior[b] = lp;
        if (status) status[b] = st;
    }
}

void dl_launch_finalize(const double* dtilde, int64_t ld, int n, const double* theta, int n_params, const double* priors, int64_t B, double* loglike, double* logprior,
                        int32_t* status, hipStream_t stream) {
    hipLaunchKernelGGL(dl_finalize_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, dtilde, ld, n, theta, n_params, priors, B, loglike, logprior, status);
}
