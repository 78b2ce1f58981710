// dl_fisher.hip -- Fisher algebra of the Gaussian likelihood on the device (include/desilike_amd.h: dl_eval_fisher).
//
// Reference: desilike/fisher.py:731-750 (Gaussian finalize of ``Fisher``): with flatdiff D [n], flatderiv dD [P, n] = d(flatdiff)/d(theta) and the precision,
//     derivp = dD . precision;  offset = -D . precision . D  (NO 1/2: line 746, reproduced);  gradient = -derivp . D;  hessian = -derivp . dD^T.
// The reference obtains dD from ``Differentiation`` (finite differences or jax) over MPI ranks; here the central-difference stencil of every centre
// (1 + 2 P points) is ONE batch through the theory kernels and the whitened window GEMM (residual rows d~ = L^T (flattheory - flatdata), precision = L L^T),
// and one wavefront per centre forms the derivative rows D~_p = (d~_{p+} - d~_{p-}) / (h_p- + h_p+) and their Gram matrix with fp64 MFMA:
//     X = [d~_0; D~_1 .. D~_P],  G = X X^T,  offset = -G[0][0],  gradient_p = -G[1 + p][0],  hessian_pq = -G[1 + p][1 + q].
#include <hip/hip_runtime.h>

#include <algorithm>

#include "dl_kernels.h"

typedef double dl_fi_double4 __attribute__((ext_vector_type(4)));
typedef double dl_fi_double2 __attribute__((ext_vector_type(2)));

// theta rows of the stencil: row (b, 0) = centre, (b, 1 + 2 p) = centre - lower_p e_p, (b, 2 + 2 p) = centre + upper_p e_p
__global__ void dl_fisher_stencil_kernel(const double* __restrict__ centers, const double* __restrict__ steps, int P, int64_t B, double* __restrict__ theta) {
    const int S = 1 + 2 * P;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * S * P) return;
    const int p = (int)(idx % P);
    const int64_t row = idx / P;
    const int r = (int)(row % S);
    const int64_t b = row / S;
    double v = centers[(size_t)b * P + p];
    if (r > 0 && (r - 1) / 2 == p) v += ((r - 1) & 1) ? steps[((size_t)b * P + p) * 2 + 1] : -steps[((size_t)b * P + p) * 2];
    theta[idx] = v;
}

void dl_launch_fisher_stencil(const double* centers, const double* steps, int P, int64_t B, double* theta, hipStream_t stream) {
    const int64_t total = B * (1 + 2 * P) * P;
    DL_LAUNCH(dl_fisher_stencil_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, centers, steps, P, B, theta);
}

// TILES = ceil((P + 1) / 16): 16 x 16 tiles of the Gram matrix per side (1 or 2).  One wavefront per centre, WAVES of them per workgroup.
template <int TILES>
__global__ __launch_bounds__(256) void dl_fisher_kernel(const double* __restrict__ rows, int64_t ld, int n, int n_slabs, int64_t slab_stride, const double* __restrict__ bias,
                                                        const double* __restrict__ steps, int P, int64_t B, int waves, int chunk, double* __restrict__ hessian,
                                                        double* __restrict__ gradient, double* __restrict__ offset) {
    extern __shared__ __attribute__((aligned(16))) double dl_fi_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * waves + wave;
    if (wave >= waves || b >= B) return;
    const int S = 1 + 2 * P, nrows = 1 + P;
    // the rows are staged `chunk` columns at a time (a multiple of 4; the whole row when it fits: short data vectors), the Gram tiles accumulate over the chunks
    const int n4 = 4 * ((n + 3) / 4), stride = chunk + 4;   // (+4 doubles: rows of an operand read 16 apart fall on different LDS banks)
    double* X = dl_fi_lds + (size_t)wave * (TILES * 16) * stride;
    const double* row0 = rows + (size_t)b * S * ld;
    const int xr = lane & 15, g = lane >> 4;
    dl_fi_double4 acc[TILES][TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i)
#pragma unroll
        for (int j = 0; j < TILES; ++j) acc[i][j] = dl_fi_double4{0., 0., 0., 0.};
    for (int base = 0; base < n4; base += chunk) {
    const int width = n4 - base < chunk ? n4 - base : chunk;
    // stage X: coalesced 16-byte loads of the stencil rows (all slabs), differences formed on the way in
    for (int cc = 2 * lane; cc < width; cc += 128) {
        const int c0 = base + cc;
        for (int r = 0; r < TILES * 16; ++r) {
            dl_fi_double2 v = {0., 0.};
            if (r == 0) {
                if (bias) v = *reinterpret_cast<const dl_fi_double2*>(bias + c0);
                for (int sl = 0; sl < n_slabs; ++sl) v += *reinterpret_cast<const dl_fi_double2*>(row0 + (size_t)sl * slab_stride + c0);
            } else if (r < nrows) {
                const int p = r - 1;
                const double* lo = row0 + (size_t)(1 + 2 * p) * ld;
                const double* hi = row0 + (size_t)(2 + 2 * p) * ld;
                dl_fi_double2 vl = {0., 0.}, vh = {0., 0.};
                for (int sl = 0; sl < n_slabs; ++sl) {
                    vl += *reinterpret_cast<const dl_fi_double2*>(lo + (size_t)sl * slab_stride + c0);
                    vh += *reinterpret_cast<const dl_fi_double2*>(hi + (size_t)sl * slab_stride + c0);
                }
                const double h = steps[((size_t)b * P + p) * 2] + steps[((size_t)b * P + p) * 2 + 1];
                v = (vh - vl) / h;
            }
            if (c0 >= n) v.x = 0.;
            if (c0 + 1 >= n) v.y = 0.;
            *reinterpret_cast<dl_fi_double2*>(X + (size_t)r * stride + cc) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Gram matrix with v_mfma_f64_16x16x4_f64: operand of tile t, lane l = X[16 t + (l & 15)][4 k + (l >> 4)] (A and B operands of a diagonal tile are one register)
    for (int k = 0; k < width / 4; ++k) {
        double x[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) x[t] = X[(size_t)(16 * t + xr) * stride + 4 * k + g];
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
            for (int j = i; j < TILES; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                        // the chunk has been read: the next one may overwrite it
    }
    // C layout: register r of lane l = G[16 i + (l >> 4) + 4 r][16 j + (l & 15)]
#pragma unroll
    for (int i = 0; i < TILES; ++i)
#pragma unroll
        for (int j = i; j < TILES; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * i + g + 4 * r, gj = 16 * j + xr;
                if (gi >= nrows || gj >= nrows) continue;
                const double v = -acc[i][j][r];
                if (gi == 0 && gj == 0) { if (offset) offset[b] = v; }
                if (gi == 0 && gj > 0 && gradient) gradient[(size_t)b * P + gj - 1] = v;           // G[0][1 + p] = d~ . D~_p
                if (gi > 0 && gj > 0 && hessian) {
                    hessian[((size_t)b * P + gi - 1) * P + gj - 1] = v;
                    if (j > i) hessian[((size_t)b * P + gj - 1) * P + gi - 1] = v;                 // off-diagonal tile: the transposed block
                }
                if (j > i && gj == 0) {}   // (tile (0, 1) has gj >= 16: never column 0)
            }
}

// waves per workgroup and columns staged at a time: a whole row per wavefront when four (or fewer) of them fit the LDS, else four wavefronts with the largest
// chunk of columns that fits (long data vectors: P(k, mu) grids of forecasts)
int dl_fisher_waves(int n, int P, size_t* shm_bytes, int* chunk_out) {
    const int tiles = (P + 1 + 15) / 16, n4 = 4 * ((n + 3) / 4);
    const size_t budget = 150 * 1024, row_bytes = (size_t)tiles * 16 * sizeof(double);
    int waves = (int)std::min<size_t>(4, budget / (row_bytes * (size_t)(n4 + 4))), chunk = n4;
    if (waves < 1) {
        waves = 4;
        chunk = (int)(budget / waves / row_bytes) - 4;
        chunk = chunk / 4 * 4;
    }
    if (chunk_out) *chunk_out = chunk;
    if (shm_bytes) *shm_bytes = row_bytes * (size_t)(chunk + 4) * waves;
    return chunk >= 4 ? waves : 0;
}

void dl_launch_fisher(const double* rows, int64_t ld, int n, int n_slabs, int64_t slab_stride, const double* bias, const double* steps, int P, int64_t B, double* hessian,
                      double* gradient, double* offset, hipStream_t stream) {
    size_t shm = 0;
    int chunk = 0;
    const int waves = dl_fisher_waves(n, P, &shm, &chunk);
    const unsigned grid = (unsigned)((B + waves - 1) / waves);
    auto launch = [&](auto kernel) {
        if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        DL_LAUNCH(kernel, dim3(grid), dim3(64 * waves), shm, stream, rows, ld, n, n_slabs, slab_stride, bias, steps, P, B, waves, chunk, hessian, gradient, offset);
    };
    if (P + 1 <= 16) launch(dl_fisher_kernel<1>);
    else launch(dl_fisher_kernel<2>);
}
