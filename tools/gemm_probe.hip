// tools/gemm_probe.hip -- attributes the time of the tiled split-K window GEMM (desilike_amd/csrc/dl_gemm_tiled.h) on MI355X.
// Launches the production kernel and copies of it with the loads / MFMAs / stores switched off, plus empty kernels, on the
// bench shape (M = 1024 rows, N = 128, K = 1344).  Meant to run under ``rocprofv3 --kernel-trace``: the per-kernel durations are
// read from the trace; HIP-event timings of back-to-back launches are printed as well.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I desilike_amd/csrc tools/gemm_probe.hip -o tools/bin/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dl_gemm_tiled.h"
#include "dl_chi2_gemm.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void empty_kernel(double* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.; }
__global__ void empty_lds_kernel(double* p) {
    extern __shared__ double l[];
    if (p == nullptr && threadIdx.x == 9999) p[0] = l[3];
}
// stands for the theory kernel: leaves A freshly written (dirty in the L2 of whichever XCD ran the workgroup)
__global__ void fill_kernel(double* p, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + 1e-9 * (double)(i & 1023);
}

template <typename F>
static void timeit(const char* name, int reps, hipStream_t st, F launch, bool refill, double* A, int64_t nA) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) {
        if (refill) hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, st, A, nA, 1.0 + i);
        launch();
    }
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.2f us per iteration (events, back-to-back%s)\n", name, 1e3 * ms / reps, refill ? ", incl. fill" : "");
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1024, N = 128, K = argc > 2 ? atoi(argv[2]) : 1280;
    const int reps = 50;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    double *A, *W, *slabs;
    const int nchunks = K / DL_GT_K;
    int64_t nA = (int64_t)M * K;
    CHECK(hipMalloc(&A, nA * 8)); CHECK(hipMalloc(&W, (int64_t)N * K * 8));
    CHECK(hipMalloc(&slabs, (int64_t)32 * M * N * 8));
    std::vector<double> h((size_t)N * K, 0.5);
    CHECK(hipMemcpy(W, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, st, A, nA, 1.0);
    const int mt = (M + DL_GT_M - 1) / DL_GT_M;
    auto splits = [&](int cps) { return (nchunks + cps - 1) / cps; };
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)dl_window_gemm_tiled_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    CHECK(hipFuncSetAttribute((const void*)empty_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DL_GT_LDS_BYTES));
    const int64_t ss = (int64_t)M * N;
#define GEMM(L, MM, S, cps) hipLaunchKernelGGL((dl_window_gemm_tiled_kernel<L, MM, S>), dim3(mt, 1, splits(cps)), dim3(512), DL_GT_LDS_BYTES, st, A, (int64_t)K, W, (int64_t)K, slabs, ss, (int64_t)N, M, cps, nchunks)
    printf("M=%d N=%d K=%d  m-tiles=%d  LDS per workgroup %d B\n", M, N, K, mt, (int)DL_GT_LDS_BYTES);
    timeit("empty 224 x 512", reps, st, [&] { hipLaunchKernelGGL(empty_kernel, dim3(224), dim3(512), 0, st, slabs); }, false, A, nA);
    timeit("empty 224 x 512, 147 KB LDS", reps, st, [&] { hipLaunchKernelGGL(empty_lds_kernel, dim3(224), dim3(512), DL_GT_LDS_BYTES, st, slabs); }, false, A, nA);
    timeit("fill only (A rewrite)", reps, st, [&] {}, true, A, nA);
    for (int cps : {6, 12, 3}) {
        char nm[128];
        snprintf(nm, sizeof nm, "cps=%d (%d splits): full", cps, splits(cps));
        timeit(nm, reps, st, [&] { GEMM(true, true, true, cps); }, false, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: full, A refilled", cps);
        timeit(nm, reps, st, [&] { GEMM(true, true, true, cps); }, true, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: no store", cps);
        timeit(nm, reps, st, [&] { GEMM(true, true, false, cps); }, false, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: no load", cps);
        timeit(nm, reps, st, [&] { GEMM(false, true, true, cps); }, false, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: no mfma", cps);
        timeit(nm, reps, st, [&] { GEMM(true, false, true, cps); }, false, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: mfma only", cps);
        timeit(nm, reps, st, [&] { GEMM(false, true, false, cps); }, false, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: load only", cps);
        timeit(nm, reps, st, [&] { GEMM(true, false, false, cps); }, false, A, nA);
        snprintf(nm, sizeof nm, "cps=%d: nothing", cps);
        timeit(nm, reps, st, [&] { GEMM(false, false, false, cps); }, false, A, nA);
    }
    // chi2 GEMM (column split, full K, partial chi2 only)
    {
        double *bias, *part;
        CHECK(hipMalloc(&bias, N * 8)); CHECK(hipMemset(bias, 0, N * 8));
        CHECK(hipMalloc(&part, (int64_t)M * (N / DL_CG_N) * 8));
        CHECK(hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES));
        CHECK(hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES));
        CHECK(hipFuncSetAttribute((const void*)dl_chi2_gemm_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DL_CG_LDS_BYTES));
        const int ntl = N / DL_CG_N;
        const unsigned grid = 8 * ntl * (((M + DL_CG_M - 1) / DL_CG_M + 7) / 8);
#define CHI2(L, MM) hipLaunchKernelGGL((dl_chi2_gemm_kernel<L, MM>), dim3(grid), dim3(64 * DL_CG_WAVES), DL_CG_LDS_BYTES, st, A, (int64_t)K, W, (int64_t)K, bias, part, M, K, ntl, DlChi2Fin{}, DlChi2Panels{}, K)
        printf("chi2 GEMM: grid %u, LDS %d B\n", grid, (int)DL_CG_LDS_BYTES);
        timeit("chi2: full", reps, st, [&] { CHI2(true, true); }, false, A, nA);
        timeit("chi2: full, A refilled", reps, st, [&] { CHI2(true, true); }, true, A, nA);
        timeit("chi2: no load", reps, st, [&] { CHI2(false, true); }, false, A, nA);
        timeit("chi2: no mfma", reps, st, [&] { CHI2(true, false); }, false, A, nA);
    }
    CHECK(hipStreamSynchronize(st));
    return 0;
}
