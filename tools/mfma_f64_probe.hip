// mfma_f64_probe.hip -- measures v_mfma_f64_16x16x4_f64 issue interval and dependent-accumulator latency on gfx950,
// and the fp64 FMA rate, to price the roofline (the CDNA4 guide has no fp64 rows).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void probe_mfma(double* out, long long* cycles, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0., 0., 0., 0.};
    double a = 1. + threadIdx.x * 1e-3, b = 1. - threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void probe_fma(double* out, long long* cycles, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kernel, int nacc, int threads, int blocks, int iters, double flop_per_inst) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks); hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double insts_per_wave = (double)iters * nacc;
    double waves = (double)threads / 64 * blocks;
    printf("%-10s nacc=%d threads=%4d blocks=%5d : %8.1f memtime-ticks/inst/wave, wall %.3f ms, %.2f TFLOP/s\n", name, nacc, threads, blocks, h[0] / insts_per_wave, ms,
           insts_per_wave * waves * flop_per_inst / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    int iters = 20000;
    // one wave alone: dependent latency (nacc=1) and issue interval (nacc large)
    run("mfma_f64", probe_mfma<1>, 1, 64, 1, iters, 2048.);
    run("mfma_f64", probe_mfma<2>, 2, 64, 1, iters, 2048.);
    run("mfma_f64", probe_mfma<4>, 4, 64, 1, iters, 2048.);
    run("mfma_f64", probe_mfma<8>, 8, 64, 1, iters, 2048.);
    // 4 waves (one per SIMD), 8 waves (two per SIMD) on one CU
    run("mfma_f64", probe_mfma<2>, 2, 256, 1, iters, 2048.);
    run("mfma_f64", probe_mfma<2>, 2, 512, 1, iters, 2048.);
    run("mfma_f64", probe_mfma<4>, 4, 512, 1, iters, 2048.);
    // whole chip
    run("mfma_f64", probe_mfma<4>, 4, 256, 256, iters, 2048.);
    run("mfma_f64", probe_mfma<4>, 4, 512, 512, iters, 2048.);
    run("fma_f64", probe_fma<1>, 1, 64, 1, iters, 128.);
    run("fma_f64", probe_fma<8>, 8, 64, 1, iters, 128.);
    run("fma_f64", probe_fma<8>, 8, 256, 1, iters, 128.);
    run("fma_f64", probe_fma<8>, 8, 1024, 1, iters, 128.);
    run("fma_f64", probe_fma<8>, 8, 1024, 512, iters, 128.);
    return 0;
}
