// mfma4_probe.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950: issue interval (independent / dependent accumulators) and the operand layout, found with one-hot operands.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma4_probe mfma4_probe.hip.  Round 6: does a FOUR-row tile run at the rate of the 16-row one? (it would let a wave carry
// four points of a network through all its layers alone.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC>
__global__ void rate4(double* out, long long* cycles, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.;
    double a = 1. + threadIdx.x * 1e-3, b = 1. - threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// block (la, lb): A is 1 in lane la, B is 1 in lane lb; D of every lane is written
__global__ void onehot(double* d) {
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
    const double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
    d[(size_t)blockIdx.x * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
}

template <typename K>
void run(const char* name, K kernel, int nacc, int threads, int blocks, int iters) {
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
    const double insts = (double)iters * nacc, waves = (double)threads / 64 * blocks;
    printf("%-12s nacc=%d threads=%4d blocks=%5d : %7.1f memtime-ticks/inst/wave (x 24 = shader cycles at 2.4 GHz / 100 MHz), wall %.3f ms, %.2f TFLOP/s\n", name, nacc, threads, blocks,
           h[0] / insts, ms, insts * waves * 512. / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    run("mfma_4x4x4", rate4<1>, 1, 64, 1, 100000);
    run("mfma_4x4x4", rate4<2>, 2, 64, 1, 100000);
    run("mfma_4x4x4", rate4<4>, 4, 64, 1, 100000);
    run("mfma_4x4x4", rate4<8>, 8, 64, 1, 100000);
    run("mfma_4x4x4", rate4<4>, 4, 256, 1, 100000);
    run("mfma_4x4x4", rate4<4>, 4, 512, 1, 100000);
    run("mfma_4x4x4", rate4<4>, 4, 512, 512, 100000);
    double* d; hipMalloc(&d, sizeof(double) * 4096 * 64);
    hipLaunchKernelGGL(onehot, dim3(4096), dim3(64), 0, 0, d);
    std::vector<double> h(4096 * 64); hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it meets and the D lane of each product
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int ld = 0; ld < 64; ++ld)
                if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.) printf(" (B %2d -> D %2d)", lb, ld);
        printf("\n");
    }
    return 0;
}
